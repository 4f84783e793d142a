// gemm_h2n.hip -- f16x2 GEMMs with a NARROW output over a LARGE map: N <= 96 columns, M >= tens of thousands of pixels, fp32 A (plain
// rows or the 3x3-tap gather over an NHWC map), the bits of gemm_h2_kernel.  These are the 3x3 / 1x1 convs of qres34m's bottleneck
// blocks (C/4 = 48 or 96 channels: qresvae/model.py:103-118) and the narrow 3x3 heads of qarv's latent blocks (qarv/model.py:38-39),
// which gemm_h2_kernel runs at 20-75 TFLOP/s: with one 64-wide column tile every A element is used by ONE wave, so staging it through
// LDS buys nothing, and a 16-deep stage of a 128 x 64 tile (six MFMAs per wave) cannot cover its own barrier and load latency.
// Here the roles of the operands are swapped:
//   * a wave owns 32 rows and ALL the columns (NB = 1..3 blocks of 32) and fetches its A values itself, four k16 steps ahead: two 16-B
//     loads per lane and step, four lanes on the 64 contiguous bytes a row has in a step (16 rows per instruction: a quarter of the L1
//     requests of the layout the MFMA wants, in which every lane would sit on a line of its own -- measured 67 -> ... us), split in
//     registers (split_pair_h2: the conversion gemm_h2_kernel applies before its LDS store) and turned into the MFMA's fragment layout
//     (8 consecutive k of one row per lane) through a WAVE-PRIVATE 2 x 2.5 KB LDS tile: LDS operations of one wave execute in order, so
//     this needs no barrier, only lgkmcnt; the 3x3 gather is a per-step uniform offset on the lane's pixel address, taps outside the
//     image are out-of-range addresses = hardware zeros (as in gemm_h2_kernel);
//   * the weights (pre-split, k16-interleaved: [K/16][hi 16 | lo' 16] halves per row) stream through LDS in chunks of eight k16 steps,
//     double-buffered: ONE workgroup barrier per 48..72 MFMAs of a wave, no barrier on the A side at all.
// Arithmetic: per accumulator the MFMA sequence of gemm_h2_kernel (k16 steps ascending; X += a_lo' w_hi, X += a_hi w_lo', H += a_hi w_hi),
// fma(accX, 2^-11, accH), gemm_epilogue -- every output bit equals gemm_h2_kernel's (tests/test_gpu_f16x2.py::test_gemm_h2n_equals_gemm_h2),
// so the choice between the two is the dispatcher's (shape only).  Split-K (d.ksplit = S, a function of the per-image shape) is taken in its
// SERIAL form like gemm_h2p's FOLD: the wave restarts its accumulators at every slice boundary and adds the slices' partial sums in slice
// order, then applies splitk_epilogue_store -- the operations of the parallel form + reduction in the same order, without the workspace.
#include "gemm_common.h"

#if !defined(LVAE_EXPERIMENTAL_BUILD) && (defined(H2N_EXP_CENTER) || defined(H2N_EXP_NOMFMA) || defined(H2N_EXP_NOLOAD))
#error "H2N_EXP_* timing studies (wrong results) need -DLVAE_EXPERIMENTAL_BUILD (tools/build_exp.sh)"
#endif
#ifdef H2N_EXP_NOMFMA
#define H2N_MFMA(a, b, c) (c)
#else
#define H2N_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only: __syncthreads() carries s_waitcnt vmcnt(0), i.e. would wait once per chunk for the A
// loads of the next four steps and the weight chunk requested a moment ago.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NB>
struct CfgN { static constexpr int TM = 1, TN = NB; };       // what gemm_epilogue needs of a tile configuration

constexpr int N_CS = 8;                        // k16 steps per weight chunk
constexpr int N_WROW = N_CS * 64 + 16;         // bytes of one weight row of a chunk in LDS: 528 = 132 dwords = 4 mod 64 -> the 16-B reads of
                                               // 16 consecutive rows cover all 64 banks
// A prefetch distance in k16 steps (N_CS % N_D == 0: static register ring): eight where the registers allow it
template <int NB> constexpr int n_d() { return NB <= 2 ? 8 : 4; }

template <int NB, int AMODE>
__global__ __launch_bounds__(512, 1) void gemm_h2n_kernel(const lvae_gemm_desc d) {
#pragma clang fp contract(off)
    constexpr int ROWS = NB * 32, BUF = ROWS * N_WROW, NT = 512, N_D = n_d<NB>();
    constexpr int PIECES = ROWS * N_CS * 4;                       // 16-B pieces of a chunk
    constexpr int NQW = (PIECES + NT - 1) / NT;                    // ... per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    // consecutive 256-row tiles stay on one XCD (workgroups go round-robin over the eight): the 3x3 gather's rows above and below a tile
    // are its neighbours' rows -- one L2 then holds them once instead of three L2s fetching them each
    int m0;
    {
        const int n_tiles = (int)gridDim.x, b = blockIdx.x, q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        m0 = ((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc) * 256;
    }
    // load layout of a step: instruction j covers rows 16 j .. 16 j + 15 of the wave's 32, lane l the 16-B piece l % 4 of row 16 j + l / 4
    const int lrow = lane >> 2, lpc = lane & 3;
    const int nq = d.K / 16;                                       // k16 steps
    const int S = d.ksplit > 1 ? d.ksplit : 1, per = nq / S;       // steps per K slice
    const int nchunk = (nq + N_CS - 1) / N_CS;

    // ---- weights: piece e = tid + NT * j of a chunk: row e / 32, k16 group (e % 32) / 4, 16-B piece e % 4
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)d.Wt16, 0, (int)((long)d.N * d.K * 4), 0x00020000);
    int w_voff[NQW], w_st[NQW], w_grp[NQW];
#pragma unroll
    for (int j = 0; j < NQW; ++j) {
        const int e = tid + NT * j, r = e >> 5, g = (e & 31) >> 2, pc = e & 3;
        w_voff[j] = (e < PIECES && r < d.N) ? r * (d.K * 4) + g * 64 + pc * 16 : 0x7fffffff;
        w_st[j] = r * N_WROW + g * 64 + pc * 16;
        w_grp[j] = g;
    }
    u32x4 wr[NQW];
    auto load_w = [&](int c) {                                     // chunk c -> registers (groups beyond the last k16 step: zeros)
#pragma unroll
        for (int j = 0; j < NQW; ++j) {
            const bool ok = c * N_CS + w_grp[j] < nq;
            wr[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, ok ? w_voff[j] + c * (N_CS * 64) : 0x7fffffff, 0, 0);
        }
    };
    auto store_w = [&](char* buf) {
#pragma unroll
        for (int j = 0; j < NQW; ++j)
            if (tid + NT * j < PIECES) *(u32x4*)(buf + w_st[j]) = wr[j];
    };

    // ---- A: two 16-B loads per k16 step and lane (rows lrow and 16 + lrow of the wave, floats 4 lpc .. + 3 of the step's 16)
    int a_voff[2], tapok[2] = {0, 0};
    __amdgpu_buffer_rsrc_t rsA;
    if (AMODE == LVAE_A_CONV3) {
        // K = 9 * Cin tap-major; a k16 step lies inside one tap (Cin % 16 == 0)
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A0, 0, (int)((long)d.M * d.K0 * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = m0 + wave * 32 + 16 * j + lrow;
            const int w = row % d.W, h = (row / d.W) % d.H;
            a_voff[j] = row < d.M ? (int)(((long)row * d.K0 + 4 * lpc) * 4) : 0x7fffffff;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
                tapok[j] |= (row < d.M && hh >= 0 && hh < d.H && ww >= 0 && ww < d.W) ? (1 << t) : 0;
            }
        }
    } else {
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A0, 0, (int)((long)d.M * d.lda0 * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = m0 + wave * 32 + 16 * j + lrow;
            a_voff[j] = row < d.M ? (int)(((long)row * d.lda0 + 4 * lpc) * 4) : 0x7fffffff;
        }
    }
    u32x4 ar[N_D][2];
    auto load_a = [&](int slot, int q) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int vo;
            if (AMODE == LVAE_A_CONV3) {
                const int kq = q * 16, tap = kq / d.K0, kk = kq - tap * d.K0;                 // uniform
#ifdef H2N_EXP_CENTER        // timing study (wrong results): every tap reads the centre pixel -- what is left when the gather's re-reads hit L1
                const int toff = kk * 4;
#else
                const int toff = (((tap / 3 - 1) * d.W + (tap % 3 - 1)) * d.K0 + kk) * 4;
#endif
                vo = ((tapok[j] >> tap) & 1) ? a_voff[j] + toff : 0x7fffffff;
            } else {
                vo = a_voff[j] == 0x7fffffff ? a_voff[j] : a_voff[j] + q * 64;
            }
#ifdef H2N_EXP_NOLOAD          // timing study: the A loads of the steady state go to an out-of-range address (no memory traffic)
            if (q >= n_d<NB>()) vo = 0x7fffffff;
#endif
            ar[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0);
        }
    };
    // wave-private fragment tile: 32 rows x (32 B hi | 32 B lo' | 16 pad) -- gemm_h2_kernel's 80-B rows (conflict-free both ways), two of them
    char* atile = lds + 2 * BUF + wave * (2 * 32 * 80);
    const int at_w = lrow * 80 + lpc * 8;                          // + 16 j rows, + 32: lo' plane
    const int at_r = li * 80 + 16 * lh;

    f32x16 accH[NB], accX[NB], tot[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accH[b][r] = 0.f; accX[b][r] = 0.f; tot[b][r] = 0.f; }
    int in_slice = 0, slice = 0;

    // prologue: chunk 0 of the weights, the first N_D steps of A
    load_w(0);
#pragma unroll
    for (int s = 0; s < N_D; ++s) load_a(s, s < nq ? s : nq - 1);
    store_w(lds);
    load_w(nchunk > 1 ? 1 : 0);
    lds_barrier();

    const int b_fr = li * N_WROW + 16 * lh;
    // one k16 step (s: position in its chunk, compile time after unrolling; q: global step)
    auto step = [&](const char* cur, int s, int q) __attribute__((always_inline)) {
        // split this step's A values (gemm_h2_kernel's split_half) into the fragment tile, then refill the ring slot (beyond the last step:
        // a harmless re-read of it -- the steady state has no conditional loads, so the compiler's vmcnt bookkeeping stays exact)
        char* at = atile + (s & 1) * (32 * 80);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned h0, l0, h1, l1;
            split_pair_h2(__uint_as_float(ar[s % N_D][j][0]), __uint_as_float(ar[s % N_D][j][1]), h0, l0);
            split_pair_h2(__uint_as_float(ar[s % N_D][j][2]), __uint_as_float(ar[s % N_D][j][3]), h1, l1);
            *(u32x2_t*)(at + 16 * j * 80 + at_w) = (u32x2_t){h0, h1};
            *(u32x2_t*)(at + 16 * j * 80 + at_w + 32) = (u32x2_t){l0, l1};
        }
        load_a(s % N_D, q + N_D < nq ? q + N_D : nq - 1);
        // (same wave: the LDS unit executes a wave's operations in order -- the reads below see the writes above, and the writes of step
        //  s + 2 into this tile come after these reads; the compiler's lgkmcnt covers the register side)
        asm volatile("" ::: "memory");
        const f16x8 ahi = *(const f16x8*)(at + at_r), alo = *(const f16x8*)(at + at_r + 32);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const f16x8 whi = *(const f16x8*)(cur + b * 32 * N_WROW + b_fr + s * 64);
            const f16x8 wlo = *(const f16x8*)(cur + b * 32 * N_WROW + b_fr + s * 64 + 32);
            accX[b] = H2N_MFMA(alo, whi, accX[b]);
            accX[b] = H2N_MFMA(ahi, wlo, accX[b]);
            accH[b] = H2N_MFMA(ahi, whi, accH[b]);
        }
        if (S > 1 && ++in_slice == per) {
            // the slice's partial sum exactly as the parallel form stores it: fma(accX, 2^-11, accH), then the epilogue's "+ bias" with no
            // bias (x + 0.0f: turns -0 into +0); slices are added in order (gemm_h2p.hip's fold)
            in_slice = 0;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = __builtin_fmaf(accX[b][r], 1.0f / 2048.0f, accH[b][r]) + 0.0f;
                    tot[b][r] = slice == 0 ? pr : tot[b][r] + pr;
                    accH[b][r] = 0.f; accX[b][r] = 0.f;
                }
            ++slice;
        }
    };
    for (int c = 0; c + 1 < nchunk; ++c) {                        // every chunk but the last: N_CS steps, straight-line
        const char* cur = lds + (c & 1) * BUF;
#pragma unroll
        for (int s = 0; s < N_CS; ++s) step(cur, s, c * N_CS + s);
        // chunk c + 1 (in registers since the last barrier) -> the other buffer, whose last readers passed that barrier; chunk c + 2 on its
        // way (beyond the last: a re-read of it)
        store_w(lds + ((c + 1) & 1) * BUF);
        load_w(c + 2 < nchunk ? c + 2 : nchunk - 1);
        lds_barrier();
    }
    {
        const int c = nchunk - 1;                                  // the last chunk: 1 .. N_CS steps
        const char* cur = lds + (c & 1) * BUF;
#pragma unroll
        for (int s = 0; s < N_CS; ++s)
            if (c * N_CS + s < nq) step(cur, s, c * N_CS + s);
    }

    if (S > 1) {
        // the reduce kernel's tail: 4 consecutive columns of one row per lane (quad transpose), then its epilogue function
        const int lj = li & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int orow = m0 + wave * 32 + 4 * lh + 8 * g + lj;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float v0 = tot[b][4 * g + 0], v1 = tot[b][4 * g + 1], v2 = tot[b][4 * g + 2], v3 = tot[b][4 * g + 3];
                quad_transpose(v0, v1, v2, v3, lj);
                const int c4 = b * 32 + (li & ~3);
                if (orow < d.M && c4 < d.N) splitk_epilogue_store<false>(d, orow, c4, (f32x4){v0, v1, v2, v3});
            }
        }
        return;
    }
    f32x16 acc[1][NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][b][r] = __builtin_fmaf(accX[b][r], 1.0f / 2048.0f, accH[b][r]);
    gemm_epilogue<CfgN<NB>>(d, acc, m0, 0, wave, 0, li, lh);
}

template <int NB, int AMODE>
int launch_h2n(const lvae_gemm_desc* d, hipStream_t st) {
    constexpr int LDS = 2 * NB * 32 * N_WROW + 8 * 2 * 32 * 80;          // two weight chunks + the eight waves' fragment tiles
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)gemm_h2n_kernel<NB, AMODE>, LDS)) return ae;
    hipLaunchKernelGGL((gemm_h2n_kernel<NB, AMODE>), dim3((d->M + 255) / 256), dim3(512), LDS, st, *d);
    return (int)hipGetLastError();
}

}  // namespace

// Entry point for gemm_h2.hip's dispatcher.  -> 1 when this kernel takes the problem (*rc = launch status).  force: take it whenever the
// kernel CAN (tests); otherwise where it is the faster form (shape rule below; same bits either way).
int lvae_gemm_h2n_try(const lvae_gemm_desc* d, hipStream_t st, int force, int* rc) {
    const bool conv3 = d->a_mode == LVAE_A_CONV3;
    if (d->prec != 4 || d->a_h2 || d->a_gelu || d->out_h2 || (d->a_mode != LVAE_A_PLAIN && !conv3) || d->N > 96 || (d->K & 15) || d->ldw != d->K) return 0;
    if (!conv3 && (d->K1 != 0 || d->K0 != d->K || (d->lda0 & 3) || (long)d->M * d->lda0 * 4 > 0x7ffffff0L)) return 0;
    if (conv3 && ((d->K0 & 15) || d->K != 9 * d->K0 || d->K1 != 0 || d->H <= 0 || d->W <= 0 || (long)d->M * d->K0 * 4 > 0x7ffffff0L)) return 0;
    if ((long)d->N * d->K * 4 > 0x7ffffff0L) return 0;
    const int S = d->ksplit > 1 ? d->ksplit : 1;
    if (S > 1 && ((d->K / 16) % S || d->store != LVAE_ST_ROWMAJOR || (d->N & 3) || (d->ldo & 3) || (d->ldres & 3))) return 0;
    // shape rule: K = 16 (mod 32) has no other f16x2 kernel (gemm_h2_kernel walks the k16 steps in pairs) -- always; otherwise where it is
    // the faster of the two (256-row workgroups: fewer than 128 of them leave the chip idle)
    // (measured, profiles/r04_gemm_h2n_narrow_output.txt: 3x3 gathers and N = 65..96 from M = 49152 on: 1.1 - 1.25x; N <= 64 plain rows and
    //  the split-K launches: gemm_h2_kernel is as fast or faster)
    if (!force && !(d->K & 16) && !(d->M >= 32768 && S == 1 && (conv3 || d->N > 64))) return 0;
    const int nb = (d->N + 31) / 32;
#define LVAE_H2N_LAUNCH(AM) (nb == 1 ? launch_h2n<1, AM>(d, st) : nb == 2 ? launch_h2n<2, AM>(d, st) : launch_h2n<3, AM>(d, st))
    *rc = conv3 ? LVAE_H2N_LAUNCH(LVAE_A_CONV3) : LVAE_H2N_LAUNCH(LVAE_A_PLAIN);
#undef LVAE_H2N_LAUNCH
    return 1;
}
