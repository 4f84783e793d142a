// gemm_x3v2.hip -- software-pipelined bf16x3 split-MFMA GEMMs (prec 2) for the plain-A case: timm Mlp fc1/fc2 and the 1x1
// convs (lvae/models/common.py:131-132,154; qarv/model.py:36,38,39), ~90 % of the path's FLOPs.
//
// Why a second generation of prec-2 kernels: on gfx950 an MFMA in flight blocks the VALU of every OTHER wave on its SIMD
// (tools/ubench/mfma_valu_overlap*.hip; s_setprio does not change that: mfma_valu_prio.hip), but up to ~6 independent VALU /
// LDS / VMEM instructions of the SAME wave issue for free in each 32-cycle MFMA shadow (mfma_valu_interleave.hip: 32.1 -> 34.1
// cycles per MFMA with 5 fillers, 37 with 6, +4.5 per filler beyond).  gemm_x3_kernel (gemm_f32.hip) runs its fp32 -> 3 x bf16
// operand split, its LDS stores and its global loads in a separate phase between two barriers: PMC showed 42 % MFMA-busy with
// 6 VALU instructions per MFMA.  Here every non-MFMA instruction of the main loop is placed, in program order and fenced with
// sched_barrier, into the shadow of an MFMA pair of the same wave ("filler slices"):
//   * LDS fragment prefetch for the next MFMA group, buffer loads two stages ahead, the split of one 4-element chunk of A
//     (pinned with an empty asm so LLVM cannot sink it to its consumer), its ds_write into the OTHER LDS stage, its reload;
//   * LDS is double-buffered, so a stage costs one barrier and one exposed fragment read -- no ds_write phase;
//   * buffer loads with hardware range checking (rows beyond M / N read as zero) and uniform (SGPR) stage offsets replace
//     per-row pointers, clamps and ok-masks: ~35 VGPRs and all address VALU gone;
//   * the load issue order is the same in the prologue and in the loop: the s_waitcnt vmcnt values hipcc derives at the loop
//     header are the worst case over both predecessors.
// Per accumulator the sequence of MFMAs and their operands is exactly gemm_x3_kernel's, so all prec-2 kernels are
// bit-identical (tests/test_gpu_bf16.py) and the choice between them may depend on M (encoder vs decoder priors stay equal).
// Measured (B = 8 model shapes, tools/microbench.py): 111 -> 125 TF/s aggregate; 165 TF/s on the large layers at an effective
// 1.8 GHz (the chip is power-limited under dense bf16 MFMA: MI355X_MICROARCH.md, DVFS give-back).
#include "gemm_common.h"

#include <type_traits>

#ifdef LVAE_X3V2_TRACE             // tools/ubench/x3k16_trace.hip: s_memtime stamps of one wave per k16 stage
extern "C" __device__ long* lvae_trace_buf;
#define X3_STAMP(i) do { if (tracing) tstamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define X3_STAMP(i) do {} while (0)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define LVAE_FENCE() __builtin_amdgcn_sched_barrier(0)


// x (2 floats) -> packed bf16 pairs hi, mid, lo with hi + mid + lo == x to 2^-25 relative (round-to-nearest-even each time):
// the same three conversions gemm_x3_kernel and the host-side weight split perform.
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const f32x2 x = {x0, x1};
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
    lo = __builtin_bit_cast(unsigned, l);
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_x3w8_kernel: 256 x 128 tile, 8 waves (4 x 2, wave tile 64 x 64), ONE workgroup per CU, DOUBLE-buffered LDS
// (2 x 384 rows x 208 B = 156 KiB).  In a single-stage pipelined kernel (this file's first form, in the history) the ds_write phase
// between the two barriers cost ~1000 of a k-tile's ~3900 cycles (in-kernel s_memtime timeline: the VGPR->LDS write path moves
// ~70 B/clk per CU and all four waves
// write at once) and starves the co-resident workgroup's ds_reads.  Here tile t+1 is split and written into the other LDS
// stage by filler instructions in the MFMA shadows of tile t (chunk by chunk: split -> 3 x ds_write_b64 -> reload for t+2),
// so a k-tile costs one barrier plus one exposed fragment read.  Staging rows are permuted (bit 0 <-> bit 2 of the row
// index) so that the 16-lane (b64) / 8-lane (b128) LDS write groups cover 32 distinct banks (208-B rows: rows r and r+4 are
// 16 banks apart) -- an unpermuted map loses a third of its LDS write cycles to 2-way conflicts (PMC SQ_LDS_BANK_CONFLICT).
// Same per-accumulator MFMA sequence as the other two prec-2 kernels => bit-identical results.
template <bool AGELU>
__global__ __launch_bounds__(512, 1) void gemm_x3w8_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    using C = Cfg<4, 2, 2, 2, 2, 32>;                  // 256 x 128
    constexpr int ROWB = 208, STAGE = (256 + 128) * ROWB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;                           // stage s: A rows at s*STAGE, W rows at s*STAGE + 256*ROWB
    int t;
    {
        const int b = blockIdx.x, q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * C::BM, n0 = tn * C::BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    auto perm = [](int q) { return (q & ~7) | ((q & 1) << 2) | ((q >> 1) & 3); };
    const int arow = perm(tid >> 3), ak4 = tid & 7, wrow = perm(tid >> 2), wk8 = tid & 3;     // A rows arow + 64 i, i < 4
    const int rows_a = (d.M - m0) < C::BM ? (d.M - m0) : C::BM;
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.A0 + (long)m0 * d.lda0), 0, rows_a * d.lda0 * 4, 0x00020000);
    const long wrem = ((long)3 * d.N - n0) * d.ldw * 2;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(d.Wt16 + (long)n0 * d.ldw), 0, wrem > 0x7fffffffL ? 0x7fffffff : (int)wrem, 0x00020000);
    const int a_voff = (arow * d.lda0 + ak4 * 4) * 4, a_rowstep = 64 * d.lda0 * 4;
    const int w_voff = (wrow * d.ldw + wk8 * 8) * 2, w_plane = d.N * d.ldw * 2;
    const int a_st = arow * ROWB + ak4 * 8;
    const int w_st = (256 + wrow) * ROWB + wk8 * 16;
    const int a_fr = (wave_m * 64 + li) * ROWB + 16 * lh;
    const int b_fr = (256 + wave_n * 64 + li) * ROWB + 16 * lh;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    u32x4 ra[4], rb[3];
    u32x2 sa[3];
    const int nk = d.K / 32;
    auto load_a = [&](int i, int kt) { ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, a_voff, i * a_rowstep + kt * 128, 0); };
    auto load_w = [&](int p, int kt) { rb[p] = __builtin_amdgcn_raw_buffer_load_b128(rsW, w_voff, p * w_plane + kt * 64, 0); };
    auto split_half = [&](int i, int h) {
        float x0 = __uint_as_float(ra[i][2 * h]), x1 = __uint_as_float(ra[i][2 * h + 1]);
        if (AGELU) { x0 = gelu_erf(x0); x1 = gelu_erf(x1); }
        unsigned hi, mid, lo;
        split_pair(x0, x1, hi, mid, lo);
        asm volatile("" : "+v"(hi), "+v"(mid), "+v"(lo));
        sa[0][h] = hi; sa[1][h] = mid; sa[2][h] = lo;
    };
    auto store_a = [&](char* st, int i) {
#pragma unroll
        for (int p = 0; p < 3; ++p) *(u32x2*)(st + a_st + i * 64 * ROWB + p * 64) = sa[p];
    };
    auto store_w = [&](char* st, int p) { *(u32x4*)(st + w_st + p * 64) = rb[p]; };

    // prologue: tile 0 -> stage 0, tile 1 -> registers
#pragma unroll
    for (int i = 0; i < 4; ++i) load_a(i, 0);
#pragma unroll
    for (int p = 0; p < 3; ++p) load_w(p, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { split_half(i, 0); split_half(i, 1); store_a(lds, i); }
#pragma unroll
    for (int p = 0; p < 3; ++p) store_w(lds, p);
    {
        // same issue order as inside the loop (A0, W0..2, A1..3): the vmcnt the compiler derives at the loop header is the
        // worst case over both predecessors, and a different order here would make every wait in the loop conservative
        const int k1 = nk > 1 ? 1 : 0;
        load_a(0, k1);
#pragma unroll
        for (int p = 0; p < 3; ++p) load_w(p, k1);
#pragma unroll
        for (int i = 1; i < 4; ++i) load_a(i, k1);
    }
    __syncthreads();

    bf16x8 af[2][2][3], bf[2][3];
    for (int kt = 0; kt < nk; ++kt) {
        const int kt2 = kt + 2 < nk ? kt + 2 : nk - 1;
        const char* cur = lds + (kt & 1) * STAGE;
        char* nxt = lds + ((kt & 1) ^ 1) * STAGE;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[0][a][p] = *(const bf16x8*)(cur + a_fr + a * 32 * ROWB + 64 * p);
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[0][p] = *(const bf16x8*)(cur + b_fr + 64 * p);
        LVAE_FENCE();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int s = g >> 1, b = g & 1, bb = g & 1;
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][0][PA[j]], bf[bb][PB[j]], acc[0][b], 0, 0, 0);
                acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][1][PA[j]], bf[bb][PB[j]], acc[1][b], 0, 0, 0);
                // ---- filler slice j of group g
                if (j == 0 && g < 3) {
                    const int s1 = (g + 1) >> 1, b1 = (g + 1) & 1;
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[bb ^ 1][p] = *(const bf16x8*)(cur + b_fr + b1 * 32 * ROWB + 64 * p + 32 * s1);
                }
                if (g == 1 && (j == 1 || j == 2)) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) af[1][j - 1][p] = *(const bf16x8*)(cur + a_fr + (j - 1) * 32 * ROWB + 64 * p + 32);
                }
                if (g == 0 && j == 2) { store_w(nxt, 0); store_w(nxt, 1); }
                if (g == 0 && j == 4) { store_w(nxt, 2); }
                if (g == 1 && j == 0) { load_w(0, kt2); load_w(1, kt2); }
                if (g == 1 && j == 5) { load_w(2, kt2); }
                if (j == 1) split_half(g, 0);
                if (j == 3) split_half(g, 1);
                if (j == 4) store_a(nxt, g);
                if (j == 5) load_a(g, kt2);
                LVAE_FENCE();
            }
        }
        __syncthreads();
    }
    gemm_finish<C>(d, acc, m0, n0, wave_m, wave_n, li, lh, (void*)smem, t);
}

template <bool AGELU>
int launch_w8(const lvae_gemm_desc* d, hipStream_t st) {
    constexpr int LDS = 2 * (256 + 128) * 208;
    static_assert(LDS <= 160 * 1024, "LDS");
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)gemm_x3w8_kernel<AGELU>, LDS)) return ae;
    const int tiles_m = (d->M + 255) / 256, tiles_n = (d->N + 127) / 128, n_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_x3w8_kernel<AGELU>), dim3(n_tiles), dim3(512), LDS, st, *d, tiles_n, n_tiles);
    return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------
// gemm_x3k16_kernel: 4 waves, 128 x (64*TN) tiles, two workgroups per CU, with DOUBLE-buffered LDS
// made affordable by 16-deep stages: 2 x (128 + 64*TN) rows x 112 B (3 planes x 32 B + 16 pad) = 70 KB for TN = 3.  Each
// stage feeds one k16 MFMA step (12*TN MFMAs per wave); the next stage is split / written by fillers in those MFMAs'
// shadows, so a stage costs ONE barrier and one exposed fragment read, and the ~1000-cycle ds_write phase of the k32
// single-stage kernels is gone.  Register sets alternate by stage parity and are reloaded two stages ahead (a whole k32 of
// latency budget).  W comes from the k16-interleaved copy [N][K/16][3][16] that follows the three planes in the prec-2 weight
// buffer (lvae.models.base.pack_bf16x3): one stage of one row is 96 contiguous bytes.  Staging rows are enumerated
// 0,2,4,6,1,3,5,7 so that every LDS write group (16 lanes b64 / 8 lanes b128) covers 32 distinct banks with 112-B rows.
// Per accumulator: k16 steps in ascending k, six cross terms in gemm_x3_kernel's order => bit-identical to it.
template <int TN, bool AGELU, int AMODE>
// (TN = 1: three waves per SIMD -- what its 43 KB of LDS allow; without the bound the epilogue's residual prefetch cost it the third)
__global__ __launch_bounds__(256, (TN == 1 ? 3 : 2)) void gemm_x3k16_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    using C = Cfg<2, 2, 2, TN, 1, 32>;
    constexpr int ROWB = 112, ROWS = 128 + 64 * TN, STAGE = ROWS * ROWB;
    constexpr int NWC = 384 * TN, NW = (NWC + 255) / 256;          // 16-B W chunks per stage, per thread (wrap-around duplicates)
    constexpr int G = TN;                                          // MFMA groups per stage (one per n-block), 6 slices each
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    int t;
    {
        const int b = blockIdx.x, q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * C::BM, n0 = tn * C::BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    auto perm = [](int q) { return (q & ~7) | ((q & 3) << 1) | ((q >> 2) & 1); };        // 0,2,4,6,1,3,5,7
#ifdef LVAE_EXP_PRIO
    // the two workgroups of a CU get different issue priorities (told apart by their LDS allocation base), so that they drift
    // out of phase instead of reaching their per-stage barriers together
    if (__builtin_amdgcn_s_getreg(((8 - 1) << 11) | (0 << 6) | 6) != 0) __builtin_amdgcn_s_setprio(LVAE_EXP_PRIO);
#endif

    // split-K (gridDim.y slices): this workgroup covers k16 stages [q0, q0 + nq); the slice offset goes into the buffer bases
    const int nq = d.K / 16 / (int)gridDim.y, q0 = (int)blockIdx.y * nq;
    const int rows_a = (d.M - m0) < C::BM ? (d.M - m0) : C::BM;
    // A = [A0 | A1] along K (fused torch.cat, qarv/model.py:66-67): stages below K0 stream from A0, the others from A1
    const float* a0b = d.A0 + (long)m0 * d.lda0;
    const float* a1p = d.A1 ? d.A1 : d.A0;
    const long lda1 = d.A1 ? d.lda1 : d.lda0;
    const float* a1b = a1p + (long)m0 * lda1;
    const int n0rec = rows_a * d.lda0 * 4, n1rec = rows_a * (int)lda1 * 4;
    const int qsplit = d.K0 / 16;                                 // first stage that reads A1
    const int rows_w = (d.N - n0) < C::BN ? (d.N - n0) : C::BN;
    const long wrow_b = (long)6 * d.K;                            // bytes per W row in the k16-interleaved copy
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(d.Wt16 + (long)3 * d.N * d.ldw + (long)n0 * 3 * d.K + q0 * 48), 0, (int)(rows_w * wrow_b) - q0 * 96, 0x00020000);
    int a_voff[2], a_voff1[2], a_st[2], w_voff[NW], w_st[NW];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = tid + 256 * j, row = perm(c >> 2), ak4 = c & 3;
        a_voff[j] = (row * d.lda0 + ak4 * 4) * 4;
        a_voff1[j] = (row * (int)lda1 + ak4 * 4) * 4;
        a_st[j] = row * ROWB + ak4 * 8;
    }
    // 3x3-tap gather (implicit GEMM over an NHWC map, K = 9*Cin, tap-major): a stage of 16 channels lies inside one tap (Cin % 16
    // == 0); the tap is a uniform offset added to the row's own pixel address, and a tap outside the image turns the address into an
    // out-of-range one, i.e. a zero from the buffer unit -- no masks on the data path.
    int tapok[2] = {0, 0};
    if (AMODE == LVAE_A_CONV3) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = tid + 256 * j, m = m0 + perm(c >> 2), ak4 = c & 3;
            const int w = m % d.W, h = (m / d.W) % d.H;
            a_voff[j] = m < d.M ? (int)(((long)m * d.K0 + ak4 * 4) * 4) : 0x7fffffff;
#pragma unroll
            for (int sidx = 0; sidx < 9; ++sidx) {
                const int hh = h + sidx / 3 - 1, ww = w + sidx % 3 - 1;
                tapok[j] |= (m < d.M && hh >= 0 && hh < d.H && ww >= 0 && ww < d.W) ? (1 << sidx) : 0;
            }
        }
    }
    const int a_all_rec = (AMODE == LVAE_A_CONV3) ? (int)((long)d.M * d.K0 * 4) : 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int c = (tid + 256 * j) % NWC, row = perm(c / 6), piece = c % 6;
        w_voff[j] = row * (int)wrow_b + piece * 16;
        w_st[j] = (128 + row) * ROWB + piece * 16;
    }
    const int a_fr = (wave_m * 64 + li) * ROWB + 16 * lh;
    const int b_fr = (128 + wave_n * TN * 32 + li) * ROWB + 16 * lh;

    f32x16 acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    u32x4 ra[2][2], rb[2][NW];        // [stage parity][chunk]
    u32x2 sa[3];
    auto load_a = [&](int par, int j, int q) {      // q is slice-local; the source is chosen per stage with scalar selects (no branch:
        const int qg = q0 + q;                        // a branch would cut the fenced MFMA / filler stream into basic blocks)
        if (AMODE == LVAE_A_CONV3) {
            const int kq = qg * 16, tap = kq / d.K0, kk = kq - tap * d.K0;              // uniform
            int toff = (((tap / 3 - 1) * d.W + (tap % 3 - 1)) * d.K0 + kk) * 4;
            asm volatile("" : "+s"(toff));          // computed here, unconditionally: otherwise hipcc sinks it into a divergent branch
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.A0, 0, a_all_rec, 0x00020000);
            int vo = a_voff[j] + toff;
            vo = ((tapok[j] >> tap) & 1) ? vo : 0x7fffffff;
            ra[par][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
            return;
        }
        const bool second = qg >= qsplit;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(second ? a1b : a0b), 0, second ? n1rec : n0rec, 0x00020000);
        ra[par][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, second ? a_voff1[j] : a_voff[j], (second ? qg - qsplit : qg) * 64, 0);
    };
    auto load_w = [&](int par, int j, int q) { rb[par][j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, w_voff[j], q * 96, 0); };
    auto split_half = [&](int par, int j, int h) {
        float x0 = __uint_as_float(ra[par][j][2 * h]), x1 = __uint_as_float(ra[par][j][2 * h + 1]);
        if (AGELU) { x0 = gelu_erf(x0); x1 = gelu_erf(x1); }
        unsigned hi, mid, lo;
#ifdef LVAE_EXP_NOSPLIT            // experiment (wrong results): what the operand split costs inside the main loop
        hi = ra[par][j][2 * h]; mid = ra[par][j][2 * h + 1]; lo = hi;
#else
        split_pair(x0, x1, hi, mid, lo);
#endif
        asm volatile("" : "+v"(hi), "+v"(mid), "+v"(lo));
        sa[0][h] = hi; sa[1][h] = mid; sa[2][h] = lo;
    };
    auto store_a = [&](char* st, int j) {
#ifdef LVAE_EXP_NOSTORE
        if (d.M > 0) return;
#endif
#pragma unroll
        for (int p = 0; p < 3; ++p) *(u32x2*)(st + a_st[j] + p * 32) = sa[p];
    };
    auto store_w = [&](char* st, int par, int j) {
#ifdef LVAE_EXP_NOSTORE
        if (d.M > 0) return;
#endif
        *(u32x4*)(st + w_st[j]) = rb[par][j];
    };
    // load order of one register set: A0, W0, W1, A1, W2, W3, W4 -- the same in the prologue and in the loop (vmcnt bookkeeping)
    auto load_set = [&](int par, int q) {
        load_a(par, 0, q);
        if (NW > 0) load_w(par, 0, q);
        if (NW > 1) load_w(par, 1, q);
        load_a(par, 1, q);
#pragma unroll
        for (int j = 2; j < NW; ++j) load_w(par, j, q);
    };

    // prologue: stage 0 <- k16-tile 0; set 1 <- tile 1, set 0 <- tile 2
    load_set(0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) { split_half(0, j, 0); split_half(0, j, 1); store_a(lds, j); }
#pragma unroll
    for (int j = 0; j < NW; ++j) store_w(lds, 0, j);
    load_set(1, nq > 1 ? 1 : nq - 1);
    load_set(0, nq > 2 ? 2 : nq - 1);
    __syncthreads();

    bf16x8 af[2][3], bf[2][3];
#ifdef LVAE_X3V2_TRACE
#if LVAE_X3V2_TRACE == 2           // every workgroup records: [block][8 header + 4 * 256 stamps]; header: start, end, HW_ID, XCC_ID
    const bool tracing = tid == 0;
    long* const tbuf = lvae_trace_buf + (long)blockIdx.x * (8 + 4 * 256);
    if (tracing) {
        tbuf[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        tbuf[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        tbuf[4] = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    }
#else
    const bool tracing = blockIdx.x == (unsigned)(n_tiles / 2 + 8) && tid == 0;
    long* const tbuf = lvae_trace_buf - 8;
#endif
    long tstamp[4] = {0, 0, 0, 0};
    if (tracing) lvae_trace_buf[LVAE_X3V2_TRACE == 2 ? (long)blockIdx.x * (8 + 4 * 256) : 120] = __builtin_readcyclecounter();
#endif
    // one k16 stage: compute on stage PAR, write tile q+1 from register set PAR^1 into the other stage, reload that set with q+3
    auto body = [&](auto par_tag, int q) {
        constexpr int PAR = decltype(par_tag)::value, OTH = PAR ^ 1;
        const int q3 = q + 3 < nq ? q + 3 : nq - 1;
        const char* cur = lds + PAR * STAGE;
        char* nxt = lds + OTH * STAGE;
        X3_STAMP(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[a][p] = *(const bf16x8*)(cur + a_fr + a * 32 * ROWB + 32 * p);
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[0][p] = *(const bf16x8*)(cur + b_fr + 32 * p);
        LVAE_FENCE();
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int bb = g & 1;
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                acc[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][PA[j]], bf[bb][PB[j]], acc[0][g], 0, 0, 0);
                acc[1][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][PA[j]], bf[bb][PB[j]], acc[1][g], 0, 0, 0);
                const int S = g * 6 + j;                                  // filler slice index within the stage
                if (j == 0 && g + 1 < G) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[bb ^ 1][p] = *(const bf16x8*)(cur + b_fr + (g + 1) * 32 * ROWB + 32 * p);
                }
                if (TN >= 2) {
                    // A chunk i: split in slices 6i+1 / 6i+3, written in 6i+4, reloaded (with W 2i, 2i+1) in 6i+5
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (S == 6 * i + 1) split_half(OTH, i, 0);
                        if (S == 6 * i + 3) split_half(OTH, i, 1);
                        if (S == 6 * i + 4) store_a(nxt, i);
                        if (S == 6 * i + 5) load_a(OTH, i, q3);
                    }
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        if (S == 2 + 6 * (w >> 1)) store_w(nxt, OTH, w);
                        if (S == 5 + 6 * (w >> 1)) load_w(OTH, w, q3);
                    }
                } else {
                    if (S == 0) { split_half(OTH, 0, 0); store_w(nxt, OTH, 0); }
                    if (S == 1) { split_half(OTH, 0, 1); store_w(nxt, OTH, 1); }
                    if (S == 2) { store_a(nxt, 0); load_a(OTH, 0, q3); load_w(OTH, 0, q3); load_w(OTH, 1, q3); }
                    if (S == 3) split_half(OTH, 1, 0);
                    if (S == 4) split_half(OTH, 1, 1);
                    if (S == 5) { store_a(nxt, 1); load_a(OTH, 1, q3); }
                }
                LVAE_FENCE();
            }
            if (g == 0) X3_STAMP(1);
        }
        X3_STAMP(2);
        __syncthreads();
        X3_STAMP(3);
#ifdef LVAE_X3V2_TRACE
        if (tracing && q < (LVAE_X3V2_TRACE == 2 ? 256 : 28)) for (int z = 0; z < 4; ++z) tbuf[8 + q * 4 + z] = tstamp[z];
#endif
    };
    for (int q = 0; q < nq; q += 2) {
        body(std::integral_constant<int, 0>{}, q);
        body(std::integral_constant<int, 1>{}, q + 1);
    }
#ifdef LVAE_X3V2_TRACE
    if (tracing) lvae_trace_buf[LVAE_X3V2_TRACE == 2 ? (long)blockIdx.x * (8 + 4 * 256) + 1 : 121] = __builtin_readcyclecounter();
#endif
    gemm_finish<C>(d, acc, m0, n0, wave_m, wave_n, li, lh, (void*)smem, t);
#ifdef LVAE_X3V2_TRACE
    if (tracing) lvae_trace_buf[LVAE_X3V2_TRACE == 2 ? (long)blockIdx.x * (8 + 4 * 256) + 5 : 122] = __builtin_readcyclecounter();
#endif
}

#ifdef LVAE_X3V2_TRACE
int g_x3v2_lds_pad = 0;            // trace harness: extra dynamic LDS to force one workgroup per CU
#else
constexpr int g_x3v2_lds_pad = 0;
#endif

template <int TN, bool AGELU, int AMODE>
int launch_k16(const lvae_gemm_desc* d, hipStream_t st) {
    constexpr int BN = 64 * TN, LDS = 2 * (128 + BN) * 112;
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)gemm_x3k16_kernel<TN, AGELU, AMODE>, LDS + 64 * 1024)) return ae;
    const int tiles_m = (d->M + 127) / 128, tiles_n = (d->N + BN - 1) / BN, n_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_x3k16_kernel<TN, AGELU, AMODE>), dim3(n_tiles, d->ksplit > 1 ? d->ksplit : 1), dim3(256), LDS + g_x3v2_lds_pad, st, *d,
                       tiles_n, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace

// Entry point for gemm_f32.hip's dispatcher.  Returns 1 when the problem is one these kernels take (and *rc holds the launch
// status), 0 otherwise (the caller falls back to gemm_x3_kernel).  force: 0 = choose; 1..3 = k16 kernel with TN = force;
// 8 = the 8-wave 256 x 128 kernel (tuning hook LVAE_X3V2_TN).  Every choice gives the same bits.
int lvae_gemm_x3v2_try(const lvae_gemm_desc* d, hipStream_t st, int force, int* rc) {
    const bool conv3 = d->a_mode == LVAE_A_CONV3;
    if (d->prec != 2 || (d->a_mode != LVAE_A_PLAIN && !conv3) || (d->K & 31) || d->ldw != d->K) return 0;
    if (!conv3 && (d->lda0 & 3)) return 0;
    if (conv3 && ((d->K0 & 15) || d->K != 9 * d->K0 || d->K1 != 0 || d->H <= 0 || d->W <= 0 || (long)d->M * d->K0 * 4 > 0x7ffffff0L)) return 0;
    const bool cat = d->K1 != 0;                                   // [A0 | A1]: k16 kernels only, stage-aligned split
    if (cat && (!d->A1 || (d->K0 & 15) || (d->lda1 & 3) || (long)256 * d->lda1 * 4 > 0x7fffffffL)) return 0;
    const int S = d->ksplit > 1 ? d->ksplit : 1;
    if (S > 1 && (d->K % (32 * S))) return 0;
    const int M = d->M, N = d->N, K = d->K / S;
    int sel = force;
    if ((S > 1 || cat || conv3) && sel == 8) sel = 0;
    if (sel <= 0) {
        // k16 kernels: rounds of 128 x 64c tiles over 2 x 256 workgroup slots x per-tile work / measured relative efficiency
        double best = 1e300;
        const double eff[4] = {0, 0.70, 1.00, 1.03};
        for (int c = 1; c <= 3; ++c) {
            const long tiles = (long)((M + 127) / 128) * ((N + 64 * c - 1) / (64 * c)) * S;
            const long rounds = (tiles + 511) / 512;
            const double cost = rounds * (128.0 * 64 * c) * (K + 96.0) / eff[c];
            if (cost < best) { best = cost; sel = c; }
        }
        // 8-wave 256 x 128 tiles, one per CU: better inside a tile (70 % vs 55 % MFMA-busy) but whole rounds of 256 tiles; taken when
        // the problem is one well-filled round and long enough to amortise the un-overlapped prologue / epilogue
        const long t8 = (long)((M + 255) / 256) * ((N + 127) / 128);
        if (S == 1 && !cat && !conv3 && t8 >= 176 && t8 <= 256 && K >= 512 && N % 128 == 0) sel = 8;
    }
    if (sel == 8) {
        *rc = d->a_gelu ? launch_w8<true>(d, st) : launch_w8<false>(d, st);
    } else if (conv3) {
        if (d->a_gelu) *rc = sel == 1 ? launch_k16<1, true, LVAE_A_CONV3>(d, st) : sel == 2 ? launch_k16<2, true, LVAE_A_CONV3>(d, st)
                                                                               : launch_k16<3, true, LVAE_A_CONV3>(d, st);
        else *rc = sel == 1 ? launch_k16<1, false, LVAE_A_CONV3>(d, st) : sel == 2 ? launch_k16<2, false, LVAE_A_CONV3>(d, st)
                                                                        : launch_k16<3, false, LVAE_A_CONV3>(d, st);
    } else if (d->a_gelu) {
        *rc = sel == 1 ? launch_k16<1, true, LVAE_A_PLAIN>(d, st) : sel == 2 ? launch_k16<2, true, LVAE_A_PLAIN>(d, st)
                                                                  : launch_k16<3, true, LVAE_A_PLAIN>(d, st);
    } else {
        *rc = sel == 1 ? launch_k16<1, false, LVAE_A_PLAIN>(d, st) : sel == 2 ? launch_k16<2, false, LVAE_A_PLAIN>(d, st)
                                                                   : launch_k16<3, false, LVAE_A_PLAIN>(d, st);
    }
    return 1;
}
