// gemm_h2.hip -- "f16x2" split-MFMA GEMMs (prec 4): fp32-class accuracy from THREE fp16 MFMAs per product step, half the matrix-pipe
// time of the 6-term bf16x3 arithmetic (gemm_x3v2.hip) for the same channel-mixing GEMMs: timm Mlp fc1/fc2 and the 1x1 / 3x3 convs
// (lvae/models/common.py:131-132,154; qarv/model.py:36,38,39).
//
// Arithmetic.  Every fp32 operand x is split into two fp16 terms: hi = f16(x) (RNE, 11 significant bits) and lo' = f16((x - hi) * 2^11)
// (the residual is exact in fp32; scaled by 2^11 it has hi's exponent range, so it keeps 11 more bits wherever x itself is a normal
// fp16 magnitude): hi + lo' * 2^-11 == x to 2^-24 |x| (one bit short of fp32).  A product step keeps three cross terms on
// v_mfma_f32_32x32x16_f16 in TWO fp32 accumulators:   H += a_hi * w_hi      X += a_hi * w_lo' + a_lo' * w_hi      result = H + 2^-11 * X.
// Dropped: a_lo * w_lo <= 2^-22 |a w| worst case, 2^-24.8 rms.  The matrix pipe forms every fp16 product exactly and sums the 16
// products of an instruction without intermediate rounding; fp16 subnormal inputs are honoured (tools/ubench/mfma_f16_probe.hip, MI355X).
// Measured on the reference goldens by CPU emulation of the split alone (tools/split_error_study.py): rms deviation of the prior /
// posterior means from exact arithmetic 9.9e-7 (bf16x3: 5.7e-7; the REFERENCE's own fp32 accumulation: 2.6e-6), i.e. the expected
// number of rounding flips against the reference rises by ~3 %.  Range: |x| must stay below 65504 (fp16); weights are checked when they
// are packed, activations of this network are O(1..100) (LayerNorm-ed blocks).  An activation >= 65520 becomes inf in its hi term, and an
// inf / NaN stays one through every later MFMA sum, GELU, residual add and LayerNorm until it reaches one of the codec's sinks -- a prior
// parameter, a posterior mean or the reconstruction -- where lvae_prior_index_f32 / lvae_quantize_f32 / the ST_IMAGE store OR a bit into
// the plan's status word (include/lvae_hip.h "status word"); lvae_encode_blocks / lvae_decode_blocks read the word per latent block and
// the Python host raises lvae.NonFiniteError naming set_gemm_precision('bf16x3') before any byte string or image is returned
// (tests/test_gpu_overflow.py).
//
// Kernel.  Same software pipeline as gemm_x3k16_kernel (gemm_x3v2.hip: fenced filler slices in the MFMA shadows of the same wave,
// double-buffered LDS with 16-deep stages, buffer loads with hardware range checks two stages ahead, k16-interleaved weights,
// XCD-aware tile order, fused concat / 3x3-tap gather / GELU-on-load, split-K), with two planes instead of three:
// LDS rows are 2 x 32 B + 16 pad = 80 B (16-lane ds_read_b128 groups and the 16-/8-lane write groups still cover distinct banks:
// 20 r mod 64 is a permutation of the multiples of 4 for r = 0..15), a stage of one W row is 64 contiguous bytes, and a stage holds
// 6*TN MFMAs per wave instead of 12*TN.  Per accumulator the MFMA sequence is fixed (k16 steps ascending; X: a_lo'*w_hi then
// a_hi*w_lo'), so results do not depend on M, batch, tile shape (TN) or split-K workgroup placement.
#include "gemm_common.h"

#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define LVAE_FENCE() __builtin_amdgcn_sched_barrier(0)

// TN = 1 or 2 (128 x 64 / 128 x 128 tiles): the two accumulator sets of a 128 x 192 tile (192 registers) do not fit the 256 unified
// registers a wave has at two workgroups per CU.
template <int TN, bool AGELU, int AMODE>
__global__ __launch_bounds__(256, (TN == 1 ? 3 : 2)) void gemm_h2_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    using C = Cfg<2, 2, 2, TN, 1, 32>;
    constexpr int ROWB = 80, ROWS = 128 + 64 * TN, STAGE = ROWS * ROWB;
    constexpr int NW = TN;                                         // 16-B W chunks per thread and stage: 64*TN rows x 4 chunks / 256
    constexpr int NS = 3 * TN;                                     // filler slices per stage (one per MFMA pair)
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

    // split-K (gridDim.y slices): this workgroup covers k16 stages [q0, q0 + nq)
    const int nq = d.K / 16 / (int)gridDim.y, q0 = (int)blockIdx.y * nq;
    const int rows_a = (d.M - m0) < C::BM ? (d.M - m0) : C::BM;
    const float* a0b = d.A0 + (long)m0 * d.lda0;
    const float* a1p = d.A1 ? d.A1 : d.A0;
    const long lda1 = d.A1 ? d.lda1 : d.lda0;
    const float* a1b = a1p + (long)m0 * lda1;
    const int n0rec = rows_a * d.lda0 * 4, n1rec = rows_a * (int)lda1 * 4;
    const int qsplit = d.K0 / 16;                                 // first stage that reads A1 (fused torch.cat, qarv/model.py:66-67)
    const int rows_w = (d.N - n0) < C::BN ? (d.N - n0) : C::BN;
    const long wrow_b = (long)4 * d.K;                            // bytes per W row: [K/16][2 planes][16] halves
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(d.Wt16 + (long)n0 * 2 * d.K + q0 * 32), 0, (int)(rows_w * wrow_b) - q0 * 64, 0x00020000);
    int a_voff[2], a_voff1[2], a_st[2], w_voff[NW], w_st[NW];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = tid + 256 * j, row = perm(c >> 2), ak4 = c & 3;
        a_voff[j] = (row * d.lda0 + ak4 * 4) * 4;
        a_voff1[j] = (row * (int)lda1 + ak4 * 4) * 4;
        a_st[j] = row * ROWB + ak4 * 8;
#ifdef LVAE_EXP_H2_FULLLINE        // experiment (WRONG RESULTS): a wave's load instruction covers 8 rows x 128 B (whole cache lines)
        a_voff[j] = ((c >> 3) * d.lda0 + (c & 7) * 4) * 4;
#endif
    }
    // 3x3-tap gather (implicit GEMM over an NHWC map, K = 9*Cin, tap-major): a stage of 16 channels lies inside one tap; the tap is a
    // uniform offset added to the row's pixel address, a tap outside the image an out-of-range address = a hardware zero.
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
    // 2x2 / stride-2 patch gather (patch_downsample, common.py:29-30; K = 4*Cin ordered (i, j, ci)): output row m = (bh, w) reads the
    // input pixels (2 bh + i, 2 w + j); the two pixels of one i are adjacent in NHWC, so k -> i = k / (2 Cin) selects a uniform row
    // offset and the rest is contiguous.  A stage of 16 channels lies inside one i (Cin % 8 == 0).
    if (AMODE == LVAE_A_PATCH2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = tid + 256 * j, m = m0 + perm(c >> 2), ak4 = c & 3;
            const int w = m % d.W, bh = m / d.W;
            a_voff[j] = m < d.M ? (int)((((long)bh * 2 * (2L * d.W) + 2L * w) * d.K0 + ak4 * 4) * 4) : 0x7fffffff;
        }
    }
    const int a_all_rec = (AMODE == LVAE_A_CONV3) ? (int)((long)d.M * d.K0 * 4) : (AMODE == LVAE_A_PATCH2) ? (int)((long)d.M * 4 * d.K0 * 4) : 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int c = tid + 256 * j, row = perm(c >> 2), piece = c & 3;
        w_voff[j] = row * (int)wrow_b + piece * 16;
        w_st[j] = (128 + row) * ROWB + piece * 16;
#ifdef LVAE_EXP_H2_FULLLINE
        w_voff[j] = (c >> 3) * (int)wrow_b + (c & 7) * 16;
#endif
    }
    const int a_fr = (wave_m * 64 + li) * ROWB + 16 * lh;
    const int b_fr = (128 + wave_n * TN * 32 + li) * ROWB + 16 * lh;

    f32x16 accH[2][TN], accX[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accH[a][b][r] = 0.f; accX[a][b][r] = 0.f; }

    u32x4 ra[2][2], rb[2][NW];        // [stage parity][chunk]
    u32x2 sa[2];
    auto load_a = [&](int par, int j, int q) {      // the source is chosen per stage with scalar selects (a branch would cut the
        const int qg = q0 + q;                        // fenced MFMA / filler stream into basic blocks)
        if (AMODE == LVAE_A_CONV3) {
            const int kq = qg * 16, tap = kq / d.K0, kk = kq - tap * d.K0;              // uniform
            int toff = (((tap / 3 - 1) * d.W + (tap % 3 - 1)) * d.K0 + kk) * 4;
            asm volatile("" : "+s"(toff));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.A0, 0, a_all_rec, 0x00020000);
            int vo = a_voff[j] + toff;
            vo = ((tapok[j] >> tap) & 1) ? vo : 0x7fffffff;
            ra[par][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
            return;
        }
        if (AMODE == LVAE_A_PATCH2) {
            const int kq = qg * 16, seg = 2 * d.K0, i = kq / seg, kk = kq - i * seg;     // uniform
            int toff = (i * 2 * d.W * d.K0 + kk) * 4;
            asm volatile("" : "+s"(toff));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.A0, 0, a_all_rec, 0x00020000);
            ra[par][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_voff[j], toff, 0);
            return;
        }
        const bool second = qg >= qsplit;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(second ? a1b : a0b), 0, second ? n1rec : n0rec, 0x00020000);
        ra[par][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, second ? a_voff1[j] : a_voff[j], (second ? qg - qsplit : qg) * 64, 0);
    };
    auto load_w = [&](int par, int j, int q) { rb[par][j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, w_voff[j], q * 64, 0); };
    auto split_half = [&](int par, int j, int h) {
        float x0 = __uint_as_float(ra[par][j][2 * h]), x1 = __uint_as_float(ra[par][j][2 * h + 1]);
        if (AGELU) { x0 = gelu_erf(x0); x1 = gelu_erf(x1); }
        unsigned hi, lo;
        split_pair_h2(x0, x1, hi, lo);
        asm volatile("" : "+v"(hi), "+v"(lo));       // keeps the split in this slice (LLVM would sink it to the ds_write)
        sa[0][h] = hi; sa[1][h] = lo;
    };
    auto store_a = [&](char* st, int j) {
#pragma unroll
        for (int p = 0; p < 2; ++p) *(u32x2*)(st + a_st[j] + p * 32) = sa[p];
    };
    auto store_w = [&](char* st, int par, int j) { *(u32x4*)(st + w_st[j]) = rb[par][j]; };
    // load order of one register set: A0, W0, W1, A1 -- the same in the prologue and in the loop (vmcnt bookkeeping)
    auto load_set = [&](int par, int q) {
        load_a(par, 0, q);
        load_w(par, 0, q);
        if (NW > 1) load_w(par, 1, q);
        load_a(par, 1, q);
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

    f16x8 af[2][2], bf[2][2];         // af[a][plane]; bf[ping-pong][plane]
    // one k16 stage: compute on stage PAR, write tile q+1 from register set PAR^1 into the other stage, reload that set with q+3
    auto body = [&](auto par_tag, int q) {
        constexpr int PAR = decltype(par_tag)::value, OTH = PAR ^ 1;
        const int q3 = q + 3 < nq ? q + 3 : nq - 1;
        const char* cur = lds + PAR * STAGE;
        char* nxt = lds + OTH * STAGE;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int p = 0; p < 2; ++p) af[a][p] = *(const f16x8*)(cur + a_fr + a * 32 * ROWB + 32 * p);
#pragma unroll
        for (int p = 0; p < 2; ++p) bf[0][p] = *(const f16x8*)(cur + b_fr + 32 * p);
        LVAE_FENCE();
#pragma unroll
        for (int g = 0; g < TN; ++g) {
            const int bb = g & 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                // cross terms first (they are 2^-11 of the main term and have their own accumulator), smallest-first inside X
                if (j == 0) {
                    accX[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][1], bf[bb][0], accX[0][g], 0, 0, 0);
                    accX[1][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][1], bf[bb][0], accX[1][g], 0, 0, 0);
                } else if (j == 1) {
                    accX[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][0], bf[bb][1], accX[0][g], 0, 0, 0);
                    accX[1][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][0], bf[bb][1], accX[1][g], 0, 0, 0);
                } else {
                    accH[0][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][0], bf[bb][0], accH[0][g], 0, 0, 0);
                    accH[1][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][0], bf[bb][0], accH[1][g], 0, 0, 0);
                }
                const int S = g * 3 + j;                                  // filler slice index within the stage
                if (j == 0 && g + 1 < TN) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) bf[bb ^ 1][p] = *(const f16x8*)(cur + b_fr + (g + 1) * 32 * ROWB + 32 * p);
                }
                if (TN == 2) {
                    if (S == 0) split_half(OTH, 0, 0);
                    if (S == 1) { split_half(OTH, 0, 1); store_w(nxt, OTH, 0); }
                    if (S == 2) { store_a(nxt, 0); load_a(OTH, 0, q3); load_w(OTH, 0, q3); }
                    if (S == 3) { split_half(OTH, 1, 0); store_w(nxt, OTH, 1); }
                    if (S == 4) { split_half(OTH, 1, 1); load_w(OTH, 1, q3); }
                    if (S == 5) { store_a(nxt, 1); load_a(OTH, 1, q3); }
                } else {
                    if (S == 0) { split_half(OTH, 0, 0); store_w(nxt, OTH, 0); }
                    if (S == 1) { split_half(OTH, 0, 1); store_a(nxt, 0); load_a(OTH, 0, q3); load_w(OTH, 0, q3); }
                    if (S == 2) { split_half(OTH, 1, 0); split_half(OTH, 1, 1); store_a(nxt, 1); load_a(OTH, 1, q3); }
                }
                LVAE_FENCE();
            }
        }
        __syncthreads();
    };
    static_assert(NS == 3 * TN && TN <= 2, "slices");
    for (int q = 0; q < nq; q += 2) {
        body(std::integral_constant<int, 0>{}, q);
        body(std::integral_constant<int, 1>{}, q + 1);
    }
    // result = H + 2^-11 * X (one fma per element, exact scaling)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) accH[a][b][r] = __builtin_fmaf(accX[a][b][r], 1.0f / 2048.0f, accH[a][b][r]);
    // (gemm_common.h's straight-line epilogues were measured here too -- profiles/r05_bench_ab_gemm_h2_straight_line_epilogue.txt: no gain
    //  on the bench, this kernel's launches are bound by their operand conversion -- and are not taken)
    gemm_finish<C>(d, accH, m0, n0, wave_m, wave_n, li, lh, (void*)smem, t);
}

template <int TN, bool AGELU, int AMODE>
int launch_h2(const lvae_gemm_desc* d, hipStream_t st) {
    constexpr int BN = 64 * TN, LDS = 2 * (128 + BN) * 80;
    const int tiles_m = (d->M + 127) / 128, tiles_n = (d->N + BN - 1) / BN, n_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_h2_kernel<TN, AGELU, AMODE>), dim3(n_tiles, d->ksplit > 1 ? d->ksplit : 1), dim3(256), LDS, st, *d, tiles_n, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace

// Entry point for gemm_f32.hip's dispatcher (prec 4).  Returns 1 when the problem is one this kernel takes (*rc = launch status), 0
// otherwise -- the host (lvae/engine.py: h2_eligible) only asks for prec 4 where it is, so 0 is an argument error upstream.
// force (d.cfg): 0 = choose kernel and tile width; 1..2 = this kernel with that TN; 3 = gemm_h2n wherever it applies.  Every choice gives
// the same bits.
int lvae_gemm_h2n_try(const lvae_gemm_desc* d, hipStream_t st, int force, int* rc);          // gemm_h2n.hip: narrow N over a large M

int lvae_gemm_h2_try(const lvae_gemm_desc* d, hipStream_t st, int force, int* rc) {
    if ((force == 0 || force == 3) && lvae_gemm_h2n_try(d, st, force == 3, rc)) return 1;    // force 3: wherever that kernel can
    const bool conv3 = d->a_mode == LVAE_A_CONV3, patch2 = d->a_mode == LVAE_A_PATCH2;
    if (d->prec != 4 || (d->a_mode != LVAE_A_PLAIN && !conv3 && !patch2) || (d->K & 31) || d->ldw != d->K) return 0;
    if (patch2 && ((d->K0 & 7) || d->K != 4 * d->K0 || d->K1 != 0 || d->H <= 0 || d->W <= 0 || d->a_gelu || (long)d->M * 4 * d->K0 * 4 > 0x7ffffff0L))
        return 0;
    if (!conv3 && !patch2 && ((d->lda0 & 3) || d->K0 + d->K1 != d->K)) return 0;
    if (conv3 && ((d->K0 & 15) || d->K != 9 * d->K0 || d->K1 != 0 || d->H <= 0 || d->W <= 0 || (long)d->M * d->K0 * 4 > 0x7ffffff0L)) return 0;
    const bool cat = d->K1 != 0 && !patch2;
    if (cat && (!d->A1 || (d->K0 & 15) || (d->lda1 & 3) || (long)256 * d->lda1 * 4 > 0x7fffffffL)) return 0;
    if ((long)128 * d->lda0 * 4 > 0x7fffffffL || (long)192 * 4 * d->K > 0x7fffffffL) return 0;
    const int S = d->ksplit > 1 ? d->ksplit : 1;
    if (S > 1 && (d->K % (32 * S))) return 0;
    const int M = d->M, N = d->N, K = d->K / S;
    int sel = force;
    if (sel <= 0 || sel > 2) {
        // rounds of 128 x 64c tiles over 2 x 256 workgroup slots x per-tile work / relative efficiency of the tile width
        double best = 1e300;
        const double eff[3] = {0, 0.70, 1.00};
        for (int c = 1; c <= 2; ++c) {
            const long tiles = (long)((M + 127) / 128) * ((N + 64 * c - 1) / (64 * c)) * S;
            const long rounds = (tiles + 511) / 512;
            const double cost = rounds * (128.0 * 64 * c) * (K + 96.0) / eff[c];
            if (cost < best) { best = cost; sel = c; }
        }
    }
#define LVAE_H2_LAUNCH(G, AM) (sel == 1 ? launch_h2<1, G, AM>(d, st) : launch_h2<2, G, AM>(d, st))
    if (patch2) *rc = LVAE_H2_LAUNCH(false, LVAE_A_PATCH2);
    else if (conv3) *rc = d->a_gelu ? LVAE_H2_LAUNCH(true, LVAE_A_CONV3) : LVAE_H2_LAUNCH(false, LVAE_A_CONV3);
    else *rc = d->a_gelu ? LVAE_H2_LAUNCH(true, LVAE_A_PLAIN) : LVAE_H2_LAUNCH(false, LVAE_A_PLAIN);
#undef LVAE_H2_LAUNCH
    return 1;
}
