// gemm_f32.hip -- the dense channel-mixing kernel of the QARV / QRes-VAE hot path for gfx950 (MI355X).
//
//   out[m][n] = epilogue( sum_k A[m][k] * Wt[n][k] + bias[n] )      fp32 in, fp32 accumulate
//
// Replaces (reference file:line):  timm Mlp fc1/fc2 inside ConvNeXtBlockAdaLN (lvae/models/common.py:131-132,154;
// 87% of the path's FLOPs), conv1x1 post_merge / z_proj / prior (lvae/models/qarv/model.py:36,38,39), the conv of
// patch_upsample + PixelShuffle (common.py:33-38), conv k=s patch_downsample (common.py:29-30) and the 3x3
// posterior head (qarv/model.py:37).
//
// Design (MI355X-first, not a translation of cuBLAS/cuDNN calls):
//  * rows m are NHWC pixels, so A is K-contiguous and so are PyTorch's [N][K] weights: both operands stream with
//    16-B per-lane loads, no transposes anywhere;
//  * v_mfma_f32_32x32x2_f32 (exact fp32 = an fmaf chain, 157 TF/s peak): 256 threads = 4 wave64, each wave owns a
//    (TM*32)x(TN*32) sub-tile; per 8-deep k-substep a lane reads ONE ds_read_b128 per operand block and feeds
//    4 MFMAs per (m-block, n-block) pair -> LDS traffic is ~1/16 of MFMA time;
//  * LDS tiles [rows][32+4] floats: the +4 pad makes the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots
//    (rows*36 mod 64 are distinct multiples of 4 for rows distinct mod 16) -> conflict-free;
//  * register-staged double buffering, one barrier per 32-deep k-tile: global loads for tile t+1 are issued before
//    the MFMAs of tile t and written to the other LDS buffer after them;
//  * fused torch.cat (two A sources), fused 2x2 patch gather / 3x3 tap gather (implicit GEMM over an NHWC map),
//    fused bias / exact-erf GELU / layer-scale / residual / PixelShuffle / output clamp epilogues;
//  * XCD-aware tile order: consecutive tiles (n fastest) are kept on one XCD so an A row-panel is fetched from HBM
//    once and re-read from that XCD's L2 by the other n-tiles;
//  * the per-element accumulation order depends only on K (k-tile 32, substep 8, pairs (j, j+4)), never on the
//    tile configuration or on M: a batch of 8 images gives bit-identical rows to 8 single-image calls, and the
//    encoder and decoder (different M) derive bit-identical priors.  No split-K, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/lvae_hip.h"
#include "device_math.h"

#ifdef LVAE_GEMM_TRACE
extern "C" __device__ long* lvae_trace_buf;          // [16 k-tiles][8 stamps], filled by one wave of one block
// stamps go to LDS (beyond the tiles) so that they do not sit on the vmcnt queue the loader waits on; dumped at the end
#define TRACE_STAMP(slot) do { if (tracing && kt < 16) ((long*)(smem + C::NBUF * (C::BM + C::BN) * C::LDT))[kt * 8 + (slot)] = clock64(); } while (0)
#else
#define TRACE_STAMP(slot) do {} while (0)
#endif

#include "gemm_common.h"

namespace {

struct RowInfo {      // per staged A row of this thread (rows beyond M are clamped to M-1: loaded, never stored)
    const float* p0;  // PLAIN: A0 + m*lda0;      PATCH2/CONV3: address of the row's own pixel / first tap
    const float* p1;  // PLAIN: A1 + m*lda1 - K0  (so that p1 + k is the element for k >= K0)
    int h, w;         // CONV3: pixel coordinates
};

template <int AMODE>
__device__ __forceinline__ RowInfo row_info(const lvae_gemm_desc& d, int m) {
    RowInfo ri;
    ri.h = ri.w = 0;
    m = m < d.M ? m : d.M - 1;
    ri.p1 = nullptr;
    if (AMODE == LVAE_A_PLAIN) {
        ri.p0 = d.A0 + (long)m * d.lda0;
        ri.p1 = d.A1 ? d.A1 + (long)m * d.lda1 - d.K0 : ri.p0;
    } else if (AMODE == LVAE_A_PATCH2) {
        const int w = m % d.W, bh = m / d.W;          // bh = b*H + h on the OUTPUT grid; input rows 2*bh, 2*bh+1
        ri.p0 = d.A0 + ((long)bh * 2 * (2L * d.W) + 2L * w) * d.K0;
    } else {
        const int w = m % d.W, bh = m / d.W;
        ri.w = w;
        ri.h = bh % d.H;
        ri.p0 = d.A0 + (long)m * d.K0;
    }
    return ri;
}

// Branch-free operand fetch: ONE unconditional 16-B load per staged element from an always-valid address; elements
// that must read as zero (k beyond K, 3x3 taps outside the image) are flagged in `ok` and zeroed when the registers
// are written to LDS.  (Divergent "if (valid) load" forms make hipcc guard every load with s_waitcnt vmcnt(0) --
// WAW on the destination registers -- which serialises the 8 loads of a k-tile: measured 1200-3000 cycles per tile.)
template <int AMODE>
__device__ __forceinline__ f32x4 load_a(const lvae_gemm_desc& d, const RowInfo& ri, int k, bool& ok) {
    ok = k < d.K;
    const int kc = ok ? k : d.K - 4;
    if (AMODE == LVAE_A_PLAIN) {
        const float* p = (kc < d.K0) ? ri.p0 : ri.p1;
        return *(const f32x4*)(p + kc);
    } else if (AMODE == LVAE_A_PATCH2) {
        const int seg = 2 * d.K0;                 // one input row of the 2x2 patch: 2 pixels x Cin
        const int s = kc / seg, kk = kc - s * seg;
        return *(const f32x4*)(ri.p0 + (long)s * (2L * d.W) * d.K0 + kk);
    } else {
        const int s = kc / d.K0, kk = kc - s * d.K0;
        const int di = s / 3 - 1, dj = s - (s / 3) * 3 - 1;
        const int hh = ri.h + di, ww = ri.w + dj;
        const bool in = (hh >= 0) && (hh < d.H) && (ww >= 0) && (ww < d.W);
        ok = ok && in;
        const long tap = in ? ((long)di * d.W + dj) * d.K0 : 0;
        return *(const f32x4*)(ri.p0 + tap + kk);
    }
}

// ---------------------------------------------------------------- epilogue (shared by the f32 and bf16 main loops)

template <class C, int AMODE>
__global__ __launch_bounds__(C::NT, 2) void gemm_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [NBUF][BM][LDT]
    constexpr int BK = C::BK, LDT = C::LDT;
    float* Ws = smem + C::NBUF * C::BM * LDT;  // [NBUF][BN][LDT]

#ifdef LVAE_GEMM_TRACE
    const long t_entry = clock64();
#endif
    // XCD-aware bijective remap (block b runs on XCD b%8): each XCD gets a contiguous chunk of the tile list
    int t;
    {
        const int b = blockIdx.x, q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * C::BM, n0 = tn * C::BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / C::WGN, wave_n = wave % C::WGN;
    const int li = lane & 31, lh = lane >> 5;

    // staging assignment: thread -> rows (srow + 32*i), 16-B column sk4
    const int srow = tid / C::CPR, sk4 = tid % C::CPR;
    RowInfo ri[C::NA];
#pragma unroll
    for (int i = 0; i < C::NA; ++i) ri[i] = row_info<AMODE>(d, m0 + srow + C::RP * i);
    const float* wrow[C::NB];
#pragma unroll
    for (int i = 0; i < C::NB; ++i) {
        const int n = n0 + srow + C::RP * i;
        wrow[i] = d.Wt + (long)(n < d.N ? n : d.N - 1) * d.ldw;     // columns beyond N: loaded, never stored
    }

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 ra[C::NA], rb[C::NB];
    const int nk = ((d.K + BK - 1) / BK) / (int)gridDim.y, kt0 = (int)blockIdx.y * nk;      // split-K: this slice's k-tiles
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    unsigned okmask = 0;      // bit i: ra[i] is real data; bit 16: the W chunk is real data (k < K)
    // operand loads of k-tile kt, slice `part` of `nparts` (nparts = 1: everything).  In the main loop the loads are
    // issued in BK/8 slices, one behind each 8-deep MFMA substep: all 8 waves of a CU issuing their whole tile at once
    // right after the barrier saturated the CU's vector-memory issue (in-kernel timeline: 2-5k cycles with NO wave in
    // its MFMA phase, 15-25% of a k-tile); spread out, each load issues in the shadow of the preceding MFMAs.
    auto gload = [&](int kt, int part, int nparts) {
        const int k = (kt0 + kt) * BK + sk4 * 4;
        if (part == 0) okmask = 0;
#pragma unroll
        for (int i = 0; i < C::NA; ++i) {
            if (i % nparts != part) continue;
            bool ok;
            ra[i] = load_a<AMODE>(d, ri[i], k, ok);
            okmask |= (ok ? 1u : 0u) << i;
        }
        const bool wok = k < d.K;
        const int kc = wok ? k : d.K - 4;
        if (part == 0) okmask |= (wok ? 1u : 0u) << 16;
#pragma unroll
        for (int i = 0; i < C::NB; ++i)
            if ((i + C::NA) % nparts == part) rb[i] = *(const f32x4*)(wrow[i] + kc);
    };
    auto lstore = [&](int buf) {
        float* a = As + buf * C::BM * LDT + srow * LDT + sk4 * 4;
        float* w = Ws + buf * C::BN * LDT + srow * LDT + sk4 * 4;
#pragma unroll
        for (int i = 0; i < C::NA; ++i) {
            f32x4 v = ((okmask >> i) & 1u) ? ra[i] : zero4;
            if (d.a_gelu) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            *(f32x4*)(a + C::RP * i * LDT) = v;
        }
#pragma unroll
        for (int i = 0; i < C::NB; ++i)
            if (C::BN % C::RP == 0 || srow + C::RP * i < C::BN)
                *(f32x4*)(w + C::RP * i * LDT) = ((okmask >> 16) & 1u) ? rb[i] : zero4;
    };

    gload(0, 0, 1);
    lstore(0);
    __syncthreads();

#ifdef LVAE_GEMM_TRACE
    const bool tracing = (blockIdx.x == gridDim.x / 2 + 3) && tid == 0;
    const long t_prologue = clock64();
#endif
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = (C::NBUF == 2) ? (kt & 1) : 0;
        TRACE_STAMP(0);
        const bool more = kt + 1 < nk;
        // small wave tiles (<= 2 MFMA blocks, 1-2k cycles of MFMA per k-tile) are latency-bound: issue their loads first
        constexpr bool kSpreadLoads = C::TM * C::TN >= 4;
        // all slices go out in the FIRST half of the substeps so that the remaining MFMAs still cover their latency
        constexpr int kLoadSlices = (BK / 8 >= 4) ? BK / 16 : 1;
#ifndef LVAE_GEMM_NOLOAD
        if (!kSpreadLoads && more) gload(kt + 1, 0, 1);
#endif
        TRACE_STAMP(1);
        const float* a_base = As + cur * C::BM * LDT + (wave_m * C::TM * 32 + li) * LDT + 4 * lh;
        const float* b_base = Ws + cur * C::BN * LDT + (wave_n * C::TN * 32 + li) * LDT + 4 * lh;
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            f32x4 af[C::TM], bf[C::TN];
#pragma unroll
            for (int a = 0; a < C::TM; ++a) af[a] = *(const f32x4*)(a_base + a * 32 * LDT + 8 * s);
#pragma unroll
            for (int b = 0; b < C::TN; ++b) bf[b] = *(const f32x4*)(b_base + b * 32 * LDT + 8 * s);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < C::TM; ++a)
#pragma unroll
                    for (int b = 0; b < C::TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][j], bf[b][j], acc[a][b], 0, 0, 0);
#ifndef LVAE_GEMM_NOLOAD
            if (kSpreadLoads) {
                __builtin_amdgcn_sched_barrier(0);      // keep this slice of loads BEHIND the substep's MFMAs
                if (more && s < kLoadSlices) gload(kt + 1, s, kLoadSlices);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        TRACE_STAMP(2);
        if (C::NBUF == 2) {
            if (kt + 1 < nk) lstore(cur ^ 1);
            TRACE_STAMP(3);
            __syncthreads();
        } else if (kt + 1 < nk) {
            __syncthreads();                    // every wave has finished reading the (only) stage
            lstore(0);
            TRACE_STAMP(3);
            __syncthreads();
        }
        TRACE_STAMP(4);
    }

#ifdef LVAE_GEMM_TRACE
    const long t_loop = clock64();
    if (tracing) for (int i = 0; i < 120; ++i) lvae_trace_buf[i] = ((long*)(smem + C::NBUF * (C::BM + C::BN) * C::LDT))[i];
#define TRACE_END() do { if (tracing) { __builtin_amdgcn_s_waitcnt(0); lvae_trace_buf[120] = t_entry; lvae_trace_buf[121] = t_prologue; \
                                         lvae_trace_buf[122] = t_loop; lvae_trace_buf[123] = clock64(); } } while (0)
#else
#define TRACE_END() do {} while (0)
#endif
    gemm_finish<C>(d, acc, m0, n0, wave_m, wave_n, li, lh, (void*)smem, t);
    TRACE_END();
}

// ---------------------------------------------------------------- reduced-precision variant (BASELINE config 5)
// Same tiles, loaders and epilogues, but the operands are rounded to bf16 (RNE) on their way into LDS and multiplied on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate): a k-tile is 64 deep (one 144-B LDS row = 64 bf16 + 8 pad, same 36-dword
// pitch and conflict-free ds_read_b128 as the f32 tiles).  Activations stay fp32 in HBM; weights are pre-converted
// (`Wt16`).  MFMA time drops 16x, so this kernel is bound by the operand/result streams.  NOT the parity path: results
// differ from the reference at the 2^-9 relative level (tolerance-checked in tests/test_gpu_bf16.py); encoder and decoder
// stay consistent because both run the same kernels.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <class C, int AMODE>
__global__ __launch_bounds__(C::NT, 2) void gemm_bf16_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KT = 64, LDT = C::LDT;          // k elements per tile; LDS row pitch in dwords (36)
    static_assert(C::BK == 32, "bf16 tiles reuse the 36-dword row pitch");
    char* As = (char*)smem;                                   // [NBUF][BM][144 B]
    char* Ws = (char*)(smem + C::NBUF * C::BM * LDT);         // [NBUF][BN][144 B]
    int t;
    {
        const int b = blockIdx.x, q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * C::BM, n0 = tn * C::BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / C::WGN, wave_n = wave % C::WGN;
    const int li = lane & 31, lh = lane >> 5;

    // A staging: 16 float4 chunks per row; W staging: 8 chunks of 8 bf16 per row
    constexpr int RPA = C::NT / 16, NA = C::BM / RPA, RPW = C::NT / 8, NB = (C::BN + RPW - 1) / RPW;
    static_assert(C::BM % RPA == 0, "A tile rows");
    const int arow = tid >> 4, ak4 = tid & 15;
    const int wrow_i = tid >> 3, wk8 = tid & 7;
    RowInfo ri[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) ri[i] = row_info<AMODE>(d, m0 + arow + RPA * i);
    const unsigned short* wptr[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + wrow_i + RPW * i;
        wptr[i] = d.Wt16 + (long)(n < d.N ? n : d.N - 1) * d.ldw;
    }

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 ra[NA];
    bf16x8 rb[NB];
    unsigned okmask = 0;
    const int nk = ((d.K + KT - 1) / KT) / (int)gridDim.y, kt0 = (int)blockIdx.y * nk;      // split-K: this slice's k-tiles
    auto gload = [&](int kt, int part, int nparts) {
        const int k = (kt0 + kt) * KT + ak4 * 4;
        if (part == 0) okmask = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i % nparts != part) continue;
            bool ok;
            ra[i] = load_a<AMODE>(d, ri[i], k, ok);
            okmask |= (ok ? 1u : 0u) << i;
        }
        const int kw = (kt0 + kt) * KT + wk8 * 8;
        const bool wok = kw < d.K;
        const int kc = wok ? kw : d.K - 8;
        if (part == 0) okmask |= (wok ? 1u : 0u) << 16;
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if ((i + NA) % nparts == part) rb[i] = *(const bf16x8*)(wptr[i] + kc);
    };
    auto lstore = [&](int buf) {
        char* a = As + ((long)buf * C::BM + arow) * (LDT * 4) + ak4 * 8;
        char* w = Ws + ((long)buf * C::BN + wrow_i) * (LDT * 4) + wk8 * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v = ((okmask >> i) & 1u) ? ra[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (d.a_gelu) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            *(bf16x4*)(a + (long)RPA * i * (LDT * 4)) = __builtin_convertvector(v, bf16x4);     // RNE
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (C::BN % RPW == 0 || wrow_i + RPW * i < C::BN) {
                bf16x8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
                *(bf16x8*)(w + (long)RPW * i * (LDT * 4)) = ((okmask >> 16) & 1u) ? rb[i] : z;
            }
    };

    gload(0, 0, 1);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = (C::NBUF == 2) ? (kt & 1) : 0;
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1, 0, 1);            // MFMA phase is short here: loads first, they dominate the tile time
        const char* a_base = As + ((long)cur * C::BM + wave_m * C::TM * 32 + li) * (LDT * 4) + 16 * lh;
        const char* b_base = Ws + ((long)cur * C::BN + wave_n * C::TN * 32 + li) * (LDT * 4) + 16 * lh;
#pragma unroll
        for (int s = 0; s < KT / 16; ++s) {       // one MFMA k-step = 16: lane (i, h) supplies k = 16 s + 8 h .. +7
            bf16x8 af[C::TM], bf[C::TN];
#pragma unroll
            for (int a = 0; a < C::TM; ++a) af[a] = *(const bf16x8*)(a_base + (long)a * 32 * (LDT * 4) + 32 * s);
#pragma unroll
            for (int b = 0; b < C::TN; ++b) bf[b] = *(const bf16x8*)(b_base + (long)b * 32 * (LDT * 4) + 32 * s);
#pragma unroll
            for (int a = 0; a < C::TM; ++a)
#pragma unroll
                for (int b = 0; b < C::TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (C::NBUF == 2) {
            if (more) lstore(cur ^ 1);
            __syncthreads();
        } else if (more) {
            __syncthreads();
            lstore(0);
            __syncthreads();
        }
    }
    gemm_finish<C>(d, acc, m0, n0, wave_m, wave_n, li, lh, (void*)smem, t);
}

// ---------------------------------------------------------------- fp32-accurate split variant ("bf16x3")
// Every fp32 operand is split exactly into three bf16 terms  x = hi + mid + lo  (hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid); residual <= 2^-25 |x|) and the product is formed from the six largest cross terms
//   a*b ~= hi*hi + (hi*mid + mid*hi) + (mid*mid + hi*lo + lo*hi)        (dropped terms <= 2^-24 |a*b|)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 6 MFMAs of 32 cycles per 16-deep step instead of 8 f32 MFMAs of
// 64 cycles -- 2.7x less matrix-pipe time at fp32-class accuracy (bf16 x bf16 products are exact in fp32).  Activations stay
// fp32 in HBM and are split on their way into LDS (3 planes of 64 B per 32-deep row, 208-B pitch: conflict-free
// ds_read_b128); weights are pre-split on the host into three [N][K] bf16 planes (Wt16, plane stride N*ldw).
// Single LDS stage (2 barriers per k-tile) so that every tile shape fits; results do not depend on the tile shape.
template <class C, int AMODE>
__global__ __launch_bounds__(C::NT, 2) void gemm_x3_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KT = 32, ROWB = 208;
    char* As = (char*)smem;                       // [BM][208 B]
    char* Ws = As + C::BM * ROWB;                 // [BN][208 B]
    int t;
    {
        const int b = blockIdx.x, q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * C::BM, n0 = tn * C::BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / C::WGN, wave_n = wave % C::WGN;
    const int li = lane & 31, lh = lane >> 5;

    constexpr int RPA = C::NT / 8, NA = C::BM / RPA, RPW = C::NT / 4, NB = (C::BN + RPW - 1) / RPW;
    static_assert(C::BM % RPA == 0, "A tile rows");
    const int arow = tid >> 3, ak4 = tid & 7;
    const int wrow_i = tid >> 2, wk8 = tid & 3;
    RowInfo ri[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) ri[i] = row_info<AMODE>(d, m0 + arow + RPA * i);
    const unsigned short* wptr[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + wrow_i + RPW * i;
        wptr[i] = d.Wt16 + (long)(n < d.N ? n : d.N - 1) * d.ldw;
    }
    const long plane = (long)d.N * d.ldw;

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 ra[NA];
    bf16x8 rb[3][NB];
    unsigned okmask = 0;
    const int nk = ((d.K + KT - 1) / KT) / (int)gridDim.y, kt0 = (int)blockIdx.y * nk;      // split-K: this slice's k-tiles
    auto gload = [&](int kt) {
        const int k = (kt0 + kt) * KT + ak4 * 4;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            bool ok;
            ra[i] = load_a<AMODE>(d, ri[i], k, ok);
            okmask |= (ok ? 1u : 0u) << i;
        }
        const int kw = (kt0 + kt) * KT + wk8 * 8;
        const bool wok = kw < d.K;
        const int kc = wok ? kw : d.K - 8;
        okmask |= (wok ? 1u : 0u) << 16;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[p][i] = *(const bf16x8*)(wptr[i] + p * plane + kc);
    };
    auto lstore = [&]() {
        char* a = As + (long)arow * ROWB + ak4 * 8;
        char* w = Ws + (long)wrow_i * ROWB + wk8 * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v = ((okmask >> i) & 1u) ? ra[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (d.a_gelu) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            const bf16x4 hi = __builtin_convertvector(v, bf16x4);
            const f32x4 r1 = v - __builtin_convertvector(hi, f32x4);
            const bf16x4 mid = __builtin_convertvector(r1, bf16x4);
            const f32x4 r2 = r1 - __builtin_convertvector(mid, f32x4);
            const bf16x4 lo = __builtin_convertvector(r2, bf16x4);
            char* dst = a + (long)RPA * i * ROWB;
            *(bf16x4*)(dst) = hi;
            *(bf16x4*)(dst + 64) = mid;
            *(bf16x4*)(dst + 128) = lo;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (C::BN % RPW == 0 || wrow_i + RPW * i < C::BN) {
                bf16x8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
                const bool wok = (okmask >> 16) & 1u;
                char* dst = w + (long)RPW * i * ROWB;
#pragma unroll
                for (int p = 0; p < 3; ++p) *(bf16x8*)(dst + 64 * p) = wok ? rb[p][i] : z;
            }
    };

    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
        const char* a_base = As + (long)(wave_m * C::TM * 32 + li) * ROWB + 16 * lh;
        const char* b_base = Ws + (long)(wave_n * C::TN * 32 + li) * ROWB + 16 * lh;
#pragma unroll
        for (int s = 0; s < KT / 16; ++s) {
            bf16x8 af[C::TM][3];
#pragma unroll
            for (int a = 0; a < C::TM; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[a][p] = *(const bf16x8*)(a_base + (long)a * 32 * ROWB + 64 * p + 32 * s);
#pragma unroll
            for (int b = 0; b < C::TN; ++b) {
                bf16x8 bf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[p] = *(const bf16x8*)(b_base + (long)b * 32 * ROWB + 64 * p + 32 * s);
                // smallest cross terms first, (hi, hi) last
#pragma unroll
                for (int a = 0; a < C::TM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[0], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < C::TM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[2], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < C::TM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[1], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < C::TM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[0], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < C::TM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[1], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int a = 0; a < C::TM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[0], acc[a][b], 0, 0, 0);
            }
        }
        if (more) {
            __syncthreads();
            lstore();
            __syncthreads();
        }
    }
    gemm_finish<C>(d, acc, m0, n0, wave_m, wave_n, li, lh, (void*)smem, t);
}

inline int ksp(const lvae_gemm_desc* d) { return d->ksplit > 1 ? d->ksplit : 1; }

template <class C, int AMODE>
int launch_cfg(const lvae_gemm_desc* d, hipStream_t st) {
    const int tiles_m = (d->M + C::BM - 1) / C::BM, tiles_n = (d->N + C::BN - 1) / C::BN;
    const int n_tiles = tiles_m * tiles_n;
    static LdsAttr attr;
    int ae;
    if constexpr (C::BK == 32)
        ae = attr.ensure((const void*)gemm_kernel<C, AMODE>, C::LDS_BYTES, (const void*)gemm_bf16_kernel<C, AMODE>, C::LDS_BYTES,
                         (const void*)gemm_x3_kernel<C, AMODE>, (C::BM + C::BN) * 208);
    else
        ae = attr.ensure((const void*)gemm_kernel<C, AMODE>, C::LDS_BYTES);
    if (ae) return ae;
    if constexpr (C::BK == 32) {
        if (d->prec == 1) {
            hipLaunchKernelGGL((gemm_bf16_kernel<C, AMODE>), dim3(n_tiles, ksp(d)), dim3(C::NT), C::LDS_BYTES, st, *d, tiles_n, n_tiles);
            return (int)hipGetLastError();
        }
        if (d->prec == 2) {
            constexpr int lds_x3 = (C::BM + C::BN) * 208;
            static_assert(lds_x3 <= 160 * 1024, "x3 LDS");
            hipLaunchKernelGGL((gemm_x3_kernel<C, AMODE>), dim3(n_tiles, ksp(d)), dim3(C::NT), lds_x3, st, *d, tiles_n, n_tiles);
            return (int)hipGetLastError();
        }
    }
    hipLaunchKernelGGL((gemm_kernel<C, AMODE>), dim3(n_tiles, ksp(d)), dim3(C::NT), C::LDS_BYTES, st, *d, tiles_n, n_tiles);
    return (int)hipGetLastError();
}

// 4-wave configs (2 workgroups per CU): small / narrow problems
typedef Cfg<2, 2, 2, 2> CfgA;   // 128 x 128, wave 64x64
typedef Cfg<2, 2, 2, 1> CfgB;   // 128 x 64,  wave 64x32
typedef Cfg<4, 1, 1, 1> CfgC;   // 128 x 32,  wave 32x32
typedef Cfg<2, 2, 1, 1> CfgS;   //  64 x 64,  wave 32x32
// 8-wave configs (1 workgroup per CU, 2 waves per SIMD): the large layers.  A CU ingests only ~10 B/clk from L2/HBM
// (MI355X_MICROARCH.md: global_load_dwordx4 ~10 B/cyc/CU) while the fp32 MFMA pipes of one CU retire a 32-deep k-step
// of a BMxBN tile in BM*BN/4 clk, i.e. the operand stream needs 512*(BM+BN)/(BM*BN) B/clk: 8.0 for 128x128 (load-bound,
// measured 72% MFMA-busy), 5.3 for 256x192, 4.0 for 256x256.
typedef Cfg<4, 2, 2, 4> CfgL256;   // 256 x 256, wave 64x128
typedef Cfg<4, 2, 2, 3> CfgL192;   // 256 x 192, wave 64x96
typedef Cfg<8, 1, 1, 7> CfgL224;   // 256 x 224, wave 32x224  (N = 448 = 1.75 x 256)
typedef Cfg<4, 2, 2, 2> CfgL128;   // 256 x 128, wave 64x64
// 4-wave single-LDS-stage configs, 2 workgroups per CU (one wave of each per SIMD): the co-resident workgroup's main
// loop covers this one's prologue / epilogue / barriers -- for the short-K layers (K = 128..768) where a 1-per-CU tile
// spends a third of its life outside the MFMA loop.
typedef Cfg<2, 2, 2, 4, 1> CfgD256;   // 128 x 256, wave 64x128, 55 KB LDS
typedef Cfg<2, 2, 2, 3, 1> CfgD192;   // 128 x 192, wave 64x96,  46 KB LDS
// 64-deep k-tiles for small problems (few workgroups, nothing co-resident to hide memory latency): twice the bytes in
// flight per barrier and half the barriers -- the B=1 / stride-32,64 layers are latency-bound, not MFMA-bound.
typedef Cfg<2, 2, 1, 1, 2, 64> CfgS64;   // 64 x 64, BK 64
typedef Cfg<2, 2, 2, 1, 2, 64> CfgB64;   // 128 x 64, BK 64

constexpr int kCUs = 256;
}  // namespace
// tuning hook (LVAE_GEMM_CFG env var): force a tile configuration id for N > 64; one object for the three translation units below
#ifdef LVAE_GEMM_TU_AMODE
extern int g_force_cfg;
#else
int g_force_cfg = -1;
#endif
namespace {

// Estimated cost (arbitrary units ~ MFMA cycles on the critical CU) of running the problem with a BMxBN tile:
// rounds of tiles over the CUs (workgroup slots) x per-tile work, plus a per-tile fixed cost (prologue + epilogue).
inline double tile_cost(int M, int N, int K, int BM, int BN, int wg_per_cu, double eff) {
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const long slots = (long)kCUs * wg_per_cu;
    const long rounds = (tiles + slots - 1) / slots;
    const double per_tile = (double)BM * BN * (K + 96.0) * wg_per_cu;   // co-resident workgroups share the MFMA pipes
    return rounds * per_tile / eff;
}

template <int AMODE>
int launch_mode(const lvae_gemm_desc* d, hipStream_t st) {
    const int N = d->N, M = d->M * ksp(d), K = d->K / ksp(d);      // split-K: S x the tiles, 1/S the depth
    if (d->cfg <= 0 && g_force_cfg < 0) {
        if (N <= 32 || N == 96) return launch_cfg<CfgC, AMODE>(d, st);
        if (N <= 64) return launch_cfg<CfgB, AMODE>(d, st);
    }
    // candidates: {cost, id}; eff = measured relative MFMA efficiency of the structure
    double best = 1e300;
    int id = 0;
    auto consider = [&](int cid, int BM, int BN, int wg, double eff) {
        const double c = tile_cost(M, N, K, BM, BN, wg, eff);
        if (c < best) { best = c; id = cid; }
    };
    consider(0, 128, 128, 2, 0.72);
    consider(1, 128, 64, 2, 0.60);
    consider(2, 64, 64, 2, 0.50);
    consider(3, 256, 256, 1, 0.90);
    consider(4, 256, 192, 1, 0.88);
    consider(5, 256, 224, 1, 0.85);
    consider(6, 256, 128, 1, 0.85);
    consider(8, 128, 192, 2, 0.92);     // measured: never slower than 256x192, 20% faster at (N=384, K=768)
    consider(7, 128, 256, 2, 0.84);
    if (g_force_cfg >= 0) id = g_force_cfg;
    if (d->cfg > 0) id = d->cfg - 1;
    if (d->prec != 0 && id == 10) id = 2;
    if (d->prec != 0 && id == 11) id = 1;
    if (d->prec == 2 && id == 7) id = 3;          // 128x256 x3 instance spills; 256x256 covers the same shapes
    if (ksp(d) > 1 && id == 10) id = 2;           // split-K slices are counted in 32-deep k-tiles
    if (ksp(d) > 1 && id == 11) id = 1;
    switch (id) {
        case 0: return launch_cfg<CfgA, AMODE>(d, st);
        case 1: return launch_cfg<CfgB, AMODE>(d, st);
        case 2: return launch_cfg<CfgS, AMODE>(d, st);
        case 3: return launch_cfg<CfgL256, AMODE>(d, st);
        case 4: return launch_cfg<CfgL192, AMODE>(d, st);
        case 5: return launch_cfg<CfgL224, AMODE>(d, st);
        case 7: return launch_cfg<CfgD256, AMODE>(d, st);
        case 8: return launch_cfg<CfgD192, AMODE>(d, st);
        case 9: return launch_cfg<CfgC, AMODE>(d, st);
        case 10: return launch_cfg<CfgS64, AMODE>(d, st);
        case 11: return launch_cfg<CfgB64, AMODE>(d, st);
        default: return launch_cfg<CfgL128, AMODE>(d, st);
    }
}

// This source is compiled three times (build_native.py): as is (plain-A instances + every entry point) and, through gemm_f32_patch2.hip /
// gemm_f32_conv3.hip, with LVAE_GEMM_TU_AMODE = 1 / 2 (the 2x2-patch and 3x3-tap gather instances of the 12 tile configurations x 3
// arithmetics) -- one translation unit with all 108 kernel instances took four minutes to compile.
#ifdef LVAE_GEMM_TU_AMODE
}  // namespace
#if LVAE_GEMM_TU_AMODE == 1
int lvae_gemm_launch_patch2(const lvae_gemm_desc* d, hipStream_t st) { return launch_mode<LVAE_A_PATCH2>(d, st); }
#else
int lvae_gemm_launch_conv3(const lvae_gemm_desc* d, hipStream_t st) { return launch_mode<LVAE_A_CONV3>(d, st); }
#endif
#else
// split-K second pass (d.cnt == NULL only): one thread per 16-B chunk of the whole output
__global__ void splitk_reduce_kernel(const lvae_gemm_desc d, int S) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n4 = d.N >> 2;
    if (e >= (long)d.M * n4) return;
    const long m = e / n4;
    splitk_reduce_chunk(d, S, m, (int)(e - m * n4) * 4);
}

}  // namespace

extern "C" int lvae_gemm_num_configs(void) { return 12; }

// The second pass of a parallel split-K GEMM as a launch of its own (csrc/mlp_sk.hip writes the S planes itself): out = epilogue(sum over
// the planes in slice order + bias) -- splitk_reduce_chunk, the function every split-K form ends in.
int lvae_splitk_reduce_launch(const lvae_gemm_desc* d, int S, hipStream_t st) {
    if (!d || S < 2 || (d->N & 3) || (d->ldo & 3) || (d->ldres & 3) || !d->ws || !d->out) return -22;
    const long n = (long)d->M * (d->N >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *d, S);
    return (int)hipGetLastError();
}

int lvae_gemm_x3v2_try(const lvae_gemm_desc* d, hipStream_t st, int force_tn, int* rc);      // gemm_x3v2.hip
int lvae_gemm_h2_try(const lvae_gemm_desc* d, hipStream_t st, int force_tn, int* rc);        // gemm_h2.hip
int lvae_gemm_h2p_try(const lvae_gemm_desc* d, hipStream_t st, int force_tile, int* rc);     // gemm_h2p.hip
int lvae_gemm_h2n_try(const lvae_gemm_desc* d, hipStream_t st, int force, int* rc);          // gemm_h2n.hip
int lvae_gemm_launch_patch2(const lvae_gemm_desc* d, hipStream_t st);                        // gemm_f32_patch2.hip
int lvae_gemm_launch_conv3(const lvae_gemm_desc* d, hipStream_t st);                         // gemm_f32_conv3.hip
int lvae_gemm_lp_dispatch(const lvae_gemm_desc* d, hipStream_t st);                           // gemm_lp.hip
int lvae_gemm_q8_dispatch(const lvae_gemm_desc* d, hipStream_t st);                           // gemm_q8.hip
static int gemm_dispatch_impl(const lvae_gemm_desc* d, hipStream_t st, int x3v2, int x3v2_tn);
static int gemm_dispatch(const lvae_gemm_desc* d, hipStream_t st, int x3v2, int x3v2_tn) { return gemm_dispatch_impl(d, st, x3v2, x3v2_tn); }

extern "C" int lvae_gemm_f32(const lvae_gemm_desc* d, void* stream) {
    static int x3v2 = 1, x3v2_tn = 0;
#ifdef LVAE_EXPERIMENTAL_BUILD             // tuning hooks of tools/build_exp.sh copies only (every choice gives the same bits; the product
    static bool env_read = false;          // library's launch paths read no environment): LVAE_X3V2=0 keeps prec 2 on gemm_x3_kernel,
    if (!env_read) {                       // LVAE_X3V2_TN forces its tile, LVAE_GEMM_CFG the legacy kernels' configuration
        const char* e = getenv("LVAE_GEMM_CFG"); if (e) g_force_cfg = atoi(e);
        e = getenv("LVAE_X3V2"); if (e) x3v2 = atoi(e);
        e = getenv("LVAE_X3V2_TN"); if (e) x3v2_tn = atoi(e);
        env_read = true;
    }
#endif
    if (!d || !d->A0 || (!d->Wt && !(d->prec != 0 && d->Wt16)) || !d->out || d->M <= 0 || d->N <= 0 || d->K <= 0) return -22;
    if ((d->K & 3) || (d->ldw & 3)) return -22;                       // 16-B operand loads
    if (d->prec < 0 || d->prec > 4) return -22;
    if (d->prec != 3 && (d->a_bf16 || d->out_bf16)) return -22;      // bf16 storage exists in the reduced-precision mode only
    if (d->prec != 4 && d->prec != 3 && (d->a_h2 || d->out_h2)) return -22;   // pre-converted operands: f16x2 planes (prec 4) / MX-fp8 (prec 3)
    if (d->prec == 3) {
        if ((d->epi == LVAE_EPI_GAMMA_RES || d->epi == LVAE_EPI_RES) && !d->res) return -22;
        if (d->epi == LVAE_EPI_GAMMA_RES && !d->gamma) return -22;
        if (d->store != LVAE_ST_ROWMAJOR && (d->r <= 0 || d->N % (d->r * d->r) || d->H <= 0 || d->W <= 0)) return -22;
        if (d->a_h2) return lvae_gemm_q8_dispatch(d, (hipStream_t)stream);          // both operands already MX-fp8 in memory
        if (d->out_h2) return -22;
        return lvae_gemm_lp_dispatch(d, (hipStream_t)stream);
    }
    if (d->prec != 0 && (!d->Wt16 || (d->K & 7) || (d->ldw & 7))) return -22;
    if ((d->epi == LVAE_EPI_GAMMA_RES || d->epi == LVAE_EPI_RES) && !d->res) return -22;
    if (d->epi == LVAE_EPI_GAMMA_RES && !d->gamma) return -22;
    if (d->store != LVAE_ST_ROWMAJOR && (d->r <= 0 || d->N % (d->r * d->r) || d->H <= 0 || d->W <= 0)) return -22;
    hipStream_t st = (hipStream_t)stream;
    const int S = ksp(d);
    if (S > 1 && d->prec == 4 && d->a_h2) {
        // pre-split operands: SERIAL split-K inside gemm_h2p_kernel (one workgroup adds the S slice sums in slice order: the parallel
        // form's bits without workspace or reduce launch)
        if (d->store != LVAE_ST_ROWMAJOR || (d->N & 3) || (d->ldo & 3) || (d->ldres & 3) || d->K % (32 * S)) return -22;
        return gemm_dispatch(d, st, x3v2, x3v2_tn);
    }
    if (S > 1 && d->prec == 4 && (d->cfg == 0 || d->cfg == 3)) {
        // narrow outputs over large maps: gemm_h2n_kernel walks the slices serially too (same bits, no workspace traffic, no reduction)
        int rc = 0;
        if (lvae_gemm_h2n_try(d, st, d->cfg == 3, &rc)) return rc;
    }
    if (S > 1) {
        if (!d->ws || d->store != LVAE_ST_ROWMAJOR || (d->N & 3) || (d->ldo & 3) || (d->ldres & 3) || d->K % (32 * S) ||
            (d->prec == 1 && d->K % (64 * S)))
            return -22;
        // In-kernel reduction only when no 128-B line of the workspace can hold columns of two different tiles (N % 32 == 0, or one
        // n-tile): a tile's reducer acquires at agent scope, which drops its CU's L1 but not lines its XCD's L2 fetched earlier in
        // this launch for a NEIGHBOURING tile's reduction.
        if (d->cnt && !((d->N & 31) == 0 || d->N <= 32)) {
            lvae_gemm_desc d2 = *d;
            d2.cnt = nullptr;
            return lvae_gemm_f32(&d2, stream);
        }
        const int rc = gemm_dispatch(d, st, x3v2, x3v2_tn);
        if (rc || d->cnt || d->defer_reduce) return rc;               // cnt: reduced in place by each tile's last-arriving slice; defer_reduce: by the consumer
        const long n = (long)d->M * (d->N >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *d, S);
        return (int)hipGetLastError();
    }
    return gemm_dispatch(d, st, x3v2, x3v2_tn);
}

static int gemm_dispatch_impl(const lvae_gemm_desc* d, hipStream_t st, int x3v2, int x3v2_tn) {
    if (d->prec == 4) {                  // f16x2: one kernel family; the host asks for it only where it applies (engine.h2_eligible)
        static int h2_tn = 0, h2p_tile = 0;
#ifdef LVAE_EXPERIMENTAL_BUILD             // tile sweeps (tools/h2p_sweep.sh) on experimental builds only
        static bool h2_env = false;
        if (!h2_env) {
            const char* e = getenv("LVAE_H2_TN"); h2_tn = e ? atoi(e) : 0;
            e = getenv("LVAE_H2P_TILE"); h2p_tile = e ? atoi(e) : 0;
            h2_env = true;
        }
#endif
        int rc = 0;
        if (d->out_h2 && (d->store != LVAE_ST_ROWMAJOR || (d->epi != LVAE_EPI_BIAS && d->epi != LVAE_EPI_BIAS_GELU) || (d->N & 31) ||
                          d->ldo != d->N || (d->ksplit > 1 && !d->a_h2)))
            return -22;
        if (d->a_h2) return lvae_gemm_h2p_try(d, st, d->cfg > 0 ? d->cfg : h2p_tile, &rc) ? rc : -22;      // cfg = 10 WM + TN: force a tile
        return lvae_gemm_h2_try(d, st, d->cfg > 0 ? d->cfg : h2_tn, &rc) ? rc : -22;                          // cfg: gemm_h2.hip's force codes
    }
    if (d->prec == 2 && x3v2 && d->cfg == 0 &&
        ((d->a_mode == LVAE_A_PLAIN && d->K0 + d->K1 == d->K) || d->a_mode == LVAE_A_CONV3)) {   // cfg -1: legacy kernel
        int rc = 0;
        if (lvae_gemm_x3v2_try(d, st, x3v2_tn, &rc)) return rc;
    }
    switch (d->a_mode) {
        case LVAE_A_PLAIN:
            if ((d->K0 & 3) || (d->lda0 & 3) || d->K0 + d->K1 != d->K || (d->K1 && (!d->A1 || (d->lda1 & 3) || (d->K1 & 3))))
                return -22;
            return launch_mode<LVAE_A_PLAIN>(d, st);
        case LVAE_A_PATCH2:
            if ((d->K0 & 3) || d->K != 4 * d->K0 || d->H <= 0 || d->W <= 0) return -22;
            return lvae_gemm_launch_patch2(d, st);
        case LVAE_A_CONV3:
            if ((d->K0 & 3) || d->K != 9 * d->K0 || d->H <= 0 || d->W <= 0) return -22;
            return lvae_gemm_launch_conv3(d, st);
    }
    return -22;
}
#endif  // LVAE_GEMM_TU_AMODE
