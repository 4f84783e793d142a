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

#include "../../include/lvae_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;        // k-tile
constexpr int LDT = BK + 4;   // padded LDS row (floats)

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int WGM_, int WGN_, int TM_, int TN_>
struct Cfg {
    static constexpr int WGM = WGM_, WGN = WGN_, TM = TM_, TN = TN_;
    static constexpr int BM = WGM * TM * 32;
    static constexpr int BN = WGN * TN * 32;
    static constexpr int NA = BM * 8 / 256;   // float4 loads per thread per k-tile (A)
    static constexpr int NB = BN * 8 / 256;   // (W)
    static constexpr int LDS_BYTES = 2 * (BM + BN) * LDT * 4;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(NA >= 1 && NB >= 1, "tile too small");
};

struct RowInfo {      // per staged A row of this thread
    long off;         // PLAIN: row index m; PATCH2/CONV3: element offset of the row's first tap; -1 = beyond M
    int h, w;         // CONV3: pixel coordinates
};

template <int AMODE>
__device__ __forceinline__ RowInfo row_info(const lvae_gemm_desc& d, int m) {
    RowInfo ri;
    ri.h = ri.w = 0;
    if (m >= d.M) { ri.off = -1; return ri; }
    if (AMODE == LVAE_A_PLAIN) {
        ri.off = m;
    } else if (AMODE == LVAE_A_PATCH2) {
        const int w = m % d.W, bh = m / d.W;          // bh = b*H + h on the OUTPUT grid; input rows 2*bh, 2*bh+1
        ri.off = ((long)bh * 2 * (2L * d.W) + 2L * w) * d.K0;
    } else {
        const int w = m % d.W, bh = m / d.W;
        ri.w = w;
        ri.h = bh % d.H;
        ri.off = (long)m * d.K0;
    }
    return ri;
}

template <int AMODE>
__device__ __forceinline__ f32x4 load_a(const lvae_gemm_desc& d, const RowInfo& ri, int k) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (ri.off < 0 || k >= d.K) return z;
    if (AMODE == LVAE_A_PLAIN) {
        if (k < d.K0) return *(const f32x4*)(d.A0 + ri.off * d.lda0 + k);
        return *(const f32x4*)(d.A1 + ri.off * d.lda1 + (k - d.K0));
    } else if (AMODE == LVAE_A_PATCH2) {
        const int seg = 2 * d.K0;                 // one input row of the 2x2 patch: 2 pixels x Cin
        const int s = k / seg, kk = k - s * seg;
        return *(const f32x4*)(d.A0 + ri.off + (long)s * (2L * d.W) * d.K0 + kk);
    } else {
        const int s = k / d.K0, kk = k - s * d.K0;
        const int di = s / 3 - 1, dj = s - (s / 3) * 3 - 1;
        const int hh = ri.h + di, ww = ri.w + dj;
        if (hh < 0 || hh >= d.H || ww < 0 || ww >= d.W) return z;
        return *(const f32x4*)(d.A0 + ri.off + ((long)di * d.W + dj) * d.K0 + kk);
    }
}

template <class C, int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDT]
    float* Ws = smem + 2 * C::BM * LDT;        // [2][BN][LDT]

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
    const int srow = tid >> 3, sk4 = tid & 7;
    RowInfo ri[C::NA];
#pragma unroll
    for (int i = 0; i < C::NA; ++i) ri[i] = row_info<AMODE>(d, m0 + srow + 32 * i);
    long woff[C::NB];
#pragma unroll
    for (int i = 0; i < C::NB; ++i) {
        const int n = n0 + srow + 32 * i;
        woff[i] = n < d.N ? (long)n * d.ldw : -1;
    }

    f32x16 acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    f32x4 ra[C::NA], rb[C::NB];
    const int nk = (d.K + BK - 1) / BK;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto gload = [&](int kt) {
        const int k = kt * BK + sk4 * 4;
#pragma unroll
        for (int i = 0; i < C::NA; ++i) ra[i] = load_a<AMODE>(d, ri[i], k);
#pragma unroll
        for (int i = 0; i < C::NB; ++i) rb[i] = (woff[i] >= 0 && k < d.K) ? *(const f32x4*)(d.Wt + woff[i] + k) : zero4;
    };
    auto lstore = [&](int buf) {
        float* a = As + buf * C::BM * LDT + srow * LDT + sk4 * 4;
        float* w = Ws + buf * C::BN * LDT + srow * LDT + sk4 * 4;
#pragma unroll
        for (int i = 0; i < C::NA; ++i) *(f32x4*)(a + 32 * i * LDT) = ra[i];
#pragma unroll
        for (int i = 0; i < C::NB; ++i) *(f32x4*)(w + 32 * i * LDT) = rb[i];
    };

    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
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
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int rr = d.r, r2 = rr * rr;
#pragma unroll
    for (int b = 0; b < C::TN; ++b) {
        const int col = n0 + (wave_n * C::TN + b) * 32 + li;
        if (col >= d.N) continue;
        const float bias = d.bias ? d.bias[col] : 0.f;
        const float gam = (d.epi == LVAE_EPI_GAMMA_RES) ? d.gamma[col] : 1.f;
        int sc = 0, si = 0, sj = 0, cp = 1;          // shuffle decomposition of the column
        if (d.store == LVAE_ST_SHUFFLE) { cp = d.N / r2; const int q = col / cp; sc = col - q * cp; si = q / rr; sj = q - si * rr; }
        if (d.store == LVAE_ST_IMAGE)   { cp = d.N / r2; sc = col / r2; const int q = col - sc * r2; si = q / rr; sj = q - si * rr; }
#pragma unroll
        for (int a = 0; a < C::TM; ++a) {
            const int rbase = m0 + (wave_m * C::TM + a) * 32 + 4 * lh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row >= d.M) continue;
                float v = acc[a][b][r] + bias;
                if (d.epi == LVAE_EPI_BIAS_GELU) v = gelu_erf(v);
                else if (d.epi == LVAE_EPI_GAMMA_RES) v = d.res[(long)row * d.ldres + col] + gam * v;
                else if (d.epi == LVAE_EPI_RES) v = d.res[(long)row * d.ldres + col] + v;
                if (d.store == LVAE_ST_ROWMAJOR) {
                    d.out[(long)row * d.ldo + col] = v;
                } else {
                    const int w = row % d.W, bh = row / d.W, h = bh % d.H, bb = bh / d.H;
                    if (d.store == LVAE_ST_SHUFFLE) {
                        const long pix = ((long)(bb * d.H + h) * rr + si) * (d.W * rr) + (long)w * rr + sj;
                        d.out[pix * cp + sc] = v;
                    } else {
                        v = fminf(fmaxf(v, -1.0f), 1.0f) * 0.5f + 0.5f;
                        const long o = (((long)bb * cp + sc) * (d.H * rr) + (long)h * rr + si) * (d.W * rr) + (long)w * rr + sj;
                        d.out[o] = v;
                    }
                }
            }
        }
    }
}

template <class C, int AMODE>
int launch_cfg(const lvae_gemm_desc* d, hipStream_t st) {
    const int tiles_m = (d->M + C::BM - 1) / C::BM, tiles_n = (d->N + C::BN - 1) / C::BN;
    const int n_tiles = tiles_m * tiles_n;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<C, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<C, AMODE>), dim3(n_tiles), dim3(256), C::LDS_BYTES, st, *d, tiles_n, n_tiles);
    return (int)hipGetLastError();
}

typedef Cfg<2, 2, 2, 2> CfgA;   // 128 x 128, wave 64x64
typedef Cfg<2, 2, 2, 1> CfgB;   // 128 x 64,  wave 64x32
typedef Cfg<4, 1, 1, 1> CfgC;   // 128 x 32,  wave 32x32
typedef Cfg<2, 2, 1, 1> CfgS;   //  64 x 64,  wave 32x32

template <int AMODE>
int launch_mode(const lvae_gemm_desc* d, hipStream_t st) {
    const int N = d->N, M = d->M;
    if (N <= 32 || N == 96) return launch_cfg<CfgC, AMODE>(d, st);
    if (N % 128 == 0) {
        const long tilesA = (long)((M + 127) / 128) * (N / 128);
        if (tilesA >= 192) return launch_cfg<CfgA, AMODE>(d, st);
        return launch_cfg<CfgS, AMODE>(d, st);
    }
    const long tilesB = (long)((M + 127) / 128) * ((N + 63) / 64);
    if (tilesB >= 192 || N < 64) return launch_cfg<CfgB, AMODE>(d, st);
    return launch_cfg<CfgS, AMODE>(d, st);
}

}  // namespace

extern "C" int lvae_gemm_f32(const lvae_gemm_desc* d, void* stream) {
    if (!d || !d->A0 || !d->Wt || !d->out || d->M <= 0 || d->N <= 0 || d->K <= 0) return -22;
    if ((d->K & 3) || (d->ldw & 3)) return -22;                       // 16-B operand loads
    if ((d->epi == LVAE_EPI_GAMMA_RES || d->epi == LVAE_EPI_RES) && !d->res) return -22;
    if (d->epi == LVAE_EPI_GAMMA_RES && !d->gamma) return -22;
    if (d->store != LVAE_ST_ROWMAJOR && (d->r <= 0 || d->N % (d->r * d->r) || d->H <= 0 || d->W <= 0)) return -22;
    hipStream_t st = (hipStream_t)stream;
    switch (d->a_mode) {
        case LVAE_A_PLAIN:
            if ((d->K0 & 3) || (d->lda0 & 3) || d->K0 + d->K1 != d->K || (d->K1 && (!d->A1 || (d->lda1 & 3) || (d->K1 & 3))))
                return -22;
            return launch_mode<LVAE_A_PLAIN>(d, st);
        case LVAE_A_PATCH2:
            if ((d->K0 & 3) || d->K != 4 * d->K0 || d->H <= 0 || d->W <= 0) return -22;
            return launch_mode<LVAE_A_PATCH2>(d, st);
        case LVAE_A_CONV3:
            if ((d->K0 & 3) || d->K != 9 * d->K0 || d->H <= 0 || d->W <= 0) return -22;
            return launch_mode<LVAE_A_CONV3>(d, st);
    }
    return -22;
}
