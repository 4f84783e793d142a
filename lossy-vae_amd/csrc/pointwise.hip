// pointwise.hip -- the HBM-bound kernels of the QARV / QRes-VAE hot path for gfx950 (MI355X).
//
//  * lvae_dwconv_ln_f32 : depthwise kxk conv + bias -> LayerNorm(C) -> (affine) -> AdaLN, one pass over an NHWC map
//                         (lvae/models/common.py:145-152; qresvae/model.py:168-176).  Replaces cuDNN depthwise conv,
//                         2 permute().contiguous() copies, ATen layer_norm and ~6 elementwise launches per block.
//  * lvae_stem_f32      : preprocess_input (qarv/model.py:221) fused into the 4x4/s4 stem conv (zoo.py:37).
//  * lvae_gemv_f32      : lambda-embedding MLP + all AdaLN embedding layers (common.py:123-127) as one GEMV.
//  * lvae_prior_index_f32 / lvae_quantize_f32 / lvae_dequantize_f32 : softplus/exp/LowerBound/build_indexes/quantize
//                         (qarv/model.py:51-53,106-108,112-113): one launch instead of >= 126 compare/sub launches.
//
// All are bandwidth-bound integer/float streaming kernels: the design rules are 16-B per-lane coalesced NHWC
// accesses, wave64 shuffles for the per-pixel reductions, and >> 256 workgroups per launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/lvae_hip.h"
#include "device_math.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Activation storage of the reduced-precision mode (BASELINE config 5): bf16 bit patterns, fp32 arithmetic in registers.
// BF = false: plain fp32 maps (the parity path).  Offsets are in ELEMENTS.
template <bool BF>
__device__ __forceinline__ f32x4 ld4(const void* base, long off) {
    if (BF) {
        const u32x2 q = *(const u32x2*)((const unsigned short*)base + off);
        return (f32x4){__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16),
                       __uint_as_float(q[1] & 0xffff0000u)};
    }
    return *(const f32x4*)((const float*)base + off);
}
__device__ __forceinline__ unsigned f2bf(float x) {            // round to nearest even
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <bool BF>
__device__ __forceinline__ void st4(void* base, long off, f32x4 v) {
    if (BF) {
        const u32x2 q = {f2bf(v[0]) | (f2bf(v[1]) << 16), f2bf(v[2]) | (f2bf(v[3]) << 16)};
        *(u32x2*)((unsigned short*)base + off) = q;
    } else {
        *(f32x4*)((float*)base + off) = v;
    }
}


// ------------------------------------------------------------------------------------------------ dwconv + LN + AdaLN
// A wave64 is split into 64/LPP pixel groups; each group of LPP lanes owns TW consecutive output pixels of one image
// row and all C = 4*VPL*LPP channels (lane cl holds float4 channel chunks cl, cl+LPP, ...).  The kxk window slides
// along W in registers: (TW+k-1) input float4 per kernel row feed TW outputs, i.e. ~(TW+k-1)/TW loads per output tap
// row instead of k.  LayerNorm statistics are reduced across the LPP lanes with xor-shuffles (two-pass variance on the
// register-resident conv outputs).
template <int KS, int VPL, int LPP, int TH, bool BF = false>
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const void* __restrict__ x, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                        const float* __restrict__ ln_b, const float* __restrict__ shift,
                                                        const float* __restrict__ scale1p, void* __restrict__ y,
                                                        int B, int H, int W, int gpr, int hgr, long total_groups) {
    constexpr int TW = 4;
    constexpr int C = 4 * VPL * LPP;
    constexpr int GPW = 64 / LPP;
    constexpr int P = (KS - 1) / 2;
    const int lane = threadIdx.x & 63;
    // XCD-aware bijective remap (workgroup b runs on XCD b%8, each XCD has its own L2): give every XCD a CONTIGUOUS band
    // of the (image, row, column-group) space so that the k-1 halo rows a workgroup shares with its neighbours are served
    // by the same L2.  With the round-robin default, PMC showed 3.4x the algorithmic bytes fetched from HBM.
    long blk;
    {
        const long nb = gridDim.x, b = blockIdx.x, q = nb / 8, r = nb % 8, xcd = b % 8, loc = b / 8;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const long wave_g = blk * 4 + (threadIdx.x >> 6);
    const long g = wave_g * GPW + lane / LPP;
    const int cl = lane % LPP;
    const bool active = g < total_groups;
    const long gg = active ? g : total_groups - 1;
    const int w0 = (int)(gg % gpr) * TW;
    const long bhg = gg / gpr;                 // b*hgr + row group
    const int h0 = (int)(bhg % hgr) * TH;      // first of the TH output rows of this group
    const long brow = (bhg / hgr) * H;         // b*H

    // Each group produces TH x TW outputs: an input row (TW+k-1 float4 per channel chunk) is loaded ONCE and feeds the
    // up-to-TH output rows it belongs to.  The 7x7 layers are bound by L2 bandwidth (the k-fold re-reads of the map go to L2,
    // not HBM: 27 TB/s of 34.5 TB/s measured with TH = 1): TH = 2 cuts the input loads per output from 17.5 to 10.
    f32x4 acc[VPL][TH][TW];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int c = 4 * (cl + v * LPP);
        const f32x4 bv = *(const f32x4*)(bias + c);
#pragma unroll
        for (int th = 0; th < TH; ++th)
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[v][th][t] = bv;
        // input rows are a RUNTIME loop on purpose: fully unrolled, hipcc hoists all (TW+k-1)*k loads of a channel
        // chunk to the top and spills kilobytes per lane to scratch (measured: 3.7 KB/lane, 16x slower).
#pragma unroll 1
        for (int r = 0; r < KS + TH - 1; ++r) {
            const int hh = h0 + r - P;
            const bool rv = (hh >= 0) && (hh < H);
            const long xrow = ((brow + hh) * W) * (long)C + c;
            f32x4 xr[TW + KS - 1];
#pragma unroll
            for (int q = 0; q < TW + KS - 1; ++q) {
                const int ww = w0 + q - P;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                xr[q] = (rv && ww >= 0 && ww < W) ? ld4<BF>(x, xrow + (long)ww * C) : z;
            }
#pragma unroll
            for (int th = 0; th < TH; ++th) {
                const int i = r - th;                        // kernel row this input row is for output row h0 + th
                if (i >= 0 && i < KS) {
                    const float* wrow = wt + (long)(i * KS) * C + c;
#pragma unroll
                    for (int j = 0; j < KS; ++j) {
                        const f32x4 wv = *(const f32x4*)(wrow + (long)j * C);
#pragma unroll
                        for (int t = 0; t < TW; ++t) {
                            acc[v][th][t][0] = fmaf(xr[t + j][0], wv[0], acc[v][th][t][0]);
                            acc[v][th][t][1] = fmaf(xr[t + j][1], wv[1], acc[v][th][t][1]);
                            acc[v][th][t][2] = fmaf(xr[t + j][2], wv[2], acc[v][th][t][2]);
                            acc[v][th][t][3] = fmaf(xr[t + j][3], wv[3], acc[v][th][t][3]);
                        }
                    }
                }
            }
        }
    }

    const float inv_c = 1.0f / (float)C;
#pragma unroll
    for (int th = 0; th < TH; ++th) {
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) s += (acc[v][th][t][0] + acc[v][th][t][1]) + (acc[v][th][t][2] + acc[v][th][t][3]);
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * inv_c;
            float sq = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = acc[v][th][t][e] - mean;
                    acc[v][th][t][e] = dlt;
                    sq = fmaf(dlt, dlt, sq);
                }
            }
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            const float rstd = 1.0f / sqrtf(sq * inv_c + 1e-6f);
            const int ww = w0 + t, hh = h0 + th;
            if (active && ww < W && hh < H) {
                const long yp = (((brow + hh) * W) + ww) * (long)C;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int c = 4 * (cl + v * LPP);
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = acc[v][th][t][e] * rstd;
                    if (ln_w) {
                        const f32x4 lw = *(const f32x4*)(ln_w + c), lb = *(const f32x4*)(ln_b + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = o4[e] * lw[e] + lb[e];
                    }
                    if (shift) {
                        const f32x4 sc = *(const f32x4*)(scale1p + c), sh = *(const f32x4*)(shift + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = o4[e] * sc[e] + sh[e];
                    }
                    st4<BF>(y, yp + c, o4);
                }
            }
        }
    }
}

// Tiled, software-pipelined form of the same operator for the large C = 128 / 192 maps (stride 4).  dwconv_ln_kernel re-reads every
// input pixel k times per channel chunk from L2 (nothing survives in the 32 KB L1): ~27 of the 34.5 TB/s of L2 bandwidth at k = 7.
// Here 512 threads (32 groups of 16 lanes x 4 pixels) own an 8 x 16 pixel tile; per channel chunk the tile and its (k-1) halo are
// staged ONCE in LDS and the k x k window slides over LDS; all k*k*C weights sit in LDS too, so the compute phase issues no
// vector-memory instruction and the global loads of the NEXT chunk (or of the next tile's first chunk: workgroups are persistent)
// stay in flight in registers behind it (an in-order vmcnt wait on a weight load would otherwise drain them).
// Channel -> lane ownership, tap order and the LayerNorm reduction order are dwconv_ln_kernel's: same bits, free choice per launch.
template <int KS, int VPL>
__global__ __launch_bounds__(512, 1) void dwconv_ln_tile_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                                const float* __restrict__ ln_b, const float* __restrict__ shift,
                                                                const float* __restrict__ scale1p, float* __restrict__ y,
                                                                int B, int H, int W, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int LPP = 16, TW = 4, GX = 4, GY = 8, TBW = GX * TW, TBH = GY, PGB = 32;
    constexpr int C = 4 * VPL * LPP;
    constexpr int P = (KS - 1) / 2, LW = TBW + KS - 1, LH = TBH + KS - 1, NPIX = LW * LH;
    constexpr int NF = (NPIX + PGB - 1) / PGB;                       // staged float4 per thread per chunk
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    f32x4* tile = (f32x4*)dw_lds;                                   // [NPIX][16]
    f32x4* wl = tile + NPIX * LPP;                                  // [VPL][KS*KS][16]
    const int tid = threadIdx.x, cl = tid % LPP, pg = tid / LPP;
    const int gy = pg / GX, gx = pg % GX;
    // contiguous range of tiles per workgroup (neighbouring tiles share halos in one XCD's L2: blockIdx % 8 is the XCD)
    const int nb = gridDim.x, bq = nb / 8, br = nb % 8, xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    const int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + loc;
    const int per = n_tiles / nb, rem = n_tiles % nb;
    const int t_begin = bid * per + (bid < rem ? bid : rem), t_end = t_begin + per + (bid < rem ? 1 : 0);

    for (int e = tid; e < VPL * KS * KS * LPP; e += 512) {           // weights: wl[(v*KK + tap)*16 + lane] = wt[tap*C + 4*(lane + 16 v)]
        const int lane = e % LPP, tap = (e / LPP) % (KS * KS), v = e / (LPP * KS * KS);
        wl[e] = *(const f32x4*)(wt + (long)tap * C + 4 * (lane + v * LPP));
    }
    int lyx[NF];                                                     // halo-tile coordinates of this thread's staged pixels
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        const int pix = pg + PGB * n, ly = pix / LW;
        lyx[n] = (pix < NPIX) ? ((ly << 8) | (pix - ly * LW)) : -1;
    }
    f32x4 st[NF];
    unsigned inmask = 0;
    auto prefetch = [&](int t, int v) {                              // global -> registers, branch-free (clamped address + mask)
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        const long b = t / (tiles_x * tiles_y);
        const float* xb = x + b * (long)H * W * C + 4 * (cl + v * LPP);
        const int h0 = ty * TBH - P, w0 = tx * TBW - P;
        inmask = 0;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int hh = h0 + (lyx[n] >> 8), ww = w0 + (lyx[n] & 255);
            const bool in = lyx[n] >= 0 && hh >= 0 && hh < H && ww >= 0 && ww < W;
            inmask |= (in ? 1u : 0u) << n;
            st[n] = *(const f32x4*)(xb + (in ? ((long)hh * W + ww) * C : 0));
        }
    };
    auto commit = [&]() {                                            // registers -> LDS tile
#pragma unroll
        for (int n = 0; n < NF; ++n)
            if (lyx[n] >= 0) tile[(pg + PGB * n) * LPP + cl] = ((inmask >> n) & 1u) ? st[n] : (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    if (t_begin < t_end) prefetch(t_begin, 0);
    for (int t = t_begin; t < t_end; ++t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        const long b = t / (tiles_x * tiles_y);
        const int h0 = ty * TBH, w0 = tx * TBW;
        f32x4 acc[VPL][TW];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            __syncthreads();                                         // previous chunk's readers are done (first pass: weights are in)
            commit();
            __syncthreads();
            if (v + 1 < VPL) prefetch(t, v + 1);
            else if (t + 1 < t_end) prefetch(t + 1, 0);
            const f32x4 bv = *(const f32x4*)(bias + 4 * (cl + v * LPP));
#pragma unroll
            for (int q = 0; q < TW; ++q) acc[v][q] = bv;
#pragma unroll 1
            for (int i = 0; i < KS; ++i) {
                const f32x4* row = tile + ((gy + i) * LW + gx * TW) * LPP + cl;
                const f32x4* wrow = wl + (v * KS * KS + i * KS) * LPP + cl;
                f32x4 xr[TW + KS - 1];
#pragma unroll
                for (int q = 0; q < TW + KS - 1; ++q) xr[q] = row[q * LPP];
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    const f32x4 wv = wrow[j * LPP];
#pragma unroll
                    for (int q = 0; q < TW; ++q) {
                        acc[v][q][0] = fmaf(xr[q + j][0], wv[0], acc[v][q][0]);
                        acc[v][q][1] = fmaf(xr[q + j][1], wv[1], acc[v][q][1]);
                        acc[v][q][2] = fmaf(xr[q + j][2], wv[2], acc[v][q][2]);
                        acc[v][q][3] = fmaf(xr[q + j][3], wv[3], acc[v][q][3]);
                    }
                }
            }
        }
        const float inv_c = 1.0f / (float)C;
        const int hh = h0 + gy;
#pragma unroll
        for (int q = 0; q < TW; ++q) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) s += (acc[v][q][0] + acc[v][q][1]) + (acc[v][q][2] + acc[v][q][3]);
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * inv_c;
            float sq = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dlt = acc[v][q][e] - mean;
                    acc[v][q][e] = dlt;
                    sq = fmaf(dlt, dlt, sq);
                }
            }
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            const float rstd = 1.0f / sqrtf(sq * inv_c + 1e-6f);
            const int ww = w0 + gx * TW + q;
            if (ww < W && hh < H) {
                float* yp = y + ((b * H + hh) * (long)W + ww) * C;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int c = 4 * (cl + v * LPP);
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = acc[v][q][e] * rstd;
                    if (ln_w) {
                        const f32x4 lw = *(const f32x4*)(ln_w + c), lb = *(const f32x4*)(ln_b + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = o4[e] * lw[e] + lb[e];
                    }
                    if (shift) {
                        const f32x4 sc = *(const f32x4*)(scale1p + c), sh = *(const f32x4*)(shift + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = o4[e] * sc[e] + sh[e];
                    }
                    *(f32x4*)(yp + c) = o4;
                }
            }
        }
    }
}

// Second-generation tiled form ("t2"): the same operator for every C = 64 NV layer on maps with enough tiles to fill the chip, fp32
// or bf16 storage.  What the PMC counters said about dwconv_ln_tile_kernel (B = 8, 128x192, C = 192: 127 us): its 2048 waves are
// parked 40 % of their cycles (s_waitcnt / barrier) and issue-stalled another 23 %, VALU-active 25 %, LDS array 32 % busy -- one
// 512-thread workgroup per CU (116 KB of LDS) runs its phases [barrier, registers -> LDS, barrier, prefetch, taps] one after the
// other and nothing else is resident to fill the gaps.  Here a workgroup is 256 threads on a 4 x 16 pixel tile and keeps only ONE
// channel chunk's halo tile + weights in LDS (69 KB at k = 7, 47 KB at k = 5): two to three workgroups share a CU and overlap each
// other's staging, barriers and global-load latency; the taps are explicit packed FMAs (v_pk_fma_f32, half the VALU issue slots;
// each component is an fmaf, so the bits do not change).  16 lanes own a pixel group (4 pixels x all channels, chunk c of lane cl =
// channels 4 (cl + 16 c) ..+3); the LayerNorm sums are formed in exactly the association of dwconv_ln_kernel's <.., LPP = RL, ..>
// instance for this C (RL = 32: the first butterfly step, lane ^ 16, happens inside the lane), so all three kernels give the same
// bits and the launcher may choose by map size.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KS, int NV, int RL, bool BF>
__global__ __launch_bounds__(256, 2) void dwconv_ln_t2_kernel(const void* __restrict__ x, const float* __restrict__ wt,
                                                              const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, const float* __restrict__ shift,
                                                              const float* __restrict__ scale1p, void* __restrict__ y,
                                                              int B, int H, int W, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int LPP = 16, TW = 4, GX = 4, TBW = 16, TBH = 4, PGB = 16;
    constexpr int C = 64 * NV;
    constexpr int P = (KS - 1) / 2, LW = TBW + KS - 1, LH = TBH + KS - 1, NPIX = LW * LH, KK = KS * KS;
    constexpr int NF = (NPIX + PGB - 1) / PGB;                       // staged pixels per thread per chunk
    constexpr int NWF = (KK + PGB - 1) / PGB;                        // staged weight taps per thread per chunk
    static_assert(RL == 16 || (RL == 32 && NV % 2 == 0), "reference lane count");
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    f32x4* tile = (f32x4*)dw_lds;                                   // [NPIX][16]
    f32x4* wl = tile + NPIX * LPP;                                  // [KK][16]
    const int tid = threadIdx.x, cl = tid % LPP, pg = tid / LPP;
    const int gy = pg / GX, gx = pg % GX;
    const int nb = gridDim.x, bq = nb / 8, br = nb % 8, xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    const int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + loc;
    const int per = n_tiles / nb, rem = n_tiles % nb;
    const int t_begin = bid * per + (bid < rem ? bid : rem), t_end = t_begin + per + (bid < rem ? 1 : 0);

    f32x4 st[NF], sw[NWF];
    unsigned inmask = 0;
    auto prefetch = [&](int t, int v) {                              // global -> registers, branch-free (clamped address + mask)
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        const long b = t / (tiles_x * tiles_y);
        const long xb = b * (long)H * W * C + 4 * (cl + v * LPP);
        const int h0 = ty * TBH - P, w0 = tx * TBW - P;
        inmask = 0;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int pix = pg + PGB * n, ly = pix / LW, lx = pix - ly * LW;     // halo-tile coordinates (recomputed: registers are scarce)
            const int hh = h0 + ly, ww = w0 + lx;
            const bool in = pix < NPIX && hh >= 0 && hh < H && ww >= 0 && ww < W;
            inmask |= (in ? 1u : 0u) << n;
            st[n] = ld4<BF>(x, xb + (in ? ((long)hh * W + ww) * C : 0));
        }
#pragma unroll
        for (int n = 0; n < NWF; ++n) {
            const int tap = pg + PGB * n;
            sw[n] = *(const f32x4*)(wt + (long)(tap < KK ? tap : 0) * C + 4 * (cl + v * LPP));
        }
    };
    auto commit = [&]() {                                            // registers -> LDS
#pragma unroll
        for (int n = 0; n < NF; ++n)
            if (pg + PGB * n < NPIX) tile[(pg + PGB * n) * LPP + cl] = ((inmask >> n) & 1u) ? st[n] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NWF; ++n)
            if (pg + PGB * n < KK) wl[(pg + PGB * n) * LPP + cl] = sw[n];
    };

    if (t_begin < t_end) prefetch(t_begin, 0);
    for (int t = t_begin; t < t_end; ++t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        const long b = t / (tiles_x * tiles_y);
        const int h0 = ty * TBH, w0 = tx * TBW;
        f32x4 acc[NV][TW];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            __syncthreads();                                         // previous chunk's readers are done
            commit();
            __syncthreads();
            if (v + 1 < NV) prefetch(t, v + 1);
            else if (t + 1 < t_end) prefetch(t + 1, 0);
            const f32x4 bv = *(const f32x4*)(bias + 4 * (cl + v * LPP));
#pragma unroll
            for (int q = 0; q < TW; ++q) acc[v][q] = bv;
#pragma unroll 1
            for (int i = 0; i < KS; ++i) {
                const f32x4* row = tile + ((gy + i) * LW + gx * TW) * LPP + cl;
                const f32x4* wrow = wl + (i * KS) * LPP + cl;
                f32x4 xr[TW + KS - 1];
#pragma unroll
                for (int q = 0; q < TW + KS - 1; ++q) xr[q] = row[q * LPP];
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    const f32x4 wv = wrow[j * LPP];
#pragma unroll
                    for (int q = 0; q < TW; ++q) acc[v][q] = __builtin_elementwise_fma(xr[q + j], wv, acc[v][q]);
                }
            }
        }
        const float inv_c = 1.0f / (float)C;
        const int hh = h0 + gy;
#pragma unroll
        for (int q = 0; q < TW; ++q) {
            // per-lane partial sums in the reference kernel's order: lane c of its RL lanes adds its chunks c, c + RL, c + 2 RL, ...
            float s;
            if (RL == 16) {
                s = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) s += (acc[v][q][0] + acc[v][q][1]) + (acc[v][q][2] + acc[v][q][3]);
            } else {
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int v = 0; v < NV; v += 2) {
                    sa += (acc[v][q][0] + acc[v][q][1]) + (acc[v][q][2] + acc[v][q][3]);
                    sb += (acc[v + 1][q][0] + acc[v + 1][q][1]) + (acc[v + 1][q][2] + acc[v + 1][q][3]);
                }
                s = sa + sb;                                         // the reference's lane ^ 16 step
            }
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * inv_c;
            float sq;
            if (RL == 16) {
                sq = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dlt = acc[v][q][e] - mean;
                        acc[v][q][e] = dlt;
                        sq = fmaf(dlt, dlt, sq);
                    }
                }
            } else {
                float qa = 0.f, qb = 0.f;
#pragma unroll
                for (int v = 0; v < NV; v += 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float da = acc[v][q][e] - mean, db = acc[v + 1][q][e] - mean;
                        acc[v][q][e] = da; acc[v + 1][q][e] = db;
                        qa = fmaf(da, da, qa);
                        qb = fmaf(db, db, qb);
                    }
                }
                sq = qa + qb;
            }
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            const float rstd = 1.0f / sqrtf(sq * inv_c + 1e-6f);
            const int ww = w0 + gx * TW + q;
            if (ww < W && hh < H) {
                const long yp = ((b * H + hh) * (long)W + ww) * C;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int c = 4 * (cl + v * LPP);
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = acc[v][q][e] * rstd;
                    if (ln_w) {
                        const f32x4 lw = *(const f32x4*)(ln_w + c), lb = *(const f32x4*)(ln_b + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = o4[e] * lw[e] + lb[e];
                    }
                    if (shift) {
                        const f32x4 sc = *(const f32x4*)(scale1p + c), sh = *(const f32x4*)(shift + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = o4[e] * sc[e] + sh[e];
                    }
                    st4<BF>(y, yp + c, o4);
                }
            }
        }
    }
}

// Third-generation tiled form ("t3"), built from what the counters said about the two before it: per CU the operator needs ~28 us of
// packed FMAs, ~33 us of LDS fragment reads, ~20-36 us of halo staging and 30-60 us of HBM time on the stride-4 maps -- the earlier
// kernels run these one after the other (dwconv_ln_tile_kernel: waves parked 40 % + stalled 23 % of their cycles), t2 overlapped
// them across workgroups but staged 3.4 halo pixels per output pixel.  Here ONE 512-thread workgroup per CU overlaps them itself:
//   * a pixel group is 8 lanes (chunk = 32 channels = 128 B per pixel), so a halo tile of a chunk is small enough to DOUBLE-BUFFER
//     in LDS next to ALL the weights (k = 7, C = 192: 2 x 61 KB + 37 KB): chunk q+1 is written into the other buffer in the middle of
//     chunk q's taps and its global loads were issued a whole chunk earlier -- one barrier per chunk, no exposed staging phase;
//   * 16 x 16 pixel tiles for C <= 192 (4 pixels per lane group: 1.9 halo pixels staged per output pixel), 8 x 16 tiles with 2 pixels
//     per group for C = 256..512 (the accumulators of all C channels of a pixel must stay in registers for the LayerNorm);
//   * packed FMAs (v_pk_fma_f32): each component is an fmaf, same bits, half the VALU issue slots.
// The LayerNorm sums reproduce the association of dwconv_ln_kernel's <.., LPP = RL, ..> instance for this C: lane l of the 8 holds
// the 4-channel chunks l + 8 j; reference lane c = (l + 8 j) mod RL adds its chunks in ascending order, then the reference's xor
// butterfly: its steps >= 8 combine partial sums that live in ONE lane here, the steps 4, 2, 1 cross the 8 lanes.  Same bits as every
// other form of this operator (tests/test_gpu_kernels.py), so the launcher may choose by map size.
template <bool BF> struct DwStage { typedef f32x4 type; };
template <> struct DwStage<true> { typedef u32x2 type; };

template <int KS, int NV8, int RL, int TW, bool BF>
__global__ __launch_bounds__(512, 1) void dwconv_ln_t3_kernel(const void* __restrict__ x, const float* __restrict__ wt,
                                                              const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, const float* __restrict__ shift,
                                                              const float* __restrict__ scale1p, void* __restrict__ y,
                                                              int B, int H, int W, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int LPP = 8, TBW = 16, GX = TBW / TW, GY = 64 / GX, TBH = GY, PGB = 64;
    constexpr int C = 32 * NV8, NG = RL / 8;                         // NG reference lanes (c = l + 8 g) share one lane here
    constexpr int P = (KS - 1) / 2, LW = TBW + KS - 1, LH = TBH + KS - 1, NPIX = LW * LH, KK = KS * KS;
    constexpr int NF = (NPIX + PGB - 1) / PGB;                       // staged pixels per thread per chunk
    static_assert(NV8 % NG == 0 && NV8 % 2 == 0, "chunks per reference lane / buffer parity");
    typedef typename DwStage<BF>::type stage_t;
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    f32x4* tile0 = (f32x4*)dw_lds;                                  // [2][NPIX][8]
    f32x4* wl = tile0 + 2 * NPIX * LPP;                             // [NV8][KK][8]
    const int tid = threadIdx.x, cl = tid % LPP, pg = tid / LPP;
    const int gy = pg / GX, gx = pg % GX;
    const int nb = gridDim.x, bq = nb / 8, br = nb % 8, xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    const int bid = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + loc;
    const int per = n_tiles / nb, rem = n_tiles % nb;
    const int t_begin = bid * per + (bid < rem ? bid : rem), t_end = t_begin + per + (bid < rem ? 1 : 0);
    if (t_begin >= t_end) return;

    for (int e = tid; e < NV8 * KK * LPP; e += 512) {                // weights: wl[(v*KK + tap)*8 + lane] = wt[tap*C + 4*(lane + 8 v)]
        const int lane = e % LPP, tap = (e / LPP) % KK, v = e / (LPP * KK);
        wl[e] = *(const f32x4*)(wt + (long)tap * C + 4 * (lane + v * LPP));
    }
    // LDS pixel slot: p ^ ((p >> 2) & 1).  The four pixel groups a 16-lane ds_read_b128 service group spans sit 4 pixels apart
    // (same parity => same 32-bank half with 128-B pixels: 2-way conflicts); swapping the pixels of every other aligned quad in pairs
    // alternates the half from group to group.
    auto slot = [](int p) { return p ^ ((p >> 2) & 1); };
    stage_t st[NF];
    unsigned inmask = 0;
    auto prefetch = [&](int t, int v) {                              // global -> registers, branch-free (clamped address + mask)
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        const long b = t / (tiles_x * tiles_y);
        const long xb = b * (long)H * W * C + 4 * (cl + v * LPP);
        const int h0 = ty * TBH - P, w0 = tx * TBW - P;
        inmask = 0;
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int pix = pg + PGB * n, ly = pix / LW, lx = pix - ly * LW;
            const int hh = h0 + ly, ww = w0 + lx;
            const bool in = pix < NPIX && hh >= 0 && hh < H && ww >= 0 && ww < W;
            inmask |= (in ? 1u : 0u) << n;
            const long off = xb + (in ? ((long)hh * W + ww) * C : 0);
            if (BF) st[n] = *(const stage_t*)((const unsigned short*)x + off);
            else st[n] = *(const stage_t*)((const float*)x + off);
        }
    };
    auto commit = [&](f32x4* tile) {                                 // registers -> LDS (other buffer)
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            if (pg + PGB * n < NPIX) {
                f32x4 v4;
                if constexpr (BF) {
                    v4 = (f32x4){__uint_as_float(st[n][0] << 16), __uint_as_float(st[n][0] & 0xffff0000u), __uint_as_float(st[n][1] << 16),
                                 __uint_as_float(st[n][1] & 0xffff0000u)};
                } else {
                    v4 = st[n];
                }
                tile[slot(pg + PGB * n) * LPP + cl] = ((inmask >> n) & 1u) ? v4 : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    prefetch(t_begin, 0);
    commit(tile0);
    prefetch(t_begin, 1);                                            // NV8 >= 2
    __syncthreads();                                                 // weights + first chunk visible
    for (int t = t_begin; t < t_end; ++t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        const long b = t / (tiles_x * tiles_y);
        const int h0 = ty * TBH, w0 = tx * TBW;
        f32x4 acc[NV8][TW];
#pragma unroll
        for (int v = 0; v < NV8; ++v) {
            const f32x4* tile = tile0 + (v & 1) * NPIX * LPP;        // NV8 is even: chunk parity == buffer parity for every tile
            f32x4* other = tile0 + ((v & 1) ^ 1) * NPIX * LPP;
            const bool has_next = (v + 1 < NV8) || (t + 1 < t_end);
            const f32x4 bv = *(const f32x4*)(bias + 4 * (cl + v * LPP));
#pragma unroll
            for (int q = 0; q < TW; ++q) acc[v][q] = bv;
            // (a two-row software pipeline of this loop -- row i + 1's 17 fragment reads in flight behind row i's FMAs -- needs ~270
            //  registers with the staging set and spilled: 141 -> 207 us; measured and dropped)
#pragma unroll 1
            for (int i = 0; i < KS; ++i) {
                const int p0 = (gy + i) * LW + gx * TW;
                const f32x4* wrow = wl + (v * KK + i * KS) * LPP + cl;
                f32x4 xr[TW + KS - 1];
#pragma unroll
                for (int q = 0; q < TW + KS - 1; ++q) xr[q] = tile[slot(p0 + q) * LPP + cl];
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    const f32x4 wv = wrow[j * LPP];
#pragma unroll
                    for (int q = 0; q < TW; ++q) acc[v][q] = __builtin_elementwise_fma(xr[q + j], wv, acc[v][q]);
                }
                if (i == KS / 2 && has_next) {
                    // the next chunk (in flight since the previous step) goes into the other buffer, whose last readers passed the
                    // barrier that ended the previous step; then the chunk after it is requested
                    commit(other);
                    int t2 = t, v2 = v + 2;
                    if (v2 >= NV8) { t2 = t + 1; v2 -= NV8; }
                    if (t2 < t_end) prefetch(t2, v2);
                }
            }
            if (v + 1 < NV8) __syncthreads();
        }
        const float inv_c = 1.0f / (float)C;
        const int hh = h0 + gy;
#pragma unroll
        for (int q = 0; q < TW; ++q) {
            float pgs[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float s = 0.f;
#pragma unroll
                for (int v = g; v < NV8; v += NG) s += (acc[v][q][0] + acc[v][q][1]) + (acc[v][q][2] + acc[v][q][3]);
                pgs[g] = s;
            }
            float s;
            if (NG == 4) s = (pgs[0] + pgs[2]) + (pgs[1] + pgs[3]);   // the reference's lane ^ 16 step, then lane ^ 8
            else s = pgs[0] + pgs[1];                                 // lane ^ 8
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * inv_c;
            float qgs[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float sq = 0.f;
#pragma unroll
                for (int v = g; v < NV8; v += NG) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float dlt = acc[v][q][e] - mean;
                        acc[v][q][e] = dlt;
                        sq = fmaf(dlt, dlt, sq);
                    }
                }
                qgs[g] = sq;
            }
            float sq;
            if (NG == 4) sq = (qgs[0] + qgs[2]) + (qgs[1] + qgs[3]);
            else sq = qgs[0] + qgs[1];
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            const float rstd = 1.0f / sqrtf(sq * inv_c + 1e-6f);
            const int ww = w0 + gx * TW + q;
            if (ww < W && hh < H) {
                const long yp = ((b * H + hh) * (long)W + ww) * C;
#pragma unroll
                for (int v = 0; v < NV8; ++v) {
                    const int c = 4 * (cl + v * LPP);
                    f32x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = acc[v][q][e] * rstd;
                    if (ln_w) {
                        const f32x4 lw = *(const f32x4*)(ln_w + c), lb = *(const f32x4*)(ln_b + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = o4[e] * lw[e] + lb[e];
                    }
                    if (shift) {
                        const f32x4 sc = *(const f32x4*)(scale1p + c), sh = *(const f32x4*)(shift + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = o4[e] * sc[e] + sh[e];
                    }
                    st4<BF>(y, yp + c, o4);
                }
            }
        }
        __syncthreads();                                             // the last chunk's readers are done before its buffer is rewritten
    }
}

int g_dw_t3 = -1;      // tuning hook (LVAE_DW_T3): 0 = never, 1 = whenever an instance exists, -1 = heuristic

template <int KS, int NV8, int RL, int TW, bool BF>
int launch_dwln_t3(const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                   const float* shift, const float* scale1p, void* y, int B, int H, int W, hipStream_t st) {
    constexpr int TBH = 64 / (16 / TW), NPIX = (16 + KS - 1) * (TBH + KS - 1), LDS = (2 * NPIX * 8 + NV8 * KS * KS * 8) * 16;
    static_assert(LDS <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)dwconv_ln_t3_kernel<KS, NV8, RL, TW, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_x = (W + 15) / 16, tiles_y = (H + TBH - 1) / TBH;
    const long n_tiles = (long)B * tiles_x * tiles_y;
    const int grid = (int)(n_tiles < 256 ? n_tiles : 256);
    hipLaunchKernelGGL((dwconv_ln_t3_kernel<KS, NV8, RL, TW, BF>), dim3(grid), dim3(512), LDS, st, x, wt, bias, ln_w, ln_b, shift, scale1p, y,
                       B, H, W, tiles_x, tiles_y, (int)n_tiles);
    return (int)hipGetLastError();
}

// t3 is taken when the map has at least one tile per CU (below that the register sliding-window kernel's finer granularity wins)
template <int KS, bool BF>
int try_dwln_t3(int C, const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b, const float* shift,
                const float* scale1p, void* y, int B, int H, int W, hipStream_t st, int* rc) {
    // bf16 maps only: on fp32 maps it ties dwconv_ln_tile_kernel (B = 8, 128x192, C = 192: 137 vs 134 us) and loses to the sliding-window
    // kernel at C >= 256 (106 vs 82 us).  What all forms share is the tap loop: per tap ~0.55 us of FMA issue + ~0.67 us of LDS
    // fragment reads that do not overlap at two waves per SIMD (measured: t = 57 + 1.45 k^2 us for k = 1, 3, 5, 7), and the registers
    // that a two-row software pipeline would need are taken by the accumulators of all C channels (LayerNorm needs them together).
    if constexpr (KS >= 3 && BF) {
        if (g_dw_t3 == 0) return 0;
        const int tbh = C <= 192 ? 16 : 8;
        const long n_tiles = (long)B * ((W + 15) / 16) * ((H + tbh - 1) / tbh);
        if (g_dw_t3 < 0 && n_tiles < 256) return 0;
        switch (C) {
            case 128: *rc = launch_dwln_t3<KS, 4, 16, 4, BF>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st); return 1;
            case 192: *rc = launch_dwln_t3<KS, 6, 16, 4, BF>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st); return 1;
            case 256: *rc = launch_dwln_t3<KS, 8, 32, 2, BF>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st); return 1;
            case 384: *rc = launch_dwln_t3<KS, 12, 32, 2, BF>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st); return 1;
            case 512:
                if constexpr (KS <= 5) { *rc = launch_dwln_t3<KS, 16, 32, 2, BF>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st); return 1; }
                return 0;
        }
    }
    return 0;
}

int g_dw_t2 = -1;      // tuning hook (LVAE_DW_T2): 0 = never, 1 = whenever an instance exists, -1 = heuristic

template <int KS, int NV, int RL, bool BF>
int launch_dwln_t2(const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                   const float* shift, const float* scale1p, void* y, int B, int H, int W, hipStream_t st) {
    constexpr int NPIX = (16 + KS - 1) * (4 + KS - 1), LDS = (NPIX + KS * KS) * 256;
    static_assert(LDS <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)dwconv_ln_t2_kernel<KS, NV, RL, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 3) / 4;
    const long n_tiles = (long)B * tiles_x * tiles_y;
    const int per_cu = (160 * 1024) / LDS < 4 ? (160 * 1024) / LDS : 4;
    const long cap = 256L * per_cu;
    const int grid = (int)(n_tiles < cap ? n_tiles : cap);
    hipLaunchKernelGGL((dwconv_ln_t2_kernel<KS, NV, RL, BF>), dim3(grid), dim3(256), LDS, st, x, wt, bias, ln_w, ln_b, shift, scale1p, y,
                       B, H, W, tiles_x, tiles_y, (int)n_tiles);
    return (int)hipGetLastError();
}

// t2 is taken when the map has at least ~1.5 tiles per CU (below that the register sliding-window kernel's finer granularity wins)
template <int KS, bool BF>
int try_dwln_t2(int C, const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b, const float* shift,
                const float* scale1p, void* y, int B, int H, int W, hipStream_t st, int* rc) {
    if constexpr (KS >= 3) {
        const long n_tiles = (long)B * ((W + 15) / 16) * ((H + 3) / 4);
        // fp32 maps: measured no faster than dwconv_ln_tile_kernel / the sliding-window kernel (B = 8, 128x192, C = 192: 172 vs 134 us --
        // its 4 x 16 tiles stage 3.4 halo pixels per output pixel against 2.4 for the 8 x 16 tiles), so it is taken for bf16 maps only,
        // where it replaces the 8-byte-per-lane sliding-window loads (257 -> 140 us); LVAE_DW_T2=1 forces it for experiments
        if (g_dw_t2 == 0 || (g_dw_t2 < 0 && (!BF || n_tiles < 384))) return 0;
        switch (C) {
            case 128: *rc = launch_dwln_t2<KS, 2, 16, BF>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st); return 1;
            case 192: *rc = launch_dwln_t2<KS, 3, 16, BF>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st); return 1;
            case 256: *rc = launch_dwln_t2<KS, 4, 32, BF>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st); return 1;
            case 384: *rc = launch_dwln_t2<KS, 6, 32, BF>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st); return 1;
        }
    }
    return 0;
}

int g_dw_tile = -1;    // tuning hook (LVAE_DW_TILE): 0 = never, 1 = whenever an instance exists, -1 = heuristic

template <int KS, int VPL>
int launch_dwln_tile(const float* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                     const float* shift, const float* scale1p, float* y, int B, int H, int W, hipStream_t st) {
    constexpr int NPIX = (16 + KS - 1) * (8 + KS - 1), LDS = (NPIX + VPL * KS * KS) * 256;
    static_assert(LDS <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)dwconv_ln_tile_kernel<KS, VPL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 7) / 8;
    const long n_tiles = (long)B * tiles_x * tiles_y;
    const int grid = n_tiles < 256 ? (int)n_tiles : 256;
    hipLaunchKernelGGL((dwconv_ln_tile_kernel<KS, VPL>), dim3(grid), dim3(512), LDS, st, x, wt, bias, ln_w, ln_b, shift, scale1p, y,
                       B, H, W, tiles_x, tiles_y, (int)n_tiles);
    return (int)hipGetLastError();
}

int g_dw_th = 0;       // tuning hook (LVAE_DW_TH): 1 or 2 output rows per group; 0 = heuristic

template <int KS, int VPL, int LPP, int TH, bool BF = false>
int launch_dwln_th(const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                   const float* shift, const float* scale1p, void* y, int B, int H, int W, hipStream_t st) {
    const int gpr = (W + 3) / 4, hgr = (H + TH - 1) / TH;
    const long total = (long)B * hgr * gpr;
    const int gpw = 64 / LPP;
    const long waves = (total + gpw - 1) / gpw;
    const long blocks = (waves + 3) / 4;
    hipLaunchKernelGGL((dwconv_ln_kernel<KS, VPL, LPP, TH, BF>), dim3((unsigned)blocks), dim3(256), 0, st, x, wt, bias, ln_w, ln_b,
                       shift, scale1p, y, B, H, W, gpr, hgr, total);
    return (int)hipGetLastError();
}

template <int KS, int VPL, int LPP>
int launch_dwln(const float* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                const float* shift, const float* scale1p, float* y, int B, int H, int W, hipStream_t st) {
    // two output rows per group (measured, B = 8): +12..17 % on the stride-4 maps (C <= 192, ~200k pixels, L2-bandwidth-bound);
    // slower on the C >= 256 layers, where 200+ VGPRs halve the occupancy.  Same accumulation order => same bits either way.
    const long px = (long)B * H * W;
    constexpr int C = 4 * VPL * LPP;
    if constexpr (KS >= 5 && LPP == 16 && VPL <= 3) {
        if (g_dw_tile == 1 || (g_dw_tile < 0 && px >= 90000 && H >= 16 && W >= 32))      // >= ~3 tiles per CU; measured equal at 1.5
            return launch_dwln_tile<KS, VPL>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
    }
    int th = g_dw_th ? g_dw_th : ((KS == 7 && C <= 192 && VPL <= 3 && px >= 100000) ? 2 : 1);
    if (KS == 1 || VPL > 4) th = 1;          // VPL = 9 (C = 144, 288) has no registers for a second row
    if (th == 2) return launch_dwln_th<KS, VPL, LPP, (KS == 1 ? 1 : 2)>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
    return launch_dwln_th<KS, VPL, LPP, 1>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
}

template <int KS>
int dispatch_dwln_c(int C, const float* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                    const float* shift, const float* scale1p, float* y, int B, int H, int W, hipStream_t st) {
    switch (C) {
        case 128: return launch_dwln<KS, 2, 16>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 144: return launch_dwln<KS, 9, 4>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);      // qres17m
        case 288: return launch_dwln<KS, 9, 8>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 192: return launch_dwln<KS, 3, 16>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 256: return launch_dwln<KS, 2, 32>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 384: return launch_dwln<KS, 3, 32>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 512: return launch_dwln<KS, 4, 32>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
    }
    return -22;
}

// bf16-storage form (reduced-precision mode): the register sliding-window kernel, one output row per group
template <int KS>
int dispatch_dwln_bf16(int C, const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                       const float* shift, const float* scale1p, void* y, int B, int H, int W, hipStream_t st) {
    switch (C) {
        case 128: return launch_dwln_th<KS, 2, 16, 1, true>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 192: return launch_dwln_th<KS, 3, 16, 1, true>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 256: return launch_dwln_th<KS, 2, 32, 1, true>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 384: return launch_dwln_th<KS, 3, 32, 1, true>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 512: return launch_dwln_th<KS, 4, 32, 1, true>(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
    }
    return -22;
}

// ------------------------------------------------------------------------------------------------ stem
// One block = 64 output pixels x Cout channels (thread n = output channel, its 48 weights live in registers; the
// 64x48 preprocessed patch matrix is staged in LDS and read as wave-uniform broadcasts).
// range_flag (optional): the reference's input contract `0 <= im.min() <= im.max() <= 1` (qarv/model.py:219-220, qresvae/model.py:492)
// checked on the values this kernel loads anyway: bit 0 of *range_flag is set when any pixel is outside [0, 1] or NaN; the host
// reads the flag at a synchronisation point it already has (no extra device sync, unlike the reference's .min()/.max()).
template <bool BF>
__global__ void stem_kernel(const float* __restrict__ im, const float* __restrict__ wt, const float* __restrict__ bias,
                            void* __restrict__ out, int B, int H, int W, int Cout, float im_shift, float im_scale, long M,
                            int* __restrict__ range_flag) {
    __shared__ __attribute__((aligned(16))) float patch[64][48];
    const int n = threadIdx.x;
    const int Ho = H / 4, Wo = W / 4;
    const long p0 = (long)blockIdx.x * 64;
    bool bad = false;
    for (int e = n; e < 64 * 48; e += blockDim.x) {
        const int j = e & 3, p = (e >> 2) & 63, ci_i = e >> 8;       // ci_i = ci*4 + i
        const long pg = p0 + p;
        float v = 0.f;
        if (pg < M) {
            const int wo = (int)(pg % Wo);
            const long bho = pg / Wo;
            const int ho = (int)(bho % Ho);
            const long b = bho / Ho;
            const int ci = ci_i >> 2, i = ci_i & 3;
            v = im[((b * 3 + ci) * H + (4 * ho + i)) * (long)W + 4 * wo + j];
            bad |= !(v >= 0.0f && v <= 1.0f);
            v = (v + im_shift) * im_scale;
        }
        patch[p][ci_i * 4 + j] = v;
    }
    if (range_flag && bad) atomicOr(range_flag, 1);
    float wr[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) wr[k] = wt[k * Cout + n];
    const float bv = bias[n];
    __syncthreads();
    for (int p = 0; p < 64; ++p) {
        const long pg = p0 + p;
        if (pg >= M) break;
        float a = bv;
#pragma unroll
        for (int k4 = 0; k4 < 12; ++k4) {
            const f32x4 pv = *(const f32x4*)&patch[p][k4 * 4];
            a = fmaf(pv[0], wr[k4 * 4 + 0], a);
            a = fmaf(pv[1], wr[k4 * 4 + 1], a);
            a = fmaf(pv[2], wr[k4 * 4 + 2], a);
            a = fmaf(pv[3], wr[k4 * 4 + 3], a);
        }
        if (BF) ((unsigned short*)out)[pg * Cout + n] = (unsigned short)f2bf(a);
        else ((float*)out)[pg * Cout + n] = a;
    }
}

// ------------------------------------------------------------------------------------------------ gemv
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ Wt, const float* __restrict__ b,
                                                   const float* __restrict__ x, float* __restrict__ y, int N, int K,
                                                   int gelu_in, int gelu_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 wv = *(const f32x4*)(Wt + (long)n * K + k);
        f32x4 xv = *(const f32x4*)(x + k);
        if (gelu_in) { xv[0] = gelu_erf(xv[0]); xv[1] = gelu_erf(xv[1]); xv[2] = gelu_erf(xv[2]); xv[3] = gelu_erf(xv[3]); }
        s = fmaf(wv[0], xv[0], s); s = fmaf(wv[1], xv[1], s); s = fmaf(wv[2], xv[2], s); s = fmaf(wv[3], xv[3], s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        s += b[n];
        y[n] = gelu_out ? gelu_erf(s) : s;
    }
}

// ------------------------------------------------------------------------------------------------ lossless output net
__global__ void lossless_params_kernel(const float* __restrict__ raw, const float* __restrict__ im, float* __restrict__ pm,
                                       uint8_t* __restrict__ idx, int32_t* __restrict__ sym, const float* __restrict__ table,
                                       int n_scales, float bound, long total, int HW) {
#pragma clang fp contract(off)
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;     // e = (b*3 + c)*HW + p   (NCHW raster)
    if (e >= total) return;
    const long bc = e / HW;
    const int p = (int)(e - bc * HW);
    const long b = bc / 3;
    const int c = (int)(bc - b * 3);
    const float* r = raw + (b * HW + p) * 6;
    const float bin = (float)(1.0 / 127.5);
    float m = r[c] * 127.5f;
    m = m + 127.5f;
    m = rintf(m) / 127.5f;
    m = m - 1.0f;
    m = m / bin;
    const float ls = r[3 + c] - (float)(-4.848116360536466);        // - math.log(1/127.5)
    const float s = fmaxf(expf(ls), bound);
    int lo = 0, hi = n_scales - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (table[mid] < s) lo = mid + 1; else hi = mid;
    }
    pm[e] = m;
    idx[e] = (uint8_t)lo;
    if (im) {
        float x = im[e] - 0.5f;
        x = x * 2.0f;
        x = x / bin;
        sym[e] = (int32_t)rintf(x - m);
    }
}

__global__ void lossless_output_kernel(const int32_t* __restrict__ sym, const float* __restrict__ pm, float* __restrict__ out, long n) {
#pragma clang fp contract(off)
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float x = (float)sym[e] + pm[e];
    x = x * (float)(1.0 / 127.5);
    x = fminf(fmaxf(x, -1.0f), 1.0f);
    x = x * 0.5f;
    out[e] = x + 0.5f;
}

// ------------------------------------------------------------------------------------------------ prior sampling
// Philox4x32-10 (Salmon et al., SC'11): counter-based, so element e of a launch gets the same variates whatever the grid.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void prior_sample_kernel(const float* __restrict__ prm, float* __restrict__ z, long total, int zdim, int ldz, float t,
                                    uint64_t seed, uint64_t offset) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;     // e = m*ldz + c
    if (e >= total) return;
    const long m = e / ldz;
    const int c = (int)(e - m * ldz);
    if (c >= zdim) { z[e] = 0.f; return; }
    const float mean = prm[m * 2 * zdim + c];
    const float lv = prm[m * 2 * zdim + zdim + c];
    const float xs = lv + 2.3f;
    const float sp = xs > 20.0f ? xs : log1pf(expf(xs));
    const float pv = expf(sp - 2.3f);
    const uint64_t ctr = offset + (uint64_t)(m * zdim + c);
    uint32_t r[4];
    philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
    const float u2 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float nrm = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
    const float uni = ((float)(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f) - 0.5f;
    z[e] = mean + pv * nrm * t + uni * t;
}

// ------------------------------------------------------------------------------------------------ entropy parameters
__global__ void prior_index_kernel(const float* __restrict__ prm, float* __restrict__ pm, uint8_t* __restrict__ idx,
                                   const float* __restrict__ table, int n_scales, float bound, long total, int HW, int z) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;     // e = m*z + c
    if (e >= total) return;
    const long m = e / z;
    const int c = (int)(e - m * z);
    const float mean = prm[m * 2 * z + c];
    const float lv = prm[m * 2 * z + z + c];
    // softplus(x + 2.3) - 2.3  (torch: beta=1, threshold=20)
    const float xs = lv + 2.3f;
    const float sp = xs > 20.0f ? xs : log1pf(expf(xs));
    const float pv = expf(sp - 2.3f);
    const float s = fmaxf(pv, bound);
    // idx = #{i < n_scales-1 : table[i] < s}   (binary search, table ascending)
    int lo = 0, hi = n_scales - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (table[mid] < s) lo = mid + 1; else hi = mid;
    }
    pm[e] = mean;
    const long b = m / HW;
    const int p = (int)(m - b * HW);
    idx[(b * z + c) * HW + p] = (uint8_t)lo;
}

__global__ void quantize_kernel(const float* __restrict__ qm, const float* __restrict__ pm, int32_t* __restrict__ sym,
                                float* __restrict__ zhat, long total, int HW, int z, int ldz) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;     // e = m*ldz + c over the PADDED rows
    if (e >= total) return;
    const long m = e / ldz;
    const int c = (int)(e - m * ldz);
    if (c >= z) { zhat[e] = 0.f; return; }
    const float mu = pm[m * z + c];
    const float r = rintf(qm[m * z + c] - mu);  // v_rndne_f32: round-half-to-even == torch.round
    zhat[e] = r + mu;
    const long b = m / HW;
    const int p = (int)(m - b * HW);
    sym[(b * z + c) * HW + p] = (int32_t)r;
}

__global__ void dequantize_kernel(const int32_t* __restrict__ sym, const float* __restrict__ pm, float* __restrict__ zhat,
                                  long total, int HW, int z, int ldz) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const long m = e / ldz;
    const int c = (int)(e - m * ldz);
    if (c >= z) { zhat[e] = 0.f; return; }
    const long b = m / HW;
    const int p = (int)(m - b * HW);
    zhat[e] = (float)sym[(b * z + c) * HW + p] + pm[m * z + c];
}

// Eval-mode rate estimate (qarv/model.py:95-96; CompressAI GaussianConditional._likelihood): per latent element
// P = Phi((.5-|v|)/s) - Phi((-.5-|v|)/s), v = zhat - mean = the integer symbol, s = max(exp(softplus(x+2.3)-2.3), bound),
// P = max(P, 1e-9); accumulates sum(-ln P) per image (nats) in fp64.  Phi in fp32 as the reference: erf form for
// DiscretizedGaussian (underflows to exactly 0 in the tails), erfc form for stock GaussianConditional.
__global__ __launch_bounds__(256) void gaussian_nll_kernel(const float* __restrict__ prm, const int32_t* __restrict__ sym,
                                                           double* __restrict__ out, float bound, long per_image, int HW,
                                                           int z, int cdf_form) {
    const int b = blockIdx.y;
    double acc = 0.0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < per_image; e += (long)gridDim.x * 256) {
        const long p = e / z;                    // pixel within the image
        const int c = (int)(e - p * z);
        const long m = (long)b * HW + p;
        const float lv = prm[m * 2 * z + z + c];
        const float xs = lv + 2.3f;
        const float sp = xs > 20.0f ? xs : log1pf(expf(xs));
        const float s = fmaxf(expf(sp - 2.3f), bound);
        const float v = fabsf((float)sym[((long)b * z + c) * HW + p]);
        const float a = (0.5f - v) / s, d = (-0.5f - v) / s;
        float up, lo;
        if (cdf_form == 0) {
            up = 0.5f * (1.0f + lvae_erff(a * 0.70710678118654752440f));
            lo = 0.5f * (1.0f + lvae_erff(d * 0.70710678118654752440f));
        } else {
            up = 0.5f * erfcf(-0.70710678118654752440f * a);
            lo = 0.5f * erfcf(-0.70710678118654752440f * d);
        }
        const float P = fmaxf(up - lo, 1e-9f);
        acc -= (double)logf(P);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + b, ws[0] + ws[1] + ws[2] + ws[3]);
}

__global__ void bias_expand_kernel(const float* __restrict__ bias, float* __restrict__ out, long total4, int C4) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    ((f32x4*)out)[e] = ((const f32x4*)bias)[e % C4];
}

__global__ __launch_bounds__(256) void sqerr_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    double* __restrict__ out, long n) {
    const int img = blockIdx.y;
    const float* pa = a + (long)img * n;
    const float* pb = b + (long)img * n;
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float dlt = pa[i] - pb[i];
        s += (double)dlt * (double)dlt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + img, ws[0] + ws[1] + ws[2] + ws[3]);
}

}  // namespace

// dwconv_cl.hip: the channel-per-lane form (rolling register window, weights in registers); takes the problem by (C, k) alone
int lvae_dwln_cl_try(const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b, const float* shift,
                     const float* scale1p, void* y, int B, int H, int W, int C, int k, int bf16, hipStream_t st, int* rc);

extern "C" int lvae_dwconv_ln_f32(const float* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                                  const float* shift, const float* scale1p, float* y, int B, int H, int W, int C, int k,
                                  void* stream) {
    if (!x || !wt || !bias || !y || B <= 0 || H <= 0 || W <= 0) return -22;
    if ((ln_w == nullptr) != (ln_b == nullptr) || (shift == nullptr) != (scale1p == nullptr)) return -22;
    {
        int rc = 0;
        if (lvae_dwln_cl_try(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, C, k, 0, (hipStream_t)stream, &rc)) return rc;
    }
    static bool env_read = false;
    if (!env_read) {
        const char* e = getenv("LVAE_DW_TH"); if (e) g_dw_th = atoi(e);
        e = getenv("LVAE_DW_TILE"); if (e) g_dw_tile = atoi(e);
        env_read = true;
    }
    hipStream_t st = (hipStream_t)stream;
    {
        const char* e = nullptr;
        static bool t2_read = false;
        if (!t2_read) { e = getenv("LVAE_DW_T2"); if (e) g_dw_t2 = atoi(e); t2_read = true; }
        int rc = 0;
        static bool t3_read = false;
        if (!t3_read) { const char* e3 = getenv("LVAE_DW_T3"); if (e3) g_dw_t3 = atoi(e3); t3_read = true; }
        if (g_dw_tile != 1 && g_dw_th == 0 && g_dw_t2 != 1) {
            if (k == 3 && try_dwln_t3<3, false>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
            if (k == 5 && try_dwln_t3<5, false>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
            if (k == 7 && try_dwln_t3<7, false>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
        }
        if (g_dw_tile != 1 && g_dw_th == 0) {
            if (k == 3 && try_dwln_t2<3, false>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
            if (k == 5 && try_dwln_t2<5, false>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
            if (k == 7 && try_dwln_t2<7, false>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
        }
    }
    switch (k) {
        case 1: return dispatch_dwln_c<1>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 3: return dispatch_dwln_c<3>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 5: return dispatch_dwln_c<5>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 7: return dispatch_dwln_c<7>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
    }
    return -22;
}

extern "C" int lvae_stem_f32(const float* im, const float* wt, const float* bias, float* out, int B, int H, int W,
                             int Cout, float im_shift, float im_scale, int* range_flag, void* stream) {
    if (!im || !wt || !bias || !out || B <= 0 || (H & 3) || (W & 3) || Cout <= 0 || Cout > 256) return -22;
    const long M = (long)B * (H / 4) * (W / 4);
    hipLaunchKernelGGL(stem_kernel<false>, dim3((unsigned)((M + 63) / 64)), dim3(Cout), 0, (hipStream_t)stream, im, wt, bias, out,
                       B, H, W, Cout, im_shift, im_scale, M, range_flag);
    return (int)hipGetLastError();
}

extern "C" int lvae_stem_bf16(const float* im, const float* wt, const float* bias, void* out, int B, int H, int W,
                              int Cout, float im_shift, float im_scale, int* range_flag, void* stream) {
    if (!im || !wt || !bias || !out || B <= 0 || (H & 3) || (W & 3) || Cout <= 0 || Cout > 256) return -22;
    const long M = (long)B * (H / 4) * (W / 4);
    hipLaunchKernelGGL(stem_kernel<true>, dim3((unsigned)((M + 63) / 64)), dim3(Cout), 0, (hipStream_t)stream, im, wt, bias, out,
                       B, H, W, Cout, im_shift, im_scale, M, range_flag);
    return (int)hipGetLastError();
}

extern "C" int lvae_dwconv_ln_bf16(const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                                   const float* shift, const float* scale1p, void* y, int B, int H, int W, int C, int k,
                                   void* stream) {
    if (!x || !wt || !bias || !y || B <= 0 || H <= 0 || W <= 0) return -22;
    if ((ln_w == nullptr) != (ln_b == nullptr) || (shift == nullptr) != (scale1p == nullptr)) return -22;
    hipStream_t st = (hipStream_t)stream;
    {
        int rc = 0;
        if (lvae_dwln_cl_try(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, C, k, 1, st, &rc)) return rc;
    }
    {
        static bool t2_read = false;
        if (!t2_read) { const char* e = getenv("LVAE_DW_T2"); if (e) g_dw_t2 = atoi(e); t2_read = true; }
        int rc = 0;
        static bool t3_read = false;
        if (!t3_read) { const char* e3 = getenv("LVAE_DW_T3"); if (e3) g_dw_t3 = atoi(e3); t3_read = true; }
        if (g_dw_t2 != 1) {
            if (k == 3 && try_dwln_t3<3, true>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
            if (k == 5 && try_dwln_t3<5, true>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
            if (k == 7 && try_dwln_t3<7, true>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
        }
        if (k == 3 && try_dwln_t2<3, true>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
        if (k == 5 && try_dwln_t2<5, true>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
        if (k == 7 && try_dwln_t2<7, true>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st, &rc)) return rc;
    }
    switch (k) {
        case 1: return dispatch_dwln_bf16<1>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 3: return dispatch_dwln_bf16<3>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 5: return dispatch_dwln_bf16<5>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 7: return dispatch_dwln_bf16<7>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
    }
    return -22;
}

namespace {
__global__ void bias_expand_bf16_kernel(const float* __restrict__ bias, unsigned short* __restrict__ out, long total4, int C4) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total4) return;
    st4<true>(out, e * 4, ((const f32x4*)bias)[e % C4]);
}
}  // namespace

extern "C" int lvae_bias_expand_bf16(const float* bias, void* out, long M, int C, void* stream) {
    if (!bias || !out || M <= 0 || C <= 0 || (C & 3)) return -22;
    const long total4 = M * (C / 4);
    hipLaunchKernelGGL(bias_expand_bf16_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bias,
                       (unsigned short*)out, total4, C / 4);
    return (int)hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256) void range_flag_kernel(const float* __restrict__ x, long n4, float lo, float hi, int* __restrict__ flag) {
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = ((const f32x4*)x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= !(v[e] >= lo && v[e] <= hi);
    }
    if (bad) atomicOr(flag, 1);
}
}  // namespace

extern "C" int lvae_range_flag_f32(const float* x, long n, float lo, float hi, int* flag, void* stream) {
    if (!x || !flag || n <= 0 || (n & 3)) return -22;
    const long n4 = n / 4;
    long bx = (n4 + 256 * 8 - 1) / (256 * 8);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(range_flag_kernel, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, x, n4, lo, hi, flag);
    return (int)hipGetLastError();
}

extern "C" int lvae_gemv_f32(const float* Wt, const float* b, const float* x, float* y, int N, int K, int gelu_in,
                             int gelu_out, void* stream) {
    if (!Wt || !b || !x || !y || N <= 0 || K <= 0 || (K & 3)) return -22;
    hipLaunchKernelGGL(gemv_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, Wt, b, x, y, N, K, gelu_in,
                       gelu_out);
    return (int)hipGetLastError();
}

extern "C" int lvae_prior_index_f32(const float* prm, float* pm, uint8_t* idx, const float* scale_table, int n_scales,
                                    float scale_bound, int B, int HW, int z, void* stream) {
    if (!prm || !pm || !idx || !scale_table || n_scales < 2 || n_scales > 256 || B <= 0 || HW <= 0 || z <= 0) return -22;
    const long total = (long)B * HW * z;
    hipLaunchKernelGGL(prior_index_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, prm, pm,
                       idx, scale_table, n_scales, scale_bound, total, HW, z);
    return (int)hipGetLastError();
}

extern "C" int lvae_quantize_f32(const float* qm, const float* pm, int32_t* sym, float* zhat, int B, int HW, int z, int ldz,
                                 void* stream) {
    if (!qm || !pm || !sym || !zhat || B <= 0 || HW <= 0 || z <= 0 || ldz < z) return -22;
    const long total = (long)B * HW * ldz;
    hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, qm, pm, sym,
                       zhat, total, HW, z, ldz);
    return (int)hipGetLastError();
}

extern "C" int lvae_dequantize_f32(const int32_t* sym, const float* pm, float* zhat, int B, int HW, int z, int ldz,
                                   void* stream) {
    if (!sym || !pm || !zhat || B <= 0 || HW <= 0 || z <= 0 || ldz < z) return -22;
    const long total = (long)B * HW * ldz;
    hipLaunchKernelGGL(dequantize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sym, pm,
                       zhat, total, HW, z, ldz);
    return (int)hipGetLastError();
}

extern "C" int lvae_gaussian_nll_f32(const float* prm, const int32_t* sym, double* out_nats, float scale_bound, int B, int HW,
                                     int z, int cdf_form, void* stream) {
    if (!prm || !sym || !out_nats || B <= 0 || HW <= 0 || z <= 0 || (cdf_form != 0 && cdf_form != 1)) return -22;
    const long per_image = (long)HW * z;
    long bx = (per_image + 256 * 4 - 1) / (256 * 4);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(gaussian_nll_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, prm, sym, out_nats,
                       scale_bound, per_image, HW, z, cdf_form);
    return (int)hipGetLastError();
}

extern "C" int lvae_bias_expand_f32(const float* bias, float* out, long M, int C, void* stream) {
    if (!bias || !out || M <= 0 || C <= 0 || (C & 3)) return -22;
    const long total4 = M * (C / 4);
    hipLaunchKernelGGL(bias_expand_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bias,
                       out, total4, C / 4);
    return (int)hipGetLastError();
}

namespace {
// deterministic form: block i writes its own partial sum (fixed thread -> element map, fixed in-block order, no atomics)
__global__ __launch_bounds__(256) void sqerr_partials_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             double* __restrict__ partials, long n) {
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float dlt = a[i] - b[i];
        s += (double)dlt * (double)dlt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
}  // namespace

extern "C" int lvae_sqerr_partials_f32(const float* a, const float* b, double* partials, int n_partials, long n, void* stream) {
    if (!a || !b || !partials || n_partials <= 0 || n <= 0) return -22;
    hipLaunchKernelGGL(sqerr_partials_kernel, dim3((unsigned)n_partials), dim3(256), 0, (hipStream_t)stream, a, b, partials, n);
    return (int)hipGetLastError();
}

extern "C" int lvae_sqerr_sum_f32(const float* a, const float* b, double* out, int B, long n_per_image, void* stream) {
    if (!a || !b || !out || B <= 0 || n_per_image <= 0) return -22;
    long bx = (n_per_image + 256 * 8 - 1) / (256 * 8);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(sqerr_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a, b, out,
                       n_per_image);
    return (int)hipGetLastError();
}

namespace {
__global__ void gelu_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = gelu_erf(x[i]);
}
}  // namespace

extern "C" int lvae_gelu_f32(const float* x, float* y, long n, void* stream) {
    if (!x || !y || n <= 0) return -22;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return (int)hipGetLastError();
}

extern "C" int lvae_prior_sample_f32(const float* prm, float* z, long M, int zdim, int ldz, float t, unsigned long long seed,
                                     unsigned long long offset, void* stream) {
    if (!prm || !z || M <= 0 || zdim <= 0 || ldz < zdim) return -22;
    const long total = M * ldz;
    hipLaunchKernelGGL(prior_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, prm, z, total,
                       zdim, ldz, t, (uint64_t)seed, (uint64_t)offset);
    return (int)hipGetLastError();
}

extern "C" int lvae_lossless_params_f32(const float* raw, const float* im, float* pm, uint8_t* idx, int32_t* sym, const float* table,
                                        int n_scales, float bound, int B, int H, int W, void* stream) {
    if (!raw || !pm || !idx || !table || n_scales <= 0 || n_scales > 256 || B <= 0 || H <= 0 || W <= 0 || (im && !sym)) return -22;
    const long total = (long)B * 3 * H * W;
    hipLaunchKernelGGL(lossless_params_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw, im, pm,
                       idx, sym, table, n_scales, bound, total, H * W);
    return (int)hipGetLastError();
}

extern "C" int lvae_lossless_output_f32(const int32_t* sym, const float* pm, float* out, long n, void* stream) {
    if (!sym || !pm || !out || n <= 0) return -22;
    hipLaunchKernelGGL(lossless_output_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sym, pm, out, n);
    return (int)hipGetLastError();
}

// ---- stream ordering helpers for launch plans that run independent branches on a side stream (lvae/engine.py: fork / join)
extern "C" void* lvae_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return (void*)e;
}
extern "C" int lvae_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : -22; }
// everything enqueued on `to_stream` after this call runs after everything enqueued on `from_stream` before it
extern "C" int lvae_stream_order(void* from_stream, void* to_stream, void* ev) {
    if (!ev) return -22;
    hipError_t e = hipEventRecord((hipEvent_t)ev, (hipStream_t)from_stream);
    if (e != hipSuccess) return (int)e;
    return (int)hipStreamWaitEvent((hipStream_t)to_stream, (hipEvent_t)ev, 0);
}

extern "C" int lvae_abi_version(void) { return 13; }
extern "C" const char* lvae_build_info(void) { return "liblvae_hip gfx950 (MI355X) fp32-MFMA; hipcc " __VERSION__; }
