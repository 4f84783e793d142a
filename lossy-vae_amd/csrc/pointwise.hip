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

// bf16-storage form of the sliding-window kernel (reached for the two-affine case only), one output row per group
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
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ im, const float* __restrict__ wt, const float* __restrict__ bias,
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
    // four pixels at a time: four independent 48-FMA chains per thread (one chain alone runs at the FMA's latency, not its rate); per
    // output the chain is unchanged -- bias, then k ascending -- so the bits are (round 6: 35 -> 29 us for one 512x768 image)
    for (int p = 0; p < 64; p += 4) {
        if (p0 + p >= M) break;
        float a[4] = {bv, bv, bv, bv};
#pragma unroll
        for (int k4 = 0; k4 < 12; ++k4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 pv = *(const f32x4*)&patch[p + q][k4 * 4];
                a[q] = fmaf(pv[0], wr[k4 * 4 + 0], a[q]);
                a[q] = fmaf(pv[1], wr[k4 * 4 + 1], a[q]);
                a[q] = fmaf(pv[2], wr[k4 * 4 + 2], a[q]);
                a[q] = fmaf(pv[3], wr[k4 * 4 + 3], a[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long pg = p0 + p + q;
            if (pg < M) {
                if (BF) ((unsigned short*)out)[pg * Cout + n] = (unsigned short)f2bf(a[q]);
                else ((float*)out)[pg * Cout + n] = a[q];
            }
        }
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
                                       int n_scales, float bound, long total, int HW, int* __restrict__ status) {
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
    if (status && !(fabsf(r[c]) <= 3.4028234664e38f && fabsf(r[3 + c]) <= 3.4028234664e38f)) atomicOr(status, LVAE_STATUS_NONFINITE_PRIOR);
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

__global__ void lossless_output_kernel(const int32_t* __restrict__ sym, const float* __restrict__ pm, float* __restrict__ out, long n,
                                       int* __restrict__ status) {
#pragma clang fp contract(off)
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float x = (float)sym[e] + pm[e];
    if (status && !(fabsf(x) <= 3.4028234664e38f)) atomicOr(status, LVAE_STATUS_NONFINITE_IMAGE);
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
// The coder's arrays (scale indexes, symbols) are NCHW rasters -- per image and channel HW consecutive entries -- while the maps they
// come from / go to are pixel-major ([pixel][channel]).  One thread per (pixel, channel) writing idx[(b z + c) HW + p] scatters single
// bytes / dwords HW entries apart; since round 5 these arrays may BE the host's pinned buffers (lvae_dec_block / lvae_enc_block with
// idx_dev / sym_dev = NULL: the kernels write / read host memory over the link, no blit launch and no device copy on the decode
// chain), where every scattered access would be a bus transaction of its own.  So: a workgroup owns 64 consecutive pixels of one image,
// works through them pixel-major (coalesced on the map side), transposes 64 pixels x <= 16 channels (blockIdx.y: the channel chunk) through LDS and touches the
// raster in runs of 64 consecutive entries per channel.  Per element the arithmetic is unchanged (same bits).
constexpr int CT_PIX = 64, CT_CH = 16, CT_LD = CT_PIX + 1;      // (16 channels per workgroup: 4 elements per thread -- enough workgroups to fill the chip on the stride-16 maps)

// SK: the prior parameters are first formed from the S partial-sum planes of a split-K GEMM whose reduce pass was deferred (ws, plane =
// floats per plane, bias): ((ws[0] + ws[1]) + ...) + bias in slice order -- splitk_reduce_chunk's operations, so the bits of the reduce
// launch this replaces -- and written to prm for the other consumers.
template <bool SK>
__global__ __launch_bounds__(256) void prior_index_kernel(const float* __restrict__ prm_in, float* __restrict__ pm, uint8_t* __restrict__ idx,
                                                          const float* __restrict__ table, int n_scales, float bound, int HW, int z,
                                                          int* __restrict__ status, const float* __restrict__ ws, int S, long plane,
                                                          const float* __restrict__ bias, float* __restrict__ prm_out) {
#pragma clang fp contract(off)
    __shared__ int tile[CT_CH * CT_LD];
    __shared__ float tab[256];                         // the scale table (n_scales <= 256: checked by the launcher): the binary search below is
    if ((int)threadIdx.x < n_scales) tab[threadIdx.x] = table[threadIdx.x];     // six DEPENDENT loads per element -- from LDS, not from L2
    __syncthreads();
    const int b = blockIdx.z, p0 = blockIdx.x * CT_PIX, np = (HW - p0) < CT_PIX ? (HW - p0) : CT_PIX;
    const long m0 = (long)b * HW + p0;
    {
        const int c0 = blockIdx.y * CT_CH;
        const int zc = (z - c0) < CT_CH ? (z - c0) : CT_CH;
        for (int i = threadIdx.x; i < np * zc; i += 256) {
            const int pl = i / zc, c = c0 + (i - pl * zc);
            const long m = m0 + pl;
            float mean, lv;
            if constexpr (SK) {
                const float* w = ws + m * 2 * z + c;
                mean = w[0]; lv = w[z];
                for (int sl = 1; sl < S; ++sl) { mean += w[sl * plane]; lv += w[sl * plane + z]; }
                mean += bias[c]; lv += bias[z + c];
                prm_out[m * 2 * z + c] = mean; prm_out[m * 2 * z + z + c] = lv;
            } else {
                mean = prm_in[m * 2 * z + c];
                lv = prm_in[m * 2 * z + z + c];
            }
            // a NaN / inf prior parameter (an fp16 overflow of the f16x2 arithmetic upstream, include/lvae_hip.h "status word") would become
            // index 0 / 63 and a NaN mean silently: report it (one atomic per wave that saw one)
            if (status && !(fabsf(mean) <= 3.4028234664e38f && fabsf(lv) <= 3.4028234664e38f)) atomicOr(status, LVAE_STATUS_NONFINITE_PRIOR);
            // softplus(x + 2.3) - 2.3  (torch: beta=1, threshold=20)
            const float xs = lv + 2.3f;
            const float sp = xs > 20.0f ? xs : log1pf(expf(xs));
            const float pv = expf(sp - 2.3f);
            const float sc = fmaxf(pv, bound);
            // idx = #{i < n_scales-1 : table[i] < s}   (binary search, table ascending)
            int lo = 0, hi = n_scales - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (tab[mid] < sc) lo = mid + 1; else hi = mid;
            }
            pm[m * z + c] = mean;
            tile[(c - c0) * CT_LD + pl] = lo;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < zc * CT_PIX; j += 256) {
            const int cl = j / CT_PIX, pl = j - cl * CT_PIX;
            if (pl < np) idx[((long)b * z + c0 + cl) * HW + p0 + pl] = (uint8_t)tile[cl * CT_LD + pl];
        }
        __syncthreads();
    }
}

// SK: the posterior mean is first formed from the S planes of a split-K GEMM whose reduce pass was deferred (as prior_index_kernel<true>)
// and written to qm_out for the other consumers.
template <bool SK>
__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ qm, const float* __restrict__ pm, int32_t* __restrict__ sym,
                                                       float* __restrict__ zhat, int HW, int z, int ldz, int* __restrict__ status,
                                                       const float* __restrict__ ws, int S, long plane, const float* __restrict__ bias,
                                                       float* __restrict__ qm_out) {
#pragma clang fp contract(off)
    __shared__ int tile[CT_CH * CT_LD];
    const int b = blockIdx.z, p0 = blockIdx.x * CT_PIX, np = (HW - p0) < CT_PIX ? (HW - p0) : CT_PIX;
    const long m0 = (long)b * HW + p0;
    {
        const int c0 = blockIdx.y * CT_CH;                             // over the PADDED rows (columns z .. ldz - 1 of zhat are zeroed)
        const int zc = (ldz - c0) < CT_CH ? (ldz - c0) : CT_CH;
        for (int i = threadIdx.x; i < np * zc; i += 256) {
            const int pl = i / zc, c = c0 + (i - pl * zc);
            const long m = m0 + pl;
            if (c >= z) { zhat[m * ldz + c] = 0.f; continue; }
            const float mu = pm[m * z + c];
            float q;
            if constexpr (SK) {
                const float* w = ws + m * z + c;
                q = w[0];
                for (int sl = 1; sl < S; ++sl) q += w[sl * plane];
                q += bias[c];
                qm_out[m * z + c] = q;
            } else {
                q = qm[m * z + c];
            }
            const float r = rintf(q - mu);  // v_rndne_f32: round-half-to-even == torch.round
            if (status && !(fabsf(r) < 2147483648.0f)) atomicOr(status, LVAE_STATUS_NONFINITE_LATENT);      // NaN / inf / no int32 symbol
            zhat[m * ldz + c] = r + mu;
            tile[(c - c0) * CT_LD + pl] = (int32_t)r;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < zc * CT_PIX; j += 256) {
            const int cl = j / CT_PIX, pl = j - cl * CT_PIX;
            if (pl < np && c0 + cl < z) sym[((long)b * z + c0 + cl) * HW + p0 + pl] = tile[cl * CT_LD + pl];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void dequantize_kernel(const int32_t* __restrict__ sym, const float* __restrict__ pm, float* __restrict__ zhat,
                                                         int HW, int z, int ldz) {
    __shared__ int tile[CT_CH * CT_LD];
    const int b = blockIdx.z, p0 = blockIdx.x * CT_PIX, np = (HW - p0) < CT_PIX ? (HW - p0) : CT_PIX;
    const long m0 = (long)b * HW + p0;
    {
        const int c0 = blockIdx.y * CT_CH;
        const int zc = (ldz - c0) < CT_CH ? (ldz - c0) : CT_CH;
        for (int j = threadIdx.x; j < zc * CT_PIX; j += 256) {
            const int cl = j / CT_PIX, pl = j - cl * CT_PIX;
            if (pl < np && c0 + cl < z) tile[cl * CT_LD + pl] = sym[((long)b * z + c0 + cl) * HW + p0 + pl];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < np * zc; i += 256) {
            const int pl = i / zc, c = c0 + (i - pl * zc);
            const long m = m0 + pl;
            zhat[m * ldz + c] = c >= z ? 0.f : (float)tile[(c - c0) * CT_LD + pl] + pm[m * z + c];
        }
        __syncthreads();
    }
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

// dwconv_cl.hip: the channel-per-lane form (weights in registers, LDS-DMA row buffers) takes the problem by (C, k) alone; what follows
// here is the sliding-window kernel for the other channel counts (qres17m: C = 144 / 288) and the two-affine case
int lvae_dwln_cl_try(const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b, const float* shift,
                     const float* scale1p, void* y, int B, int H, int W, int C, int k, int fmt, hipStream_t st, int* rc);

extern "C" int lvae_dwconv_ln_f32(const float* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                                  const float* shift, const float* scale1p, float* y, int B, int H, int W, int C, int k,
                                  void* stream) {
    if (!x || !wt || !bias || !y || B <= 0 || H <= 0 || W <= 0) return -22;
    if ((ln_w == nullptr) != (ln_b == nullptr) || (shift == nullptr) != (scale1p == nullptr)) return -22;
    {
        int rc = 0;
        if (lvae_dwln_cl_try(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, C, k, 0, (hipStream_t)stream, &rc)) return rc;
    }
#ifdef LVAE_EXPERIMENTAL_BUILD           // tile-height sweep hook (tools/build_exp.sh copies only)
    static bool env_read = false;
    if (!env_read) {
        const char* e = getenv("LVAE_DW_TH"); if (e) g_dw_th = atoi(e);
        env_read = true;
    }
#endif
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: return dispatch_dwln_c<1>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 3: return dispatch_dwln_c<3>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 5: return dispatch_dwln_c<5>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
        case 7: return dispatch_dwln_c<7>(C, x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, st);
    }
    return -22;
}

extern "C" int lvae_dwconv_ln_h2(const float* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                                 const float* shift, const float* scale1p, void* y, int B, int H, int W, int C, int k, void* stream) {
    if (!x || !wt || !bias || !y || B <= 0 || H <= 0 || W <= 0) return -22;
    if ((ln_w == nullptr) != (ln_b == nullptr) || (shift == nullptr) != (scale1p == nullptr)) return -22;
    int rc = 0;
    return lvae_dwln_cl_try(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, C, k, 2, (hipStream_t)stream, &rc) ? rc : -22;
}

extern "C" int lvae_dwconv_ln_q8(const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                                 const float* shift, const float* scale1p, void* y, int B, int H, int W, int C, int k, void* stream) {
    if (!x || !wt || !bias || !y || B <= 0 || H <= 0 || W <= 0) return -22;
    if ((ln_w == nullptr) != (ln_b == nullptr) || (shift == nullptr) != (scale1p == nullptr)) return -22;
    int rc = 0;
    return lvae_dwln_cl_try(x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, C, k, 3, (hipStream_t)stream, &rc) ? rc : -22;
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
                                    float scale_bound, int B, int HW, int z, int* status, void* stream) {
    if (!prm || !pm || !idx || !scale_table || n_scales < 2 || n_scales > 256 || B <= 0 || B > 65535 || HW <= 0 || z <= 0) return -22;
    hipLaunchKernelGGL(prior_index_kernel<false>, dim3((unsigned)((HW + CT_PIX - 1) / CT_PIX), (unsigned)((z + CT_CH - 1) / CT_CH), (unsigned)B), dim3(256), 0, (hipStream_t)stream, prm, pm,
                       idx, scale_table, n_scales, scale_bound, HW, z, status, (const float*)nullptr, 0, 0L, (const float*)nullptr, (float*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int lvae_prior_index_sk_f32(const float* ws, int S, const float* bias, float* prm, float* pm, uint8_t* idx, const float* scale_table,
                                       int n_scales, float scale_bound, int B, int HW, int z, int* status, void* stream) {
    if (!ws || S < 2 || !bias || !prm || !pm || !idx || !scale_table || n_scales < 2 || n_scales > 256 || B <= 0 || B > 65535 || HW <= 0 || z <= 0) return -22;
    hipLaunchKernelGGL(prior_index_kernel<true>, dim3((unsigned)((HW + CT_PIX - 1) / CT_PIX), (unsigned)((z + CT_CH - 1) / CT_CH), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const float*)nullptr, pm, idx, scale_table, n_scales, scale_bound, HW, z, status, ws, S, (long)B * HW * 2 * z, bias, prm);
    return (int)hipGetLastError();
}

extern "C" int lvae_quantize_f32(const float* qm, const float* pm, int32_t* sym, float* zhat, int B, int HW, int z, int ldz,
                                 int* status, void* stream) {
    if (!qm || !pm || !sym || !zhat || B <= 0 || B > 65535 || HW <= 0 || z <= 0 || ldz < z) return -22;
    hipLaunchKernelGGL(quantize_kernel<false>, dim3((unsigned)((HW + CT_PIX - 1) / CT_PIX), (unsigned)((ldz + CT_CH - 1) / CT_CH), (unsigned)B), dim3(256), 0, (hipStream_t)stream, qm, pm, sym,
                       zhat, HW, z, ldz, status, (const float*)nullptr, 0, 0L, (const float*)nullptr, (float*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int lvae_quantize_sk_f32(const float* ws, int S, const float* bias, float* qm, const float* pm, int32_t* sym, float* zhat, int B, int HW,
                                    int z, int ldz, int* status, void* stream) {
    if (!ws || S < 2 || !bias || !qm || !pm || !sym || !zhat || B <= 0 || B > 65535 || HW <= 0 || z <= 0 || ldz < z) return -22;
    hipLaunchKernelGGL(quantize_kernel<true>, dim3((unsigned)((HW + CT_PIX - 1) / CT_PIX), (unsigned)((ldz + CT_CH - 1) / CT_CH), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const float*)nullptr, pm, sym, zhat, HW, z, ldz, status, ws, S, (long)B * HW * z, bias, qm);
    return (int)hipGetLastError();
}

extern "C" int lvae_dequantize_f32(const int32_t* sym, const float* pm, float* zhat, int B, int HW, int z, int ldz,
                                   void* stream) {
    if (!sym || !pm || !zhat || B <= 0 || B > 65535 || HW <= 0 || z <= 0 || ldz < z) return -22;
    hipLaunchKernelGGL(dequantize_kernel, dim3((unsigned)((HW + CT_PIX - 1) / CT_PIX), (unsigned)((ldz + CT_CH - 1) / CT_CH), (unsigned)B), dim3(256), 0, (hipStream_t)stream, sym, pm,
                       zhat, HW, z, ldz);
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
                                        int n_scales, float bound, int B, int H, int W, int* status, void* stream) {
    if (!raw || !pm || !idx || !table || n_scales <= 0 || n_scales > 256 || B <= 0 || H <= 0 || W <= 0 || (im && !sym)) return -22;
    const long total = (long)B * 3 * H * W;
    hipLaunchKernelGGL(lossless_params_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw, im, pm,
                       idx, sym, table, n_scales, bound, total, H * W, status);
    return (int)hipGetLastError();
}

extern "C" int lvae_lossless_output_f32(const int32_t* sym, const float* pm, float* out, long n, int* status, void* stream) {
    if (!sym || !pm || !out || n <= 0) return -22;
    hipLaunchKernelGGL(lossless_output_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sym, pm, out, n, status);
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

extern "C" int lvae_abi_version(void) { return 24; }
extern "C" const char* lvae_build_info(void) { return "liblvae_hip gfx950 (MI355X) fp32-MFMA; hipcc " __VERSION__; }
