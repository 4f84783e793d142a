// gemm_q8.hip -- the MLP GEMMs of the REDUCED-PRECISION mode (BASELINE.json configs[4]; prec 3 of lvae_gemm_f32) with BOTH operands
// already quantised to MX-fp8 in memory (lvae_gemm_desc.a_h2 with prec 3): fc1(y) and fc2(gelu(fc1)) of every ConvNeXt block, whose A
// operand has one consumer and whose producer (the depthwise+LayerNorm kernel, fc1's GELU epilogue) can emit e4m3 elements + E8M0 block
// scales directly -- 1.03 bytes per element instead of bf16's 2.
//
// Why.  gemm_lp_kernel (gemm_lp.hip) quantises A on its way into LDS: 2 500 VALU instructions per wave against 24 MFMAs, 0.25 of the
// HBM roof (docs/MEASUREMENT_HISTORY.md 5b).  This mode is HBM-bound by construction (a 4.6 PFLOP/s matrix pipe), so the main loop here does no arithmetic
// besides the MFMAs: global -> LDS by `buffer_load_dwordx4 ... lds` for data and scales, fragment reads, v_mfma_scale_f32_32x32x64_f8f6f4.
//
// Operand format Q8 of an [R][K] matrix (K % 64 == 0): R*K e4m3 bytes row-major, then the E8M0 scales as [K/64][R][2] (one pair per row
// and 64-deep stage, stage-major: the scales a tile needs for one stage are contiguous) -- lvae.models.base.pack_mxfp8_q8.
// LDS stage: (BM + BN) rows x 64 B lane-linear (the DMA image), 16-B pieces permuted by the SOURCE address (piece at position pp of stage
// row r holds logical piece pp ^ ((r >> 2) & 3): every ds_read_b128 lane group covers 16 distinct bank quads), + 1 KB of A scales +
// 1 KB of W scales.  NBUF stages, DMA NBUF - 1 ahead, one raw s_barrier per stage, counted vmcnt (see gemm_h2p.hip, same pipeline).
// Results do not depend on M, the batch or the tile shape (fixed k order, per-row quantisation upstream).
#include "gemm_common.h"

#include <type_traits>

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));

#define LVAE_FENCE() __builtin_amdgcn_sched_barrier(0)
#define Q8_DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define Q8_DSR8(dst, addr, off) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

__device__ __forceinline__ float q8_bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float q8_bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned short q8_f32_to_bf16(float x) {
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// GELU of the mode (gemm_lp.hip::lp_gelu: Abramowitz-Stegun 7.1.26 erf, |error| <= 1.5e-7) -- the same function, so that a layer gives
// the same values whichever of the two kernels runs it
__device__ __forceinline__ float q8_gelu(float x) {
    const float z = x * 0.70710678118654752440f, az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-az * az * 1.44269504088896340736f);
    return 0.5f * x * (1.0f + copysignf(fmaf(-p, e, 1.0f), z));
}

// Straight-line epilogue per case on a full-width tile (round 5; the recipe of gemm_common.h::h2p_epilogue_fast -- this mode is
// HBM-bound by construction, and the run-time case analysis below stored 4 bytes (re-quantised output) or 8 bytes (bf16 output) per lane
// behind an exec-mask branch per unit).  The case is a template parameter, rows >= M are dropped by the buffer descriptors' range
// check, and the stores are 16 B per lane:
//   Q8OUT: the four e4m3 dwords of a lane (rows 8 g + ., g = 0 .. 3, the same 4 columns) go through a 4 x 4 transpose ACROSS THE FOUR
//     QUADS of a 16-lane row (DPP row_shl / row_shr : 4 and : 8 with bank masks -- the quads are the DPP banks, no selects needed): the
//     lane of bank k then holds 16 consecutive columns of row 8 k + . : one store instead of four;
//   bf16: quad pairs exchange halves so that even quads hold 8 columns of row g and odd quads 8 columns of row g + 1.
// Per element the operations are the generic epilogue's (it stays below for ragged tiles): same values.
template <int TN, int EPI, bool Q8OUT>
__device__ __forceinline__ void q8_epilogue_fast(const lvae_gemm_desc& d, f32x16 (&acc)[2][TN], int m0, int n0, int rows_a, int wave_m, int wave_n,
                                                 int li, int lh) {
    const int lj = li & 3, bank = (li >> 2) & 3;
    constexpr bool HAS_RES = EPI == LVAE_EPI_GAMMA_RES || EPI == LVAE_EPI_RES;
    const int N = d.N, M = d.M;
    char* const outb = (char*)d.out;
    float cbias[TN], cgam[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + (wave_n * TN + b) * 32 + li;
        cbias[b] = d.bias ? d.bias[col] : 0.f;
        cgam[b] = EPI == LVAE_EPI_GAMMA_RES ? d.gamma[col] : 1.f;
    }
    if constexpr (Q8OUT) {
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(outb + (long)m0 * N), 0, rows_a * N, 0x00020000);
        // scale plane of this wave's 64-column block(s): [N / 64][M][2] bytes behind the data
        const int blk64 = (n0 + wave_n * TN * 32) >> 6;
        const __amdgpu_buffer_rsrc_t rsS =
            __builtin_amdgcn_make_buffer_rsrc((void*)(outb + (long)M * N + ((long)blk64 * M + m0) * 2), 0, rows_a * 2, 0x00020000);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            unsigned w[TN][4], ebs[TN][4];
#pragma unroll
            for (int b = 0; b < TN; ++b) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v0 = acc[a][b][4 * g + 0] + cbias[b], v1 = acc[a][b][4 * g + 1] + cbias[b];
                    float v2 = acc[a][b][4 * g + 2] + cbias[b], v3 = acc[a][b][4 * g + 3] + cbias[b];
                    if constexpr (EPI == LVAE_EPI_BIAS_GELU) { v0 = q8_gelu(v0); v1 = q8_gelu(v1); v2 = q8_gelu(v2); v3 = q8_gelu(v3); }
                    quad_transpose(v0, v1, v2, v3, lj);
                    float am = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3)));
                    am = fmaxf(am, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(am), (4 << 10) | 0x1f)));
                    am = fmaxf(am, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(am), (8 << 10) | 0x1f)));
                    am = fmaxf(am, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(am), (16 << 10) | 0x1f)));
                    const unsigned ab = __float_as_uint(am);
                    int eb = (int)((ab >> 23) & 0xffu) - 8;                   // the block-scale rule of gemm_lp.hip::lp_quant8 / pack_mxfp8
                    if ((ab & 0x7fffffu) > 0x600000u) eb += 1;
                    eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);
                    const float inv = __uint_as_float((unsigned)(254 - eb) << 23);
                    int ww = 0;
                    ww = __builtin_amdgcn_cvt_pk_fp8_f32(v0 * inv, v1 * inv, ww, false);
                    ww = __builtin_amdgcn_cvt_pk_fp8_f32(v2 * inv, v3 * inv, ww, true);
                    w[b][g] = (unsigned)ww;
                    ebs[b][g] = (unsigned)eb;
                }
            }
            // scales: one byte per row and 32-column block, written by the lanes of quad 0 (every lane of a row holds the block's value)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;
                if (li < 4) {
                    if constexpr (TN == 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(ebs[0][g] | (ebs[1][g] << 8)), rsS, r * 2, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b8((unsigned char)ebs[0][g], rsS, r * 2 + (wave_n & 1), 0, 0);
                }
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                // 4 x 4 transpose across the banks: afterwards the lane of bank k holds, in o0 .. o3, the dwords the banks 0 .. 3 held in w[k]
                unsigned x0 = w[b][0], x1 = w[b][1], x2 = w[b][2], x3 = w[b][3];
                const unsigned y1 = __builtin_amdgcn_update_dpp(x1, x0, 0x104, 0xF, 0x5, false);     // even banks: w1 <- (bank + 1).w0
                const unsigned y0 = __builtin_amdgcn_update_dpp(x0, x1, 0x114, 0xF, 0xA, false);     // odd banks:  w0 <- (bank - 1).w1
                const unsigned y3 = __builtin_amdgcn_update_dpp(x3, x2, 0x104, 0xF, 0x5, false);
                const unsigned y2 = __builtin_amdgcn_update_dpp(x2, x3, 0x114, 0xF, 0xA, false);
                const unsigned o2 = __builtin_amdgcn_update_dpp(y2, y0, 0x108, 0xF, 0x3, false);     // banks 0, 1: w2 <- (bank + 2).w0
                const unsigned o0 = __builtin_amdgcn_update_dpp(y0, y2, 0x118, 0xF, 0xC, false);     // banks 2, 3: w0 <- (bank - 2).w2
                const unsigned o3 = __builtin_amdgcn_update_dpp(y3, y1, 0x108, 0xF, 0x3, false);
                const unsigned o1 = __builtin_amdgcn_update_dpp(y1, y3, 0x118, 0xF, 0xC, false);
                const int r = (wave_m * 2 + a) * 32 + 4 * lh + 8 * bank + lj;
                const int col = n0 + (wave_n * TN + b) * 32 + (li & 16);
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){o0, o1, o2, o3}, rsO, r * N + col, 0, 0);
            }
        }
    } else {
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(outb + (long)m0 * d.ldo * 2), 0, rows_a * d.ldo * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(HAS_RES ? (const char*)d.res + (long)m0 * d.ldres * 2 : (const char*)d.out), 0, HAS_RES ? rows_a * d.ldres * 2 : 0, 0x00020000);
        const bool odd_bank = (bank & 1) != 0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            u32x2_t rv[4][TN];
            if constexpr (HAS_RES) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int r = (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        rv[g][b] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rsR, (r * d.ldres + n0 + (wave_n * TN + b) * 32 + (li & ~3)) * 2, 0, 0));
                }
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                u32x2_t q[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v0 = acc[a][b][4 * g + 0] + cbias[b], v1 = acc[a][b][4 * g + 1] + cbias[b];
                    float v2 = acc[a][b][4 * g + 2] + cbias[b], v3 = acc[a][b][4 * g + 3] + cbias[b];
                    if constexpr (EPI == LVAE_EPI_BIAS_GELU) { v0 = q8_gelu(v0); v1 = q8_gelu(v1); v2 = q8_gelu(v2); v3 = q8_gelu(v3); }
                    else if constexpr (EPI == LVAE_EPI_GAMMA_RES) { v0 *= cgam[b]; v1 *= cgam[b]; v2 *= cgam[b]; v3 *= cgam[b]; }
                    quad_transpose(v0, v1, v2, v3, lj);
                    if constexpr (HAS_RES) {
                        v0 += q8_bf16_lo(rv[g][b][0]); v1 += q8_bf16_hi(rv[g][b][0]); v2 += q8_bf16_lo(rv[g][b][1]); v3 += q8_bf16_hi(rv[g][b][1]);
                    }
                    q[g] = (u32x2_t){(unsigned)q8_f32_to_bf16(v0) | ((unsigned)q8_f32_to_bf16(v1) << 16),
                                     (unsigned)q8_f32_to_bf16(v2) | ((unsigned)q8_f32_to_bf16(v3) << 16)};
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    // even banks: {own, partner's} columns of row g; odd banks: {partner's, own} columns of row g + 1
                    const unsigned o2 = __builtin_amdgcn_update_dpp(q[g + 1][0], q[g][0], 0x104, 0xF, 0x5, false);
                    const unsigned o3 = __builtin_amdgcn_update_dpp(q[g + 1][1], q[g][1], 0x104, 0xF, 0x5, false);
                    const unsigned o0 = __builtin_amdgcn_update_dpp(q[g][0], q[g + 1][0], 0x114, 0xF, 0xA, false);
                    const unsigned o1 = __builtin_amdgcn_update_dpp(q[g][1], q[g + 1][1], 0x114, 0xF, 0xA, false);
                    const int r = (wave_m * 2 + a) * 32 + 4 * lh + 8 * (g + (odd_bank ? 1 : 0)) + lj;
                    const int col = n0 + (wave_n * TN + b) * 32 + (li & ~7);
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){o0, o1, o2, o3}, rsO, (r * d.ldo + col) * 2, 0, 0);
                }
            }
        }
    }
}

template <int WM, int TN, int NBUF>
__global__ __launch_bounds__(128 * WM, (WM == 4 ? 1 : 2)) void gemm_q8_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    constexpr int BM = 64 * WM, BN = 64 * TN, ROWS = BM + BN, DATA = ROWS * 64, STAGE = DATA + 2048;     // + A scales (1 KB) + W scales (1 KB)
    constexpr int NWAVE = 2 * WM, NG = ROWS / 16, NI = NG / NWAVE;
    static_assert(NG % NWAVE == 0, "whole data DMA instructions per wave");
    static_assert(NBUF * STAGE <= 160 * 1024 / (WM == 4 ? 1 : 2), "LDS");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = (char*)smem;                    // [NBUF][STAGE]
    int t;
    {
        const int b = blockIdx.x, q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int nq = d.K / 64;
    const int K = d.K, dM = d.M, dN = d.N;          // (locals: a select between two FIELDS of the by-value argument sends the whole struct to scratch)

    // ---- DMA side: data instruction g = i * NWAVE + wave covers stage rows 16g .. 16g + 15 (A rows first)
    const int rows_a = (d.M - m0) < BM ? (d.M - m0) : BM, rows_w = (d.N - n0) < BN ? (d.N - n0) : BN;
    const char* const Ab = (const char*)d.A0;
    const char* const Wb = (const char*)d.Wt16;
    // (descriptors are built at the point of use from scalar selects of base / size: a select between two DESCRIPTORS makes hipcc keep
    //  them in scratch and wrap every DMA in a waterfall loop)
    const char* const pA = Ab + (long)m0 * K;
    const char* const pW = Wb + (long)n0 * K;
    const char* const pSA = Ab + (long)dM * K + (long)m0 * 2;
    const char* const pSW = Wb + (long)dN * K + (long)n0 * 2;
    const int r_in = lane >> 2, pp = lane & 3;
    const int dvoff = r_in * K + ((pp ^ ((r_in >> 2) & 3)) << 4);
    auto dma = [&](int i, int stage, int buf) __attribute__((always_inline)) {
        const int g = i * NWAVE + wave;
        const bool isA = g < BM / 16;                                // uniform
        const int rowoff = (isA ? 16 * g : 16 * g - BM) * K;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(isA ? pA : pW), 0, (isA ? rows_a : rows_w) * K, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + buf * STAGE + g * 1024), 16, dvoff + rowoff,
                                                 stage * 64, 0, 0);
    };
    // scale pairs of the stage: wave 0 fetches the tile's A rows', every other wave the W rows' (the same bytes to the same place from each
    // of them: every wave then has the same number of vector-memory operations in flight, one vmcnt rule, and no three-way select --
    // which made hipcc keep the operands in scratch and wrap the DMA in a waterfall loop)
    auto dma_scales = [&](int stage, int buf) __attribute__((always_inline)) {
        const bool a = wave == 0;
        // (the stage's plane goes into the descriptor BASE: the scalar offset of a buffer instruction takes part in the range check, so a
        //  plane offset there would put every stage but the first out of range -- zeros, i.e. scale 2^-127; the check works on whole
        //  dwords: an odd number of rows would lose the last row's pair, so the size is rounded up -- the two bytes beyond belong to the
        //  next stage's plane of the same buffer)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((a ? pSA : pSW) + (long)stage * 2 * (a ? dM : dN)), 0,
                                                                              ((a ? rows_a : rows_w) * 2 + 3) & ~3, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + buf * STAGE + DATA + (a ? 0 : 1024)), 16, lane * 16,
                                                 0, 0, 0);
    };

    // ---- fragment side: lane (row i = li, half h = lh) of v_mfma_scale_f32_32x32x64_f8f6f4 holds k = 16 h + [0, 16) and 32 + 16 h +
    // [0, 16) of its row: logical pieces h and 2 + h of the row's four; the scale of k-block b comes from the lane with h = b
    const int xr = (li >> 2) & 3;
    const unsigned lbase = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds;
    const unsigned a_row = lbase + (wave_m * 64 + li) * 64, b_row = lbase + (BM + wave_n * 32 * TN + li) * 64;
    const unsigned a_d0 = a_row + ((lh ^ xr) << 4), a_d1 = a_row + (((2 + lh) ^ xr) << 4);
    const unsigned b_d0 = b_row + ((lh ^ xr) << 4), b_d1 = b_row + (((2 + lh) ^ xr) << 4);
    const unsigned a_sc = lbase + DATA + (wave_m * 64 + li) * 2 + lh, b_sc = lbase + DATA + 1024 + (wave_n * 32 * TN + li) * 2 + lh;

    f32x16 acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s) {
#pragma unroll
        for (int i = 0; i < NI; ++i) dma(i, s < nq ? s : nq - 1, s);
        dma_scales(s < nq ? s : nq - 1, s);
    }

    auto stage_body = [&](auto buf_tag, int s) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value, NXT = (BUF + NBUF - 1) % NBUF;
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NBUF - 2) * (NI + 1)) : "memory");
        LVAE_FENCE();
        u32x4 af[2][2], bf[TN][2];
        unsigned sa[2], sb[TN];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            Q8_DSR128(af[a][0], a_d0 + BUF * STAGE, a * 2048);
            Q8_DSR128(af[a][1], a_d1 + BUF * STAGE, a * 2048);
            Q8_DSR8(sa[a], a_sc + BUF * STAGE, a * 64);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            Q8_DSR128(bf[b][0], b_d0 + BUF * STAGE, b * 2048);
            Q8_DSR128(bf[b][1], b_d1 + BUF * STAGE, b * 2048);
            Q8_DSR8(sb[b], b_sc + BUF * STAGE, b * 64);
        }
        const int sn = s + NBUF - 1 < nq ? s + NBUF - 1 : nq - 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) dma(i, sn, NXT);
        dma_scales(sn, NXT);
        if constexpr (TN == 2)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bf[0][0]), "+v"(bf[0][1]),
                         "+v"(bf[1][0]), "+v"(bf[1][1]), "+v"(sa[0]), "+v"(sa[1]), "+v"(sb[0]), "+v"(sb[1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bf[0][0]), "+v"(bf[0][1]),
                         "+v"(sa[0]), "+v"(sa[1]), "+v"(sb[0]));
        LVAE_FENCE();
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const i32x8 bv = {(int)bf[b][0][0], (int)bf[b][0][1], (int)bf[b][0][2], (int)bf[b][0][3],
                              (int)bf[b][1][0], (int)bf[b][1][1], (int)bf[b][1][2], (int)bf[b][1][3]};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const i32x8 av = {(int)af[a][0][0], (int)af[a][0][1], (int)af[a][0][2], (int)af[a][0][3],
                                  (int)af[a][1][0], (int)af[a][1][1], (int)af[a][1][2], (int)af[a][1][3]};
                acc[a][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[a][b], 0, 0, 0, (int)sa[a], 0, (int)sb[b]);
            }
        }
        LVAE_FENCE();
    };
    for (int s = 0; s < nq; s += NBUF) {
        stage_body(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nq) stage_body(std::integral_constant<int, 1 % NBUF>{}, s + 1);
        if (NBUF > 2 && s + 2 < nq) stage_body(std::integral_constant<int, 2 % NBUF>{}, s + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  Per-column ops, DPP
    // quad transpose (the lane then owns 4 consecutive COLUMNS of one row), then: out_h2 = the result re-quantised to Q8 for the next GEMM
    // (block amax over the 8 lanes that share a row and a 32-column block: ds_swizzle xor 4 / 8 / 16), else bf16 rows.
#ifndef Q8_EXP_GENERIC_EPI
    if (n0 + BN <= d.N) {                                                   // uniform: full-width tile -> a straight-line form
        if (d.out_h2) {
            if (d.epi == LVAE_EPI_BIAS_GELU) { q8_epilogue_fast<TN, LVAE_EPI_BIAS_GELU, true>(d, acc, m0, n0, rows_a, wave_m, wave_n, li, lh); return; }
            if (d.epi == LVAE_EPI_BIAS) { q8_epilogue_fast<TN, LVAE_EPI_BIAS, true>(d, acc, m0, n0, rows_a, wave_m, wave_n, li, lh); return; }
        } else if (!((d.ldo | d.ldres) & 7)) {
            if (d.epi == LVAE_EPI_GAMMA_RES) { q8_epilogue_fast<TN, LVAE_EPI_GAMMA_RES, false>(d, acc, m0, n0, rows_a, wave_m, wave_n, li, lh); return; }
            if (d.epi == LVAE_EPI_RES) { q8_epilogue_fast<TN, LVAE_EPI_RES, false>(d, acc, m0, n0, rows_a, wave_m, wave_n, li, lh); return; }
        }
    }
#endif
    const int epi = d.epi, lj = li & 3;
    const bool has_res = epi == LVAE_EPI_GAMMA_RES || epi == LVAE_EPI_RES;
    char* const outb = (char*)d.out;
    unsigned char* const osc = (unsigned char*)outb + (long)d.M * d.N;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int colb = n0 + (wave_n * TN + b) * 32, col = colb + li, cc = col < d.N ? col : 0;
        const float cbias = d.bias ? d.bias[cc] : 0.f, cgam = (epi == LVAE_EPI_GAMMA_RES) ? d.gamma[cc] : 1.f;
        const int c4 = colb + (li & ~3);
        const bool cok = c4 < d.N;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = m0 + (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;
                const bool ok = row < d.M && cok;
                float v0 = acc[a][b][4 * g + 0] + cbias, v1 = acc[a][b][4 * g + 1] + cbias;
                float v2 = acc[a][b][4 * g + 2] + cbias, v3 = acc[a][b][4 * g + 3] + cbias;
                if (epi == LVAE_EPI_BIAS_GELU) { v0 = q8_gelu(v0); v1 = q8_gelu(v1); v2 = q8_gelu(v2); v3 = q8_gelu(v3); }
                else if (epi == LVAE_EPI_GAMMA_RES) { v0 *= cgam; v1 *= cgam; v2 *= cgam; v3 *= cgam; }
                quad_transpose(v0, v1, v2, v3, lj);
                if (d.out_h2) {
                    float am = ok ? fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3))) : 0.f;
                    am = fmaxf(am, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(am), (4 << 10) | 0x1f)));
                    am = fmaxf(am, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(am), (8 << 10) | 0x1f)));
                    am = fmaxf(am, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(am), (16 << 10) | 0x1f)));
                    const unsigned ab = __float_as_uint(am);
                    int eb = (int)((ab >> 23) & 0xffu) - 8;                   // the block-scale rule of gemm_lp.hip::lp_quant8 / pack_mxfp8
                    if ((ab & 0x7fffffu) > 0x600000u) eb += 1;
                    eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);
                    const float inv = __uint_as_float((unsigned)(254 - eb) << 23);
                    int w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v0 * inv, v1 * inv, w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v2 * inv, v3 * inv, w, true);
                    if (ok) {
                        *(int*)(outb + (long)row * d.N + c4) = w;
                        if (li < 4) osc[((long)(c4 >> 6) * d.M + row) * 2 + ((c4 >> 5) & 1)] = (unsigned char)eb;
                    }
                } else if (ok) {
                    f32x4 o = {v0, v1, v2, v3};
                    if (has_res) {
                        const u32x2_t q = *(const u32x2_t*)((const unsigned short*)d.res + (long)row * d.ldres + c4);
                        o[0] += q8_bf16_lo(q[0]); o[1] += q8_bf16_hi(q[0]); o[2] += q8_bf16_lo(q[1]); o[3] += q8_bf16_hi(q[1]);
                    }
                    const u16x4_t qo = {q8_f32_to_bf16(o[0]), q8_f32_to_bf16(o[1]), q8_f32_to_bf16(o[2]), q8_f32_to_bf16(o[3])};
                    *(u16x4_t*)((unsigned short*)d.out + (long)row * d.ldo + c4) = qo;
                }
            }
        }
    }
}

template <int WM, int TN, int NBUF>
int launch_q8(const lvae_gemm_desc* d, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * TN, LDS = NBUF * ((BM + BN) * 64 + 2048);
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)gemm_q8_kernel<WM, TN, NBUF>, LDS)) return ae;
    const int tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN, n_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_q8_kernel<WM, TN, NBUF>), dim3(n_tiles), dim3(128 * WM), LDS, st, *d, tiles_n, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace

// prec 3, a_h2 = 1 (called by lvae_gemm_f32).  cfg = 10 WM + TN forces a tile (42 22 21); every choice gives the same bits.
int lvae_gemm_q8_dispatch(const lvae_gemm_desc* d, hipStream_t st) {
    if (!d->Wt16 || d->a_mode != LVAE_A_PLAIN || d->K1 != 0 || d->K0 != d->K || (d->K & 63) || d->lda0 != d->K || d->ldw != d->K || d->a_gelu ||
        d->ksplit > 1 || d->store != LVAE_ST_ROWMAJOR || (d->N & 3) || (long)256 * d->K > 0x7fffffffL)
        return -22;
    if (d->out_h2 ? ((d->N & 63) || d->ldo != d->N || (d->epi != LVAE_EPI_BIAS && d->epi != LVAE_EPI_BIAS_GELU)) : (!d->out_bf16 || (d->ldo & 3)))
        return -22;
    if ((d->epi == LVAE_EPI_GAMMA_RES || d->epi == LVAE_EPI_RES) && (!d->res || (d->ldres & 3))) return -22;
    const int M = d->M, N = d->N;
    int sel = d->cfg;
    if (sel != 42 && sel != 22 && sel != 21) {
        const int tn = (N % 128 == 0 || ((N + 127) / 128) * 128 - N < ((N + 63) / 64) * 64 - N + 1) ? 2 : 1;
        // by measurement (tools/q8_tiles.py, profiles/r03_gemm_q8_tile_sweep.txt): 64-wide tiles (N = 192: three, none half empty) exist as
        // 128 x 64 only; 128 x 128 is never the fastest; wide-output launches (fc1: N > K, epilogue-heavy) take 256 x 128 only when very
        // large, deep-K launches (fc2) as soon as there are enough 256-row tiles
        const long t256 = (long)((M + 255) / 256) * ((N + 127) / 128), t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        sel = tn == 1 ? 21 : (N > d->K ? (t128 >= 4096 ? 42 : 21) : (t256 >= 128 ? 42 : 21));
    }
    switch (sel) {
        case 42: return launch_q8<4, 2, 3>(d, st);
        case 22: return launch_q8<2, 2, 3>(d, st);
        default: return launch_q8<2, 1, 3>(d, st);
    }
}
