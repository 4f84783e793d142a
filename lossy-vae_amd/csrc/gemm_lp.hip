// gemm_lp.hip -- the channel-mixing GEMM of the REDUCED-PRECISION mode (BASELINE.json configs[4]: "bf16 activations + fp8 MFMA for
// 1x1 convs"), prec 3 of lvae_gemm_f32.  NOT a parity path: results differ visibly from the reference's fp32; encoder and decoder
// stay bit-consistent because both run exactly these kernels (fixed k order, per-row quantisation: nothing depends on M, the batch
// size or the tile shape).
//
//   out[m][n] = epilogue( sum_k q(A[m][k]) * q(W[n][k]) + bias[n] )
//
//  * Activations live in HBM as bf16 (NHWC rows); a few small operands (z_hat of the latent blocks) are fp32: `a_bf16` says which.
//  * q() = OCP MX-fp8: e4m3 elements with one E8M0 (power-of-two) scale per 32 consecutive k of a row (the block-scaled format of
//    v_mfma_scale_f32_32x32x64_f8f6f4, ~5 PFLOP/s dense).  Weights are quantised once on the host (lvae.models.base.pack_mxfp8);
//    activations are quantised on their way into LDS: 8 values per lane, block amax by a DPP quad reduction, scale = 2^e with
//    amax / 2^e in (224, 448], v_cvt_pk_fp8_f32.  fp32 accumulation inside the MFMA and across k.
//  * The mode is HBM-bound (half the activation bytes of the fp32 path, a matrix pipe 12x faster than bf16x3): the kernel is a plain
//    double-buffered LDS pipeline -- 128 x (64 TN) tiles, 4 waves (2 x 2), 64-deep stages, one barrier per stage, two workgroups per
//    CU -- whose job is to keep 16-B loads in flight; 80-byte LDS rows (64 B of fp8 + 2 scale bytes + pad) make the 16-lane
//    ds_read_b128 groups of the fragment reads conflict-free.
//  * Same fused gathers (torch.cat operand, 2x2 patch, 3x3 taps), epilogues (bias / exact-erf GELU / layer-scale + residual /
//    residual) and stores (row-major, PixelShuffle, final NCHW image) as the fp32 kernels; outputs bf16 or fp32 (`out_bf16`);
//    residual rows have the output's type.
#include "gemm_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned short f32_to_bf16(float x) {          // round to nearest even (no NaN inputs on this path)
    const unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float dpp_max_xor1(float x) {
    return fmaxf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true)));
}
__device__ __forceinline__ float dpp_max_xor2(float x) {
    return fmaxf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true)));
}

constexpr int LP_ROWB = 80;      // LDS row: 64 B fp8 (k 0..63 of the stage) | scale byte k0-31 | scale byte k32-63 | pad

// Address of the 8 consecutive k-values [k, k+8) of row m (element units), and whether they are real data (else they read as 0).
struct LpRow {
    const char* p0;       // PLAIN: A0 row; PATCH2/CONV3: the row's own pixel
    const char* p1;       // PLAIN: A1 row - K0 elements
    int h, w;
};

template <int AMODE>
__device__ __forceinline__ LpRow lp_row(const lvae_gemm_desc& d, int m, int esz) {
    LpRow r;
    r.h = r.w = 0;
    m = m < d.M ? m : d.M - 1;
    if (AMODE == LVAE_A_PLAIN) {
        r.p0 = (const char*)d.A0 + (long)m * d.lda0 * esz;
        r.p1 = d.A1 ? (const char*)d.A1 + ((long)m * d.lda1 - d.K0) * esz : r.p0;
    } else if (AMODE == LVAE_A_PATCH2) {
        const int w = m % d.W, bh = m / d.W;
        r.p0 = (const char*)d.A0 + ((long)bh * 2 * (2L * d.W) + 2L * w) * d.K0 * esz;
        r.p1 = r.p0;
    } else {
        const int w = m % d.W, bh = m / d.W;
        r.w = w; r.h = bh % d.H;
        r.p0 = (const char*)d.A0 + (long)m * d.K0 * esz;
        r.p1 = r.p0;
    }
    return r;
}

template <int AMODE>
__device__ __forceinline__ const char* lp_addr(const lvae_gemm_desc& d, const LpRow& r, int k, int esz, bool& ok) {
    ok = k < d.K;
    const int kc = ok ? k : 0;
    if (AMODE == LVAE_A_PLAIN) {
        return ((kc < d.K0) ? r.p0 : r.p1) + (long)kc * esz;
    } else if (AMODE == LVAE_A_PATCH2) {
        const int seg = 2 * d.K0, s = kc / seg, kk = kc - s * seg;
        return r.p0 + ((long)s * (2L * d.W) * d.K0 + kk) * esz;
    } else {
        const int s = kc / d.K0, kk = kc - s * d.K0;
        const int di = s / 3 - 1, dj = s - (s / 3) * 3 - 1;
        const int hh = r.h + di, ww = r.w + dj;
        const bool in = (hh >= 0) && (hh < d.H) && (ww >= 0) && (ww < d.W);
        ok = ok && in;
        const long tap = in ? ((long)di * d.W + dj) * d.K0 : 0;
        return r.p0 + (tap + kk) * esz;
    }
}

// quantise 8 floats (one quarter of a 32-element MX block: the other three quarters sit in the other lanes of the quad) to e4m3 with
// the block's shared power-of-two scale.  Returns the two packed dwords and the E8M0 scale byte.
__device__ __forceinline__ void lp_quant8(const float (&v)[8], u32x2& packed, unsigned& scale_byte) {
    float am = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))),
                     fmaxf(fmaxf(fabsf(v[4]), fabsf(v[5])), fmaxf(fabsf(v[6]), fabsf(v[7]))));
    am = dpp_max_xor1(am);
    am = dpp_max_xor2(am);
    const unsigned ab = __float_as_uint(am);
    int eb = (int)((ab >> 23) & 0xffu) - 8;                      // biased exponent of 2^(floor(log2 amax) - 8): amax / scale in [256, 512)
    if ((ab & 0x7fffffu) > 0x600000u) eb += 1;                   // mantissa > 1.75: would exceed e4m3's 448 -> one more step
    eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);                     // E8M0: 2^(eb - 127); all-zero (or tiny) blocks get 2^-126
    const float inv = __uint_as_float((unsigned)(254 - eb) << 23);      // 2^(127 - eb), exact
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, w1, true);
    packed[0] = (unsigned)w0; packed[1] = (unsigned)w1;
    scale_byte = (unsigned)eb;
}

// The same for 8 bf16 values as they come from HBM (4 packed dwords): magnitudes of bf16 compare like unsigned integers, so the block
// amax is three v_pk_max_u16 and a quad DPP reduction on the bit patterns, and v_cvt_scalef32_pk_fp8_bf16 (fp8(src / scale), round to
// nearest even; probed with tools/ubench/cvt_scale_probe.hip) unpacks, scales and converts two values per instruction: ~28 VALU
// instructions per 8 values instead of ~50 -- the quantiser, not the MFMA, is what this kernel's waves spend their issue slots on
// (PMC: 2 500 VALU instructions per wave against 24 MFMAs).  Same arithmetic as lp_quant8, so both give the same bytes.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lp_quant8_bf16(const u32x4& raw, u32x2& packed, unsigned& scale_byte) {
    const u16x2 a0 = __builtin_bit_cast(u16x2, raw[0] & 0x7fff7fffu), a1 = __builtin_bit_cast(u16x2, raw[1] & 0x7fff7fffu);
    const u16x2 a2 = __builtin_bit_cast(u16x2, raw[2] & 0x7fff7fffu), a3 = __builtin_bit_cast(u16x2, raw[3] & 0x7fff7fffu);
    const u16x2 mm = __builtin_elementwise_max(__builtin_elementwise_max(a0, a1), __builtin_elementwise_max(a2, a3));
    unsigned am = mm[0] > mm[1] ? (unsigned)mm[0] : (unsigned)mm[1];
    {
        const unsigned o1 = (unsigned)__builtin_amdgcn_mov_dpp((int)am, 0xB1, 0xF, 0xF, true);
        am = am > o1 ? am : o1;
        const unsigned o2 = (unsigned)__builtin_amdgcn_mov_dpp((int)am, 0x4E, 0xF, 0xF, true);
        am = am > o2 ? am : o2;
    }
    int eb = (int)((am >> 7) & 0xffu) - 8;                       // bf16: exponent in bits 14..7, 7 mantissa bits
    if ((am & 0x7fu) > 0x60u) eb += 1;                           // mantissa > 1.75 (same rule as lp_quant8 / pack_mxfp8)
    eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);
    const float scale = __uint_as_float((unsigned)eb << 23);    // 2^(eb - 127)
    // (inline asm: with the __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16 form hipcc 7.2 emitted both halves of a dword from the SAME
    //  source register and dropped the second dword's conversions -- seen in the ISA and as wrong products on the GPU)
    unsigned w0 = 0, w1 = 0;
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(w0) : "v"(raw[0]), "v"(scale));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w0) : "v"(raw[1]), "v"(scale));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(w1) : "v"(raw[2]), "v"(scale));
    asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(w1) : "v"(raw[3]), "v"(scale));
    packed[0] = w0; packed[1] = w1;
    scale_byte = (unsigned)eb;
}

// GELU of the reduced-precision mode's fc1 epilogue: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 the
// result is stored in) on v_rcp_f32 / v_exp_f32 -- ~16 VALU instructions per element instead of the ~28 of the < 1 ulp erf the
// parity path needs.  Deterministic, and identical in encoder and decoder (both run this kernel).
__device__ __forceinline__ float lp_gelu(float x) {
    const float z = x * 0.70710678118654752440f, az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-az * az * 1.44269504088896340736f);
    const float erf_abs = fmaf(-p, e, 1.0f);
    const float erfv = copysignf(erf_abs, z);
    return 0.5f * x * (1.0f + erfv);
}

template <bool OBF>
__device__ __forceinline__ void lp_store4(const lvae_gemm_desc& d, long off, f32x4 o) {
    if (OBF) {
        u16x4 q = {f32_to_bf16(o[0]), f32_to_bf16(o[1]), f32_to_bf16(o[2]), f32_to_bf16(o[3])};
        *(u16x4*)((unsigned short*)d.out + off) = q;
    } else {
        *(f32x4*)(d.out + off) = o;
    }
}
template <bool OBF>
__device__ __forceinline__ f32x4 lp_load4(const float* base, long off) {
    if (OBF) {
        const u32x2 q = *(const u32x2*)((const unsigned short*)base + off);
        return (f32x4){bf16_lo(q[0]), bf16_hi(q[0]), bf16_lo(q[1]), bf16_hi(q[1])};
    }
    return *(const f32x4*)(base + off);
}

template <int TN, int AMODE, bool ABF, bool OBF>
__global__ __launch_bounds__(256, (TN <= 2 && ABF ? 3 : 2)) void gemm_lp_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
    constexpr int BM = 128, BN = 64 * TN, STAGE = (BM + BN) * LP_ROWB;
    constexpr int NWCH = BN * 4, NW = (NWCH + 255) / 256;          // 16-B W chunks per stage / per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = (char*)smem;
    int t;
    {
        const int b = blockIdx.x, q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    constexpr int esz = ABF ? 2 : 4;

    // A staging: chunk c = tid + 256 j (j < 4): row c >> 3, piece c & 7 (8 k-values); a quad of lanes = one 32-element MX block
    LpRow rows[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rows[j] = lp_row<AMODE>(d, m0 + ((tid + 256 * j) >> 3), esz);
    const int piece = tid & 7;
    const unsigned char* wq = (const unsigned char*)d.Wt16;        // [N][ldw] fp8 bytes, ldw = K rounded up to 64
    const unsigned char* wsc = wq + (long)d.N * d.ldw;             // [N][ldw / 32] E8M0 bytes
    const int nsc = (int)(d.ldw >> 5);

    f32x16 acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Register staging, one k-step ahead (DEPTH = 1).  Deeper rings (2 and 3 steps in flight, branch-free so that hipcc's counted
    // s_waitcnt vmcnt survive) were measured and bought nothing: the waves are not waiting for HBM but for each other at the
    // per-step barrier while the quantiser competes for VALU issue (PMC: 2 500 VALU instructions per wave, 24 MFMAs).
    constexpr int DEPTH = 1;
    u32x4 ra[DEPTH][4][ABF ? 1 : 2];
    unsigned okm[DEPTH];
    u32x4 rw[DEPTH][NW];
    unsigned short rs[DEPTH];
    const int nk = (d.K + 63) / 64;

    // k-steps are issued for kt in [0, ceil(nk / DEPTH) * DEPTH): steps beyond the last real one re-read it (valid addresses) with
    // their A flagged as zeros, so they add exactly 0 to the accumulators -- and the whole pipeline is BRANCH-FREE: with conditional
    // loads hipcc's s_waitcnt vmcnt at the control-flow merges is the worst case over both paths, i.e. vmcnt(0) before every
    // quantiser pass (PMC: waves parked 51 % of their cycles), which threw the ring's depth away.
    auto gload = [&](auto set_tag, int kt_req) {
        constexpr int S = decltype(set_tag)::value;
        const bool real = kt_req < nk;
        const int kt = real ? kt_req : nk - 1;
        const int k = kt * 64 + piece * 8;
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok;
            const char* p = lp_addr<AMODE>(d, rows[j], k, esz, ok);
            m |= ((ok && real) ? 1u : 0u) << j;
            ra[S][j][0] = *(const u32x4*)p;
            if (!ABF) ra[S][j][1] = *(const u32x4*)(p + 16);
        }
        okm[S] = m;
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int c = tid + 256 * j;
            int n = n0 + (c >> 2);
            n = n < d.N ? n : d.N - 1;
            if (NWCH % 256 == 0 || c < NWCH) rw[S][j] = *(const u32x4*)(wq + (long)n * d.ldw + kt * 64 + (c & 3) * 16);
        }
        if (tid < BN) {
            int n = n0 + tid;
            n = n < d.N ? n : d.N - 1;
            rs[S] = *(const unsigned short*)(wsc + (long)n * nsc + kt * 2);
        }
    };
    auto lstore = [&](auto set_tag, char* st) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (okm[S] >> j) & 1u;
            u32x2 pk;
            unsigned sb;
            if (ABF) {
                u32x4 raw = ra[S][j][0];
#pragma unroll
                for (int e = 0; e < 4; ++e) raw[e] = ok ? raw[e] : 0u;
                lp_quant8_bf16(raw, pk, sb);
            } else {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(ra[S][j][0][e]); v[4 + e] = __uint_as_float(ra[S][j][ABF ? 0 : 1][e]); }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ok ? v[e] : 0.f;
                lp_quant8(v, pk, sb);
            }
            const int row = (tid + 256 * j) >> 3;
            *(u32x2*)(st + row * LP_ROWB + piece * 8) = pk;
            if ((piece & 3) == 0) *(unsigned char*)(st + row * LP_ROWB + 64 + (piece >> 2)) = (unsigned char)sb;
            __builtin_amdgcn_sched_barrier(0);           // one chunk's quantiser temporaries at a time (register pressure)
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int c = tid + 256 * j;
            if (NWCH % 256 == 0 || c < NWCH) *(u32x4*)(st + (BM + (c >> 2)) * LP_ROWB + (c & 3) * 16) = rw[S][j];
        }
        if (tid < BN) *(unsigned short*)(st + (BM + tid) * LP_ROWB + 64) = rs[S];
    };
    // one k-step: MFMAs on LDS stage kt & 1, then k-step kt + 1 (register set (kt + 1) % DEPTH, in flight since step kt + 1 - DEPTH)
    // is quantised into the other stage and its set is refilled with k-step kt + 1 + DEPTH
    auto body = [&](auto nxt_tag, int kt) {
        const char* cur = lds + (kt & 1) * STAGE;
        // Operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with 8-bit elements (found by probing; tools/debug_fp8.py): lane
        // (i = lane & 31, h = lane >> 5) holds row i, k = 16 h + [0, 16) in VGPR 0-3 and k = 32 + 16 h + [0, 16) in VGPR 4-7; the
        // E8M0 scale of the row's k-block b (k in [32 b, 32 b + 32)) is taken from the lane with h = b.
        i32x8 af[2], bfr;
        int sa[2], sb;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const char* p = cur + (wave_m * 64 + a * 32 + li) * LP_ROWB;
            const u32x4 x0 = *(const u32x4*)(p + 16 * lh), x1 = *(const u32x4*)(p + 32 + 16 * lh);
            af[a] = (i32x8){(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
            sa[a] = (int)*(const unsigned char*)(p + 64 + lh);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const char* p = cur + (BM + wave_n * TN * 32 + b * 32 + li) * LP_ROWB;
            const u32x4 x0 = *(const u32x4*)(p + 16 * lh), x1 = *(const u32x4*)(p + 32 + 16 * lh);
            bfr = (i32x8){(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
            sb = (int)*(const unsigned char*)(p + 64 + lh);
            acc[0][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[0], bfr, acc[0][b], 0, 0, 0, sa[0], 0, sb);
            acc[1][b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[1], bfr, acc[1][b], 0, 0, 0, sa[1], 0, sb);
        }
        lstore(nxt_tag, lds + ((kt & 1) ^ 1) * STAGE);
        gload(nxt_tag, kt + 1 + DEPTH);
        __syncthreads();
    };
    using T0 = std::integral_constant<int, 0>;
    gload(T0{}, 0);
    lstore(T0{}, lds);
    gload(T0{}, 1);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) body(T0{}, kt);

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    const int rr = d.r, r2 = rr * rr, epi = d.epi, store = d.store;
    const int cp = (store == LVAE_ST_ROWMAJOR) ? 1 : d.N / (r2 > 0 ? r2 : 1);
    const bool vec = (store == LVAE_ST_ROWMAJOR && !(d.N & 3) && !(d.ldo & 3) && !(d.ldres & 3)) || (store == LVAE_ST_SHUFFLE && !(cp & 3));
    if (vec) {
        const int lj = li & 3;
        float cbias[TN], cgam[TN];
        long ccol4[TN];
        bool cok4[TN];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int colb = n0 + (wave_n * TN + b) * 32, col = colb + li, cc = col < d.N ? col : 0;
            cbias[b] = d.bias ? d.bias[cc] : 0.f;
            cgam[b] = (epi == LVAE_EPI_GAMMA_RES) ? d.gamma[cc] : 1.f;
            const int c4 = colb + (li & ~3);
            cok4[b] = c4 < d.N;
            const int c4c = cok4[b] ? c4 : 0;
            if (store == LVAE_ST_ROWMAJOR) {
                ccol4[b] = c4c;
            } else {
                const int q = c4c / cp, sc = c4c - q * cp, si = q / rr, sj = q - si * rr;
                ccol4[b] = ((long)si * (d.W * rr) + sj) * cp + sc;
            }
        }
        const bool has_res = epi == LVAE_EPI_GAMMA_RES || epi == LVAE_EPI_RES;
#ifdef LVAE_EXP_NO_RES_PREFETCH
        const bool res_pf = false;
#else
        const bool res_pf = has_res && store == LVAE_ST_ROWMAJOR;
#endif
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            // residual values of this 32-row block: requested up front, back to back, from clamped (always valid) addresses -- loaded
            // inside the divergent store guards each was a memory round trip of its own (gemm_common.h has the same arrangement)
            f32x4 rvp[4][TN];
            if (res_pf) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = m0 + (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;
                    const long rbase = (long)(row < d.M ? row : 0) * d.ldres;
#pragma unroll
                    for (int b = 0; b < TN; ++b) rvp[g][b] = lp_load4<OBF>(d.res, rbase + ccol4[b]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = m0 + (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;
                const bool rok = row < d.M;
                const int rowc = rok ? row : 0;
                long obase;
                if (store == LVAE_ST_ROWMAJOR) {
                    obase = (long)rowc * d.ldo;
                } else {
                    const int w = rowc % d.W, bh = rowc / d.W, h = bh % d.H, bb = bh / d.H;
                    obase = (((long)(bb * d.H + h) * rr) * (d.W * rr) + (long)w * rr) * cp;
                }
                const long rbase = (long)rowc * d.ldres;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    float v0 = acc[a][b][4 * g + 0] + cbias[b], v1 = acc[a][b][4 * g + 1] + cbias[b];
                    float v2 = acc[a][b][4 * g + 2] + cbias[b], v3 = acc[a][b][4 * g + 3] + cbias[b];
                    if (epi == LVAE_EPI_BIAS_GELU) { v0 = lp_gelu(v0); v1 = lp_gelu(v1); v2 = lp_gelu(v2); v3 = lp_gelu(v3); }
                    else if (epi == LVAE_EPI_GAMMA_RES) { v0 *= cgam[b]; v1 *= cgam[b]; v2 *= cgam[b]; v3 *= cgam[b]; }
                    quad_transpose(v0, v1, v2, v3, lj);
                    if (rok && cok4[b]) {
                        f32x4 o = {v0, v1, v2, v3};
                        if (has_res) {
                            const f32x4 rv = res_pf ? rvp[g][b] : lp_load4<OBF>(d.res, rbase + ccol4[b]);
                            o[0] += rv[0]; o[1] += rv[1]; o[2] += rv[2]; o[3] += rv[3];
                        }
                        lp_store4<OBF>(d, obase + ccol4[b], o);
                    }
                }
            }
        }
        return;
    }
    // scalar path: final image layer (fp32 NCHW, clamp) and odd leading dimensions
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + (wave_n * TN + b) * 32 + li;
        if (col >= d.N) continue;
        const float bv = d.bias ? d.bias[col] : 0.f, gm = (epi == LVAE_EPI_GAMMA_RES) ? d.gamma[col] : 1.f;
        long ccol;
        if (store == LVAE_ST_ROWMAJOR) {
            ccol = col;
        } else if (store == LVAE_ST_SHUFFLE) {
            const int q = col / cp, sc = col - q * cp, si = q / rr, sj = q - si * rr;
            ccol = ((long)si * (d.W * rr) + sj) * cp + sc;
        } else {
            const int sc = col / r2, q = col - sc * r2, si = q / rr, sj = q - si * rr;
            ccol = ((long)sc * (d.H * rr) + si) * (d.W * rr) + sj;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wave_m * 2 + a) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
                if (row >= d.M) continue;
                long obase;
                if (store == LVAE_ST_ROWMAJOR) {
                    obase = (long)row * d.ldo;
                } else {
                    const int w = row % d.W, bh = row / d.W, h = bh % d.H, bb = bh / d.H;
                    if (store == LVAE_ST_SHUFFLE) obase = (((long)(bb * d.H + h) * rr) * (d.W * rr) + (long)w * rr) * cp;
                    else obase = ((long)bb * cp * (d.H * rr) + (long)h * rr) * (d.W * rr) + (long)w * rr;
                }
                float v = acc[a][b][r] + bv;
                if (epi == LVAE_EPI_BIAS_GELU) v = lp_gelu(v);
                else if (epi == LVAE_EPI_GAMMA_RES || epi == LVAE_EPI_RES) {
                    const long ro = (long)row * d.ldres + col;
                    const float rv = OBF ? __uint_as_float((unsigned)((const unsigned short*)d.res)[ro] << 16) : d.res[ro];
                    v = rv + gm * v;
                }
                if (store == LVAE_ST_IMAGE) {
                    if (d.status && !(fabsf(v) <= 3.4028234664e38f)) atomicOr(d.status, LVAE_STATUS_NONFINITE_IMAGE);   // the clamp would hide it
                    v = fminf(fmaxf(v, -1.0f), 1.0f) * 0.5f + 0.5f;
                }
                if (OBF && store != LVAE_ST_IMAGE) ((unsigned short*)d.out)[obase + ccol] = f32_to_bf16(v);
                else d.out[obase + ccol] = v;
            }
        }
    }
}

template <int TN, int AMODE, bool ABF, bool OBF>
int launch_lp(const lvae_gemm_desc* d, hipStream_t st) {
    constexpr int BN = 64 * TN, LDS = 2 * (128 + BN) * LP_ROWB;
    const int tiles_m = (d->M + 127) / 128, tiles_n = (d->N + BN - 1) / BN, n_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_lp_kernel<TN, AMODE, ABF, OBF>), dim3(n_tiles), dim3(256), LDS, st, *d, tiles_n, n_tiles);
    return (int)hipGetLastError();
}

template <int AMODE, bool ABF, bool OBF>
int launch_lp_tn(const lvae_gemm_desc* d, hipStream_t st) {
    // 128 x 128 tiles (TN = 2) unless N <= 64; the 128 x 192 instance needs more than 256 registers per lane with the quantiser's
    // temporaries (hipcc spills ~400 of them to scratch: measured 3x slower) and is not built.  Results do not depend on the choice.
    // fp32 A (the K = z operands of z_proj): 64-wide tiles (its wider instances would spill); N a multiple of 192: 128 x 192 tiles
    // (N = 192 is ONE column tile: A is read once instead of twice, and no half-empty 128 x 128 tile)
    int tn = (d->N <= 64 || !ABF) ? 1 : ((d->N % 192 == 0) ? 3 : 2);
#ifdef LVAE_EXPERIMENTAL_BUILD           // tile sweep hook (tools/build_exp.sh copies only)
    {
        static int force = -1;
        if (force < 0) { const char* e = getenv("LVAE_LP_TN"); force = e ? atoi(e) : 0; }
        if (force >= 1 && force <= 3 && ABF && d->N > 64) tn = force;
    }
#endif
    if (tn == 1) return launch_lp<1, AMODE, ABF, OBF>(d, st);
    if constexpr (ABF) return tn == 3 ? launch_lp<3, AMODE, ABF, OBF>(d, st) : launch_lp<2, AMODE, ABF, OBF>(d, st);
    return -22;
}

template <bool ABF, bool OBF>
int launch_lp_mode(const lvae_gemm_desc* d, hipStream_t st) {
    switch (d->a_mode) {
        case LVAE_A_PLAIN: return launch_lp_tn<LVAE_A_PLAIN, ABF, OBF>(d, st);
        case LVAE_A_PATCH2: return launch_lp_tn<LVAE_A_PATCH2, ABF, OBF>(d, st);
        case LVAE_A_CONV3: return launch_lp_tn<LVAE_A_CONV3, ABF, OBF>(d, st);
    }
    return -22;
}

}  // namespace

// prec 3 entry (called by lvae_gemm_f32 after the common argument checks)
int lvae_gemm_lp_dispatch(const lvae_gemm_desc* d, hipStream_t st) {
    if (!d->Wt16 || (d->ldw & 63) || d->ldw < d->K || (d->K & 7) || (d->K0 & 7) || d->ksplit > 1 || d->a_gelu) return -22;
    const int esz = d->a_bf16 ? 2 : 4;
    if (d->a_mode == LVAE_A_PLAIN) {
        if ((d->lda0 * esz) & 15 || d->K0 + d->K1 != d->K || (d->K1 && (!d->A1 || ((d->lda1 * esz) & 15) || (d->K1 & 7)))) return -22;
    } else if (d->a_mode == LVAE_A_PATCH2) {
        if (d->K != 4 * d->K0 || d->H <= 0 || d->W <= 0) return -22;
    } else if (d->a_mode == LVAE_A_CONV3) {
        if (d->K != 9 * d->K0 || d->H <= 0 || d->W <= 0) return -22;
    } else {
        return -22;
    }
    if (d->a_bf16) return d->out_bf16 ? launch_lp_mode<true, true>(d, st) : launch_lp_mode<true, false>(d, st);
    return d->out_bf16 ? launch_lp_mode<false, true>(d, st) : launch_lp_mode<false, false>(d, st);
}
