// gemm_common.h -- pieces shared by the GEMM translation units (gemm_f32.hip, gemm_x3v2.hip): vector types, the DPP quad
// transpose, the tile-configuration struct and the fused epilogue.  Everything lives in an anonymous namespace.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/lvae_hip.h"
#include "device_math.h"

// Experiment hooks of the round-1/2 kernel studies (docs/MEASUREMENT_HISTORY.md 5, 5b).  LVAE_EXP_NOSPLIT / LVAE_EXP_NOSTORE / LVAE_GEMM_NOLOAD give WRONG
// RESULTS by construction (they remove work to time what is left); the others change scheduling only.  None of them can be switched on
// in the product build: they compile only together with -DLVAE_EXPERIMENTAL_BUILD, which tools/build_exp.sh passes for its
// side-by-side copies under _bin/ and lossy-vae_amd/build_native.py never does.
#if !defined(LVAE_EXPERIMENTAL_BUILD) && (defined(LVAE_EXP_NOSPLIT) || defined(LVAE_EXP_NOSTORE) || defined(LVAE_EXP_PRIO) || defined(LVAE_EXP_H2_FULLLINE) || \
    defined(LVAE_EXP_NO_RES_PREFETCH) || defined(LVAE_EPI_PRIO) || defined(LVAE_GEMM_NOLOAD) || defined(LVAE_GEMM_TRACE) || defined(LVAE_X3V2_TRACE) || \
    defined(LVAE_EXP_PP_NOWAIT) || defined(LVAE_EXP_PP_NODMA) || defined(LVAE_EXP_PP_NODSR) || defined(LVAE_EXP_PP_NOBAR) || defined(LVAE_EXP_H2PP))
#error "LVAE_EXP_* / *_TRACE / *_NOLOAD experiment hooks need -DLVAE_EXPERIMENTAL_BUILD (tools/build_exp.sh); never in liblvae_hip.so"
#endif

namespace {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: a process that drives several GPUs (or a second device later on)
// must set it on each of them, and the two pipeline-group threads may get here at the same time.  `done` is one bit per device ordinal
// (a launcher-local static): set-once per device, lock-free, and setting it twice in a race is harmless.
struct LdsAttr {
    std::atomic<unsigned long long> done{0};
    int ensure(const void* fn, int bytes, const void* fn2 = nullptr, int bytes2 = 0, const void* fn3 = nullptr, int bytes3 = 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return 0;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && fn2) e = hipFuncSetAttribute(fn2, hipFuncAttributeMaxDynamicSharedMemorySize, bytes2);
        if (e == hipSuccess && fn3) e = hipFuncSetAttribute(fn3, hipFuncAttributeMaxDynamicSharedMemorySize, bytes3);
        if (e != hipSuccess) return (int)e;
        done.fetch_or(bit, std::memory_order_release);
        return 0;
    }
};

// Compute units of the CURRENT device, cached per device ordinal (a process may drive several GPUs, and the pipeline-group threads may
// ask at the same time: relaxed atomics, every writer stores the same value).  Persistent kernels size their grid by it -- one
// workgroup per CU that owns the whole LDS -- so a count taken from another device would break their residency assumption.
inline int lvae_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    std::atomic<int>& c = cached[dev & 63];
    int n = c.load(std::memory_order_relaxed);
    if (n > 0) return n;
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
    c.store(v, std::memory_order_relaxed);
    return v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));



// 4x4 transpose across the 4 lanes of a quad with DPP quad_perm moves (lane^1: [1,0,3,2] = 0xB1, lane^2: [2,3,0,1] =
// 0x4E): afterwards lane j holds in (v0..v3) what lanes 0..3 of its quad held in register j.  Used to turn the MFMA
// accumulator layout (4 consecutive ROWS per lane) into 4 consecutive COLUMNS per lane => 16-B stores / residual loads.
__device__ __forceinline__ float dpp_xor1(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));
}
// Every call site passes j = lane & 3 (the lane's place in its quad), so the two select masks are the constants 0xAAAA... (odd lanes)
// and 0xCCCC... (upper lane pair).  Form: each exchange step is ONE v_cndmask_b32_dpp per register -- dst = vcc ? own : dpp(partner's
// register) -- instead of select + v_mov_dpp + two selects (with two wait states in front of every DPP read of a fresh select): 8 VALU
// + 4 s_mov per 4 x 4 transpose instead of 16 VALU + 8 idle slots, in epilogues whose time is their VALU issue.  VOP2-DPP reads its
// mask from VCC only, hence the s_mov pairs; the leading s_nop covers "VALU wrote the source, DPP reads it" (two wait states counting the
// s_mov) for sources produced right in front of the call; inside, every DPP source is at least three instructions old.
#ifndef LVAE_QUAD_TRANSPOSE_SELECT_FORM
__device__ __forceinline__ void quad_transpose(float& v0, float& v1, float& v2, float& v3, int /* j == lane & 3 */) {
    float n0, n1, n2, n3, o0, o1, o2, o3;
    asm("s_mov_b64 vcc, %[m1]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_dpp %[n1], %[v0], %[v1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"      // odd lane keeps v1, even takes partner's v0
        "v_cndmask_b32_dpp %[n3], %[v2], %[v3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %[m1n]\n\t"
        "v_cndmask_b32_dpp %[n0], %[v1], %[v0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"      // even lane keeps v0, odd takes partner's v1
        "v_cndmask_b32_dpp %[n2], %[v3], %[v2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %[m2]\n\t"
        "v_cndmask_b32_dpp %[o3], %[n1], %[n3], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"      // upper pair keeps n3, lower takes partner's n1
        "v_cndmask_b32_dpp %[o2], %[n0], %[n2], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %[m2n]\n\t"
        "v_cndmask_b32_dpp %[o1], %[n3], %[n1], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"      // lower pair keeps n1, upper takes partner's n3
        "v_cndmask_b32_dpp %[o0], %[n2], %[n0], vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3)
        : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [m1] "s"(0xAAAAAAAAAAAAAAAAull), [m1n] "s"(0x5555555555555555ull),
          [m2] "s"(0xCCCCCCCCCCCCCCCCull), [m2n] "s"(0x3333333333333333ull)
        : "vcc");
    v0 = o0; v1 = o1; v2 = o2; v3 = o3;
}
#else
__device__ __forceinline__ void quad_transpose(float& v0, float& v1, float& v2, float& v3, int j) {
    const bool o1 = (j & 1) != 0, o2 = (j & 2) != 0;
    float t;
    t = dpp_xor1(o1 ? v0 : v1); if (o1) v0 = t; else v1 = t;
    t = dpp_xor1(o1 ? v2 : v3); if (o1) v2 = t; else v3 = t;
    t = dpp_xor2(o2 ? v0 : v2); if (o2) v0 = t; else v2 = t;
    t = dpp_xor2(o2 ? v1 : v3); if (o2) v1 = t; else v3 = t;
}
#endif

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// f16x2 arithmetic (prec 4): (x0, x1) -> packed fp16 pairs hi, lo' with hi + lo' / 2048 == x to 2^-24 relative -- the conversions
// lvae.models.base.split_f16x2 applies to the weights.  hi = f16(x) (RNE); lo' = f16((x - hi) * 2048): fma(hi, -2048, x * 2048) is
// exact before its single rounding to fp16 (x * 2048 and hi * 2048 are exact, their difference has <= 13 significant bits).
__device__ __forceinline__ void split_pair_h2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2_t x = {x0, x1};
    const f16x2_t h = __builtin_convertvector(x, f16x2_t);               // v_cvt_pk_f16_f32 (RNE)
    const f32x2_t t = x * 2048.0f;
    f16x2_t l;
    l[0] = (_Float16)__builtin_fmaf((float)h[0], -2048.0f, t[0]);
    l[1] = (_Float16)__builtin_fmaf((float)h[1], -2048.0f, t[1]);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

template <int WGM_, int WGN_, int TM_, int TN_, int NBUF_ = 2, int BK_ = 32>
struct Cfg {
    static constexpr int BK = BK_;                // k-tile depth (32, or 64 for the small latency-bound problems)
    static constexpr int LDT = BK + 4;            // padded LDS row (floats): rows*LDT mod 64 distinct multiples of 4
    static constexpr int CPR = BK / 4;            // 16-B chunks per tile row
    static constexpr int WGM = WGM_, WGN = WGN_, TM = TM_, TN = TN_;
    static constexpr int NBUF = NBUF_;            // LDS stages: 2 = double-buffered (1 barrier / k-tile), 1 = single (2 barriers)
    static constexpr int NT = 64 * WGM * WGN;     // threads per workgroup (4 or 8 wave64)
    static constexpr int BM = WGM * TM * 32;
    static constexpr int BN = WGN * TN * 32;
    static constexpr int RP = NT / CPR;           // tile rows staged per pass (CPR lanes x 16 B = one row segment)
    static constexpr int NA = (BM + RP - 1) / RP; // float4 loads per thread per k-tile (A)
    static constexpr int NB = (BN + RP - 1) / RP; // (W)
#ifdef LVAE_GEMM_TRACE
    static constexpr int LDS_BYTES = NBUF * (BM + BN) * LDT * 4 + 1024;
#else
    static constexpr int LDS_BYTES = NBUF * (BM + BN) * LDT * 4;
#endif
    static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves");
    static_assert(BM % RP == 0, "A tile must be a whole number of staging passes");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// SLAB = true: split-K partial sums going to the workspace with WRITE-THROUGH (sc1) 16-B buffer stores, so that the in-kernel
// reduction needs no agent-scope release (a buffer_wbl2 writes back every dirty line of the XCD's L2, also those of kernels
// running beside this one on another stream: measured -14 % end to end): cdna_hip_programming.md Guideline 16, form R1.
template <class C, bool SLAB = false>
__device__ __forceinline__ void gemm_epilogue(const lvae_gemm_desc& d, f32x16 (&acc)[C::TM][C::TN], int m0, int n0, int wave_m,
                                              int wave_n, int li, int lh) {
    // C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // acc[][] is only ever indexed statically (a runtime index sends the whole accumulator tile to scratch).
#ifdef LVAE_EPI_PRIO
    __builtin_amdgcn_s_setprio(LVAE_EPI_PRIO);      // experiment: favour the epilogue's VALU/VMEM issue over a co-resident MFMA wave
#endif
    const int rr = d.r, r2 = rr * rr;
    const int epi = d.epi, store = d.store;
    const int cp = (store == LVAE_ST_ROWMAJOR) ? 1 : d.N / r2;
    const bool vec = (store == LVAE_ST_ROWMAJOR && !(d.N & 3) && !(d.ldo & 3) && !(d.ldres & 3)) ||
                     (store == LVAE_ST_SHUFFLE && !(cp & 3));
    if (vec) {
        // vector path: per-column ops on the lane's own column, quad transpose, then one 16-B access per 4 outputs
        const int lj = li & 3;
        float cbias[C::TN], cgam[C::TN];
        long ccol4[C::TN];
        bool cok4[C::TN];
#pragma unroll
        for (int b = 0; b < C::TN; ++b) {
            const int colb = n0 + (wave_n * C::TN + b) * 32;
            const int col = colb + li;
            const int cc = col < d.N ? col : 0;
            cbias[b] = d.bias ? d.bias[cc] : 0.f;
            cgam[b] = (epi == LVAE_EPI_GAMMA_RES) ? d.gamma[cc] : 1.f;
            const int c4 = colb + (li & ~3);                 // first of the 4 columns this lane stores
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
        for (int a = 0; a < C::TM; ++a) {
            // The residual values of this 32-row block are requested up front, all 4 * TN of them back to back (rows and columns are
            // clamped to valid addresses, so the loads are unconditional): loaded next to their use, inside the divergent store
            // guards, every one of them was a memory round trip of its own -- 15 % of an fc2 launch.
            f32x4 rv[4][C::TN];
            if (res_pf) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = m0 + (wave_m * C::TM + a) * 32 + 4 * lh + 8 * g + lj;
                    const float* resrow = d.res + (long)(row < d.M ? row : 0) * d.ldres;
#pragma unroll
                    for (int b = 0; b < C::TN; ++b) rv[g][b] = *(const f32x4*)(resrow + ccol4[b]);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = m0 + (wave_m * C::TM + a) * 32 + 4 * lh + 8 * g + lj;     // row this lane stores
                const bool rok = row < d.M;
                const int rowc = rok ? row : 0;
                long obase;
                if (store == LVAE_ST_ROWMAJOR) {
                    obase = (long)rowc * d.ldo;
                } else {
                    const int w = rowc % d.W, bh = rowc / d.W, h = bh % d.H, bb = bh / d.H;
                    obase = (((long)(bb * d.H + h) * rr) * (d.W * rr) + (long)w * rr) * cp;
                }
                const float* resrow = d.res + (long)rowc * d.ldres;
#pragma unroll
                for (int b = 0; b < C::TN; ++b) {
                    float v0 = acc[a][b][4 * g + 0] + cbias[b], v1 = acc[a][b][4 * g + 1] + cbias[b];
                    float v2 = acc[a][b][4 * g + 2] + cbias[b], v3 = acc[a][b][4 * g + 3] + cbias[b];
                    if (epi == LVAE_EPI_BIAS_GELU) { gelu_erf2(v0, v1); gelu_erf2(v2, v3); }
                    else if (epi == LVAE_EPI_GAMMA_RES) { v0 *= cgam[b]; v1 *= cgam[b]; v2 *= cgam[b]; v3 *= cgam[b]; }
                    quad_transpose(v0, v1, v2, v3, lj);
                    if (rok && cok4[b]) {
                        f32x4 o = {v0, v1, v2, v3};
                        if (has_res) {
                            const f32x4 r4 = res_pf ? rv[g][b] : *(const f32x4*)(resrow + ccol4[b]);
                            o[0] += r4[0]; o[1] += r4[1]; o[2] += r4[2]; o[3] += r4[3];
                        }
                        if (SLAB) {
                            const long bytes = (long)d.M * d.N * 4;
                            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                                (void*)d.out, 0, bytes > 0x7fffffffL ? 0x7fffffff : (int)bytes, 0x00020000);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rs, (int)((obase + ccol4[b]) * 4), 0, 16);
                        } else if (d.out_h2) {
                            // pre-split result for a consumer GEMM with a_h2 (H2K32: per row and 32 columns, 32 hi terms then 32 lo')
                            unsigned h0, l0, h1, l1;
                            split_pair_h2(o[0], o[1], h0, l0);
                            split_pair_h2(o[2], o[3], h1, l1);
                            const u32x2_t hi2 = {h0, h1}, lo2 = {l0, l1};
                            char* q = (char*)d.out + (obase << 2) + ((ccol4[b] >> 5) << 7) + ((ccol4[b] & 31) << 1);
                            *(u32x2_t*)q = hi2;
                            *(u32x2_t*)(q + 64) = lo2;
                        } else {
                            *(f32x4*)(d.out + obase + ccol4[b]) = o;
                        }
                    }
                }
            }
        }
        return;
    }
    // scalar path (final image layer, odd leading dimensions): rows-outer / columns-inner, 4-B accesses
    float cbias[C::TN], cgam[C::TN];
    long ccol[C::TN];                 // ROWMAJOR: col; SHUFFLE/IMAGE: column part of the output offset
    bool cok[C::TN];
#pragma unroll
    for (int b = 0; b < C::TN; ++b) {
        const int col = n0 + (wave_n * C::TN + b) * 32 + li;
        cok[b] = col < d.N;
        const int cc = cok[b] ? col : 0;
        cbias[b] = d.bias ? d.bias[cc] : 0.f;
        cgam[b] = (epi == LVAE_EPI_GAMMA_RES) ? d.gamma[cc] : 1.f;
        if (store == LVAE_ST_ROWMAJOR) {
            ccol[b] = cc;
        } else if (store == LVAE_ST_SHUFFLE) {        // column n' = (i*r+j)*cp + c
            const int q = cc / cp, sc = cc - q * cp, si = q / rr, sj = q - si * rr;
            ccol[b] = ((long)si * (d.W * rr) + sj) * cp + sc;
        } else {                                      // IMAGE: column n = c*r^2 + i*r + j -> NCHW
            const int sc = cc / r2, q = cc - sc * r2, si = q / rr, sj = q - si * rr;
            ccol[b] = ((long)sc * (d.H * rr) + si) * (d.W * rr) + sj;
        }
    }
    bool bad = false;
#pragma unroll
    for (int a = 0; a < C::TM; ++a) {
        const int rbase = m0 + (wave_m * C::TM + a) * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            if (row >= d.M) continue;
            long obase;
            if (store == LVAE_ST_ROWMAJOR) {
                obase = (long)row * d.ldo;
            } else {
                const int w = row % d.W, bh = row / d.W, h = bh % d.H, bb = bh / d.H;
                if (store == LVAE_ST_SHUFFLE) obase = (((long)(bb * d.H + h) * rr) * (d.W * rr) + (long)w * rr) * cp;
                else obase = ((long)bb * cp * (d.H * rr) + (long)h * rr) * (d.W * rr) + (long)w * rr;
            }
            const float* resrow = d.res + (long)row * d.ldres;
#pragma unroll
            for (int b = 0; b < C::TN; ++b) {
                if (!cok[b]) continue;
                float v = acc[a][b][r] + cbias[b];
                if (epi == LVAE_EPI_BIAS_GELU) v = gelu_erf(v);
                else if (epi == LVAE_EPI_GAMMA_RES) v = resrow[ccol[b]] + cgam[b] * v;
                else if (epi == LVAE_EPI_RES) v = resrow[ccol[b]] + v;
                if (store == LVAE_ST_IMAGE) {
                    bad |= !(fabsf(v) <= 3.4028234664e38f);      // NaN / inf: the clamp below would hide it (include/lvae_hip.h "status word")
                    v = fminf(fmaxf(v, -1.0f), 1.0f) * 0.5f + 0.5f;
                }
                d.out[obase + ccol[b]] = v;
            }
        }
    }
    if (store == LVAE_ST_IMAGE && d.status && bad) atomicOr(d.status, LVAE_STATUS_NONFINITE_IMAGE);
}

#ifndef H2P_FULL_LINE
#define H2P_FULL_LINE true
#endif
// Straight-line epilogue for the launches gemm_h2p_kernel exists for (fc1: + bias -> GELU -> pre-split store; fc2: + bias, * gamma,
// + residual -> fp32 store; both row-major on a full-width tile).  gemm_epilogue decides epi / store / out_h2 / residual at RUN time
// inside its unit loop: every 4-element unit is a chain of uniform branches, its store sits behind an exec-mask branch, and the
// compiler can neither interleave the units' dependent FMA chains nor keep more than one store in flight.  Here the case is a
// template parameter, rows beyond M are dropped by the buffer resource's range check (no branch around any access) and the whole
// wave tile (64 rows x 32 TN columns: gemm_h2p_kernel's and gemm_h2_kernel's) unrolls.  The operations per element and their order are gemm_epilogue's (acc + bias; GELU | * gamma; quad transpose;
// + residual; split_pair_h2), hence the same bits (tests/test_gpu_f16x2.py::test_gemm_h2p_equals_h2_bit_for_bit).
// FULL (pre-split output only): quad pairs exchange halves by DPP row shifts so that one lane holds the hi terms of EIGHT consecutive
// columns (its partner the lo' terms): one 16-B store per lane and 8 rows x 128 B = whole lines per instruction instead of two 8-B
// stores covering half lines.
// FOLDED: the tail of the serial split-K form -- `acc` holds the running sum of the slices, and the operations are splitk_epilogue_store's
// (quad transpose first, then + bias as a 4-column vector, GELU | fma(gamma, v, residual) | + residual): that function loads bias,
// gamma and residual inside every unit, each load followed by a full wait -- eight units of two dependent memory round trips at the
// end of launches whose whole K loop is ~10 us (the stride-16 / 32 / 64 MLPs of both paths' dependency chains).
template <int TN, int EPI, bool H2, bool FULL, bool FOLDED = false>
__device__ __forceinline__ void h2p_epilogue_fast(const lvae_gemm_desc& d, f32x16 (&acc)[2][TN], int m0, int n0, int rows_a, int wave_m,
                                                  int wave_n, int li, int lh) {
#pragma clang fp contract(off)
    const int lj = li & 3;
    float cbias[TN], cgam[TN];
    f32x4 vbias[TN], vgam[TN];
    int c4[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int colb = n0 + (wave_n * TN + b) * 32;
        c4[b] = colb + (li & ~3);
        if constexpr (FOLDED) {
            vbias[b] = *(const f32x4*)(d.bias + c4[b]);
            if constexpr (EPI == LVAE_EPI_GAMMA_RES) vgam[b] = *(const f32x4*)(d.gamma + c4[b]);
        } else {
            cbias[b] = d.bias ? d.bias[colb + li] : 0.f;
            cgam[b] = EPI == LVAE_EPI_GAMMA_RES ? d.gamma[colb + li] : 1.f;
        }
    }
    constexpr bool HAS_RES = EPI == LVAE_EPI_GAMMA_RES || EPI == LVAE_EPI_RES;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)d.out + (long)m0 * d.ldo * 4), 0, rows_a * d.ldo * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR =
        __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_RES ? (const char*)d.res + (long)m0 * d.ldres * 4 : (const char*)d.out), 0, HAS_RES ? rows_a * d.ldres * 4 : 0, 0x00020000);
    const bool odd_quad = (li & 4) != 0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        u32x4_t rv[4][TN];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;
#pragma unroll
                for (int b = 0; b < TN; ++b) rv[g][b] = __builtin_amdgcn_raw_buffer_load_b128(rsR, (r * d.ldres + c4[b]) * 4, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = (wave_m * 2 + a) * 32 + 4 * lh + 8 * g + lj;          // tile-local row this lane stores
            const int rowoff = r * d.ldo * 4;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                float v0, v1, v2, v3;
                if constexpr (FOLDED) {
                    v0 = acc[a][b][4 * g + 0]; v1 = acc[a][b][4 * g + 1]; v2 = acc[a][b][4 * g + 2]; v3 = acc[a][b][4 * g + 3];
                    quad_transpose(v0, v1, v2, v3, lj);
                    v0 += vbias[b][0]; v1 += vbias[b][1]; v2 += vbias[b][2]; v3 += vbias[b][3];
                    if constexpr (EPI == LVAE_EPI_BIAS_GELU) gelu_erf4(v0, v1, v2, v3);
                    if constexpr (HAS_RES) {
                        const f32x4 r4 = __builtin_bit_cast(f32x4, rv[g][b]);
                        if constexpr (EPI == LVAE_EPI_GAMMA_RES) {        // (splitk_epilogue_store's "r + g * v", which hipcc contracts: one rounding)
                            v0 = __builtin_fmaf(vgam[b][0], v0, r4[0]); v1 = __builtin_fmaf(vgam[b][1], v1, r4[1]);
                            v2 = __builtin_fmaf(vgam[b][2], v2, r4[2]); v3 = __builtin_fmaf(vgam[b][3], v3, r4[3]);
                        } else { v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3]; }
                    }
                } else {
                    const lvae_f2 s01 = (lvae_f2){acc[a][b][4 * g + 0], acc[a][b][4 * g + 1]} + (lvae_f2)(cbias[b]);      // (packed adds)
                    const lvae_f2 s23 = (lvae_f2){acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]} + (lvae_f2)(cbias[b]);
                    v0 = s01[0]; v1 = s01[1]; v2 = s23[0]; v3 = s23[1];
                    if constexpr (EPI == LVAE_EPI_BIAS_GELU) gelu_erf4(v0, v1, v2, v3);
                    else if constexpr (EPI == LVAE_EPI_GAMMA_RES) { v0 *= cgam[b]; v1 *= cgam[b]; v2 *= cgam[b]; v3 *= cgam[b]; }
                    quad_transpose(v0, v1, v2, v3, lj);
                    if constexpr (HAS_RES) {
                        const f32x4 r4 = __builtin_bit_cast(f32x4, rv[g][b]);
                        v0 += r4[0]; v1 += r4[1]; v2 += r4[2]; v3 += r4[3];
                    }
                }
                if constexpr (H2) {
                    unsigned h0, l0, h1, l1;
                    split_pair_h2(v0, v1, h0, l0);
                    split_pair_h2(v2, v3, h1, l1);
                    if constexpr (FULL) {
                        // even quad: {own hi, partner's hi} = hi of columns c8 .. c8 + 7; odd quad: {partner's lo', own lo'}
                        const unsigned o2 = __builtin_amdgcn_update_dpp(l0, h0, 0x104, 0xF, 0x5, false);     // row_shl:4 into banks 0, 2
                        const unsigned o3 = __builtin_amdgcn_update_dpp(l1, h1, 0x104, 0xF, 0x5, false);
                        const unsigned o0 = __builtin_amdgcn_update_dpp(h0, l0, 0x114, 0xF, 0xA, false);     // row_shr:4 into banks 1, 3
                        const unsigned o1 = __builtin_amdgcn_update_dpp(h1, l1, 0x114, 0xF, 0xA, false);
                        const int c8 = c4[b] & ~7;
                        const int off = rowoff + ((c8 >> 5) << 7) + ((c8 & 31) << 1) + (odd_quad ? 64 : 0);
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){o0, o1, o2, o3}, rsO, off, 0, 0);
                    } else {
                        const int off = rowoff + ((c4[b] >> 5) << 7) + ((c4[b] & 31) << 1);
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){h0, h1}, rsO, off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){l0, l1}, rsO, off + 64, 0, 0);
                    }
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (f32x4){v0, v1, v2, v3}), rsO, rowoff + c4[b] * 4, 0, 0);
                }
            }
        }
    }
}

// Split-K tail: out = epilogue(sum over slices IN SLICE ORDER of ws[s] + bias) for 16-B chunks of the output.  Shared by the in-kernel
// reduction (the tile's last-arriving slice workgroup) and by the stand-alone reduce kernel (d.cnt == NULL): same operations in the
// same order, hence the same bits.
template <bool H2OUT = false>
__device__ __forceinline__ void splitk_epilogue_store(const lvae_gemm_desc& d, long m, int c, f32x4 v) {
    if (d.bias) { const f32x4 b = *(const f32x4*)(d.bias + c); v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3]; }
    if (d.epi == LVAE_EPI_BIAS_GELU) {
        float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        gelu_erf2(a0, a1); gelu_erf2(a2, a3);
        v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3;
    } else if (d.epi == LVAE_EPI_GAMMA_RES) {
        const f32x4 g = *(const f32x4*)(d.gamma + c), r = *(const f32x4*)(d.res + m * d.ldres + c);
        v[0] = r[0] + g[0] * v[0]; v[1] = r[1] + g[1] * v[1]; v[2] = r[2] + g[2] * v[2]; v[3] = r[3] + g[3] * v[3];
    } else if (d.epi == LVAE_EPI_RES) {
        const f32x4 r = *(const f32x4*)(d.res + m * d.ldres + c);
        v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
    }
    if (H2OUT && d.out_h2) {  // pre-split result (H2K32) for a consumer GEMM with a_h2: serial split-K of gemm_h2p.hip only (ldo == N, N % 32 == 0)
        unsigned h0, l0, h1, l1;
        split_pair_h2(v[0], v[1], h0, l0);
        split_pair_h2(v[2], v[3], h1, l1);
        const u32x2_t hi2 = {h0, h1}, lo2 = {l0, l1};
        char* q = (char*)d.out + ((m * d.ldo) << 2) + ((c >> 5) << 7) + ((c & 31) << 1);
        *(u32x2_t*)q = hi2;
        *(u32x2_t*)(q + 64) = lo2;
        return;
    }
    *(f32x4*)(d.out + m * d.ldo + c) = v;
}
__device__ __forceinline__ void splitk_reduce_chunk(const lvae_gemm_desc& d, int S, long m, int c) {
    const long plane = (long)d.M * d.N;
    const float* w = d.ws + m * d.N + c;
    f32x4 v = *(const f32x4*)w;
    for (int s = 1; s < S; ++s) {
        const f32x4 p = *(const f32x4*)(w + s * plane);
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
    }
    splitk_epilogue_store(d, m, c, v);
}
// U chunks per thread at once (chunk e0 + u * stride of the rows x c4n chunk grid of the tile at (m0, n0)): the U loads of a slab are in
// flight together.  One workgroup reduces a whole tile here, alone, from memory: with one chunk at a time it spent S dependent memory
// round trips per 4 KB (a 128 x 192 tile with S = 2: ~50 us; the stand-alone reduce kernel spreads the same bytes over the chip).
template <int U>
__device__ __forceinline__ void splitk_reduce_chunks(const lvae_gemm_desc& d, int S, int m0, int n0, int c4n, int e0, int stride, int total) {
    const long plane = (long)d.M * d.N;
    const float* w[U];
    f32x4 v[U];
    long m[U];
    int c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = e0 + u * stride, ee = e < total ? e : e0;
        const int r = ee / c4n;
        m[u] = m0 + r; c[u] = n0 + 4 * (ee - r * c4n);
        w[u] = d.ws + m[u] * d.N + c[u];
        v[u] = *(const f32x4*)w[u];
    }
    for (int s = 1; s < S; ++s) {
        f32x4 p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = *(const f32x4*)(w[u] + s * plane);
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u][0] += p[u][0]; v[u][1] += p[u][1]; v[u][2] += p[u][2]; v[u][3] += p[u][3]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (e0 + u * stride < total) splitk_epilogue_store(d, m[u], c[u], v[u]);
}

// Every GEMM kernel ends here.  gridDim.y = number of K slices: with split-K the raw partial sums of slice blockIdx.y go to its
// workspace plane (row-major, no bias / activation / residual).  Then either a second kernel finishes the job (d.cnt == NULL:
// splitk_reduce_kernel, gemm_f32.hip) or -- the default -- the LAST slice workgroup of the tile to arrive does it in place
// (d.cnt = one zeroed arrival counter per tile): write-through (sc1) slab stores -> every wave drains its stores -> workgroup
// barrier -> one lane: ticket = relaxed agent-scope atomic add on the tile's counter; the workgroup that draws S-1 acquires (agent scope), re-reads
// ALL S slabs -- its own included -- in slice order and applies the epilogue, then zeroes the counter for the next launch.
// The sum order is fixed (slice 0, 1, ..., S-1) whoever arrives last, so the result is deterministic and equal to the two-kernel
// form; nothing depends on dispatch order or workgroup -> XCD placement (cdna_hip_programming.md Guideline 16, counter form).
// `lds` is the kernel's dynamic LDS (free after the main loop: the broadcast word lives there, not in a second __shared__ object).
template <class C>
__device__ __forceinline__ void gemm_finish(const lvae_gemm_desc& d, f32x16 (&acc)[C::TM][C::TN], int m0, int n0, int wave_m,
                                            int wave_n, int li, int lh, void* lds, int tile) {
    lvae_gemm_desc p = d;
    const int S = (int)gridDim.y;
    if (S > 1) {
        p.out = d.ws + (long)blockIdx.y * d.M * d.N;
        p.ldo = d.N; p.bias = nullptr; p.epi = LVAE_EPI_BIAS; p.store = LVAE_ST_ROWMAJOR; p.res = nullptr; p.ldres = 0;
    }
    if (!(S > 1 && d.cnt)) {
        gemm_epilogue<C>(p, acc, m0, n0, wave_m, wave_n, li, lh);
        return;
    }
    {
        gemm_epilogue<C, true>(p, acc, m0, n0, wave_m, wave_n, li, lh);     // write-through slab stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its slab stores
        __syncthreads();
        volatile int* last = (volatile int*)lds;
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(d.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int is_last = old == S - 1;
            if (is_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(d.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            }
            *last = is_last;
        }
        __syncthreads();
        if (*last) {
            // every wave of the reducing workgroup acquires (drops its view of lines it may hold from before the ticket): the slab reads
            // below are plain loads, and the ticket holder's fence covers its own wave only
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int rows = (d.M - m0) < C::BM ? (d.M - m0) : C::BM;
            const int cols = (d.N - n0) < C::BN ? (d.N - n0) : C::BN;       // N % 4 == 0 (checked on the host)
            const int c4n = cols >> 2;
            for (int e0 = threadIdx.x; e0 < rows * c4n; e0 += C::NT * 8) splitk_reduce_chunks<8>(d, S, m0, n0, c4n, e0, C::NT, rows * c4n);
        }
    }
}

}  // namespace
