// device_math.h -- shared device helpers for the lvae HIP kernels.
#pragma once
#include <hip/hip_runtime.h>

// erf(x) to < 1 ulp (max abs error 5.8e-8, checked against fp64 erf on a dense grid: tests/test_gpu_kernels.py::test_erf)
// with 13 FMAs + one v_exp_f32, branch-free: a minimax polynomial in x^2 for |x| <= 0.9277 and 1 - exp(p(|x|)) beyond.
// Replaces the device-library erff (about twice the VALU work) in the GELU epilogue, where it was ~15% of an fc1 launch.
// No implicit mul+add contraction in these helpers (explicit fmaf only): the same GELU is evaluated in GEMM epilogues, GEMM
// operand loaders and the pointwise kernel, and its bits must not depend on the surrounding code (torch's CPU kernel, the
// reference, does not fuse x*0.5*(1+erf) either).
__device__ __forceinline__ float lvae_erff(float a) {
#pragma clang fp contract(off)
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    r = 1.0f - __expf(r);
    r = copysignf(r, a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    q = fmaf(q, a, a);
    return t > 0.927734375f ? r : q;
}

// Two-element form on packed f32 (v_pk_fma_f32 / v_pk_mul_f32: two lanes-worth of FMAs per VALU issue).  f32 MFMA and VALU
// do not overlap on a SIMD (docs/MEASUREMENT_HISTORY.md 5), so every VALU instruction in an epilogue is serial time: the polynomial halves.
typedef float lvae_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lvae_f2 lvae_erff2(lvae_f2 a) {
#pragma clang fp contract(off)
    const lvae_f2 t = {fabsf(a[0]), fabsf(a[1])};
    const lvae_f2 s = a * a;
    lvae_f2 r = __builtin_elementwise_fma((lvae_f2)(-1.72853470e-5f), t, (lvae_f2)(3.83197126e-4f));
    const lvae_f2 u = __builtin_elementwise_fma((lvae_f2)(-3.88396438e-3f), t, (lvae_f2)(2.42546219e-2f));
    r = __builtin_elementwise_fma(r, s, u);
    r = __builtin_elementwise_fma(r, t, (lvae_f2)(-1.06777877e-1f));
    r = __builtin_elementwise_fma(r, t, (lvae_f2)(-6.34846687e-1f));
    r = __builtin_elementwise_fma(r, t, (lvae_f2)(-1.28717512e-1f));
    r = __builtin_elementwise_fma(r, t, -t);
    lvae_f2 q = __builtin_elementwise_fma((lvae_f2)(-5.96761703e-4f), s, (lvae_f2)(4.99119423e-3f));
    q = __builtin_elementwise_fma(q, s, (lvae_f2)(-2.67681349e-2f));
    q = __builtin_elementwise_fma(q, s, (lvae_f2)(1.12819925e-1f));
    q = __builtin_elementwise_fma(q, s, (lvae_f2)(-3.76125336e-1f));
    q = __builtin_elementwise_fma(q, s, (lvae_f2)(1.28379166e-1f));
    q = __builtin_elementwise_fma(q, a, a);
    lvae_f2 o;
    o[0] = t[0] > 0.927734375f ? copysignf(1.0f - __expf(r[0]), a[0]) : q[0];
    o[1] = t[1] > 0.927734375f ? copysignf(1.0f - __expf(r[1]), a[1]) : q[1];
    return o;
}
// GELU's erf, ONE branch (round 5).  The two-branch erf above (kept for the likelihood kernel, which needs differences of erf values to
// RELATIVE accuracy) evaluates BOTH polynomials -- 13 FMAs -- plus a product by log2(e), a compare and a select per element: 31 VALU
// instructions per two elements, ~17 of the ~27 slots an fc1 epilogue spends per hidden element, in phases that run at the vector pipe's
// rate with the matrix pipe idle (docs/MEASUREMENT_HISTORY.md 5d).  GELU needs erf to ABSOLUTE accuracy only (it is multiplied by x / 2), so the cancellation
// of 1 - exp(.) near 0 -- the reason for the second branch -- is harmless here:
//     erf(z) = sign(z) (1 - 2^(t q(t))),  t = min(|z|, 4),  q of degree 7        (1 - erf(4) = 1.5e-8: rounds to 1)
// (weighted minimax fit of log2(erfc(t)) / t on [0, 4]: tools/fit_gelu_erf.py), and  gelu(x) = fma(x / 2, erf, x / 2)  -- one rounding
// less than (x / 2) (1 + erf).  18 VALU instructions per two elements (7 FMAs + 1 product + v_exp_f32 with log2(e) folded into q).
// fp32-evaluated accuracy: erf max |error| 9.8e-8, GELU 8.9e-8 max(1, |x|) -- the two-branch form's GELU: 1.1e-7 (its last two
// roundings) -- tests/test_gpu_kernels.py::test_gelu_erf_accuracy (bar 2e-7).  Seven candidate forms (degree 7 / 8 / 9, two fits,
// with / without the final fma) were run against every reference golden: six keep all of them flip-free, this one is the cheapest
// and the most accurate of them (profiles/r05_gelu_single_branch_study.txt).
// Scalar and packed forms apply the same IEEE operations per element (explicit fma, no contraction): identical bits.
#define LVAE_GQ0 (-1.6279085874557495f)
#define LVAE_GQ1 (-0.9184163808822632f)
#define LVAE_GQ2 (-0.14848165214061737f)
#define LVAE_GQ3 (0.028253760188817978f)
#define LVAE_GQ4 (-0.0007747217314317822f)
#define LVAE_GQ5 (-0.001489384681917727f)
#define LVAE_GQ6 (0.00044549000449478626f)
#define LVAE_GQ7 (-4.535645348369144e-05f)
__device__ __forceinline__ float lvae_gelu_erf1(float z) {
#pragma clang fp contract(off)
    const float t = fminf(fabsf(z), 4.0f);
    float r = fmaf(LVAE_GQ7, t, LVAE_GQ6);
    r = fmaf(r, t, LVAE_GQ5);
    r = fmaf(r, t, LVAE_GQ4);
    r = fmaf(r, t, LVAE_GQ3);
    r = fmaf(r, t, LVAE_GQ2);
    r = fmaf(r, t, LVAE_GQ1);
    r = fmaf(r, t, LVAE_GQ0);
    return copysignf(1.0f - __builtin_amdgcn_exp2f(r * t), z);
}
__device__ __forceinline__ void gelu_erf2(float& x0, float& x1) {
#pragma clang fp contract(off)
    const lvae_f2 x = {x0, x1};
    const lvae_f2 z = x * (lvae_f2)(0.70710678118654752440f);
    const lvae_f2 t = {fminf(fabsf(z[0]), 4.0f), fminf(fabsf(z[1]), 4.0f)};
    lvae_f2 r = __builtin_elementwise_fma((lvae_f2)(LVAE_GQ7), t, (lvae_f2)(LVAE_GQ6));
    r = __builtin_elementwise_fma(r, t, (lvae_f2)(LVAE_GQ5));
    r = __builtin_elementwise_fma(r, t, (lvae_f2)(LVAE_GQ4));
    r = __builtin_elementwise_fma(r, t, (lvae_f2)(LVAE_GQ3));
    r = __builtin_elementwise_fma(r, t, (lvae_f2)(LVAE_GQ2));
    r = __builtin_elementwise_fma(r, t, (lvae_f2)(LVAE_GQ1));
    r = __builtin_elementwise_fma(r, t, (lvae_f2)(LVAE_GQ0));
    const lvae_f2 p = r * t;
    lvae_f2 e;
    e[0] = copysignf(1.0f - __builtin_amdgcn_exp2f(p[0]), z[0]);
    e[1] = copysignf(1.0f - __builtin_amdgcn_exp2f(p[1]), z[1]);
    const lvae_f2 h = (lvae_f2)(0.5f) * x;
    const lvae_f2 g = __builtin_elementwise_fma(h, e, h);
    x0 = g[0]; x1 = g[1];
}

// Four elements: the two packed chains of gelu_erf2 advance in lockstep (each v_pk_fma_f32 depends on the one before it: a lone chain
// issues every other slot; two interleaved chains fill each other's gaps).  Per element the operations are gelu_erf2's: same bits.
__device__ __forceinline__ void gelu_erf4(float& x0, float& x1, float& x2, float& x3) {
#pragma clang fp contract(off)
    const lvae_f2 xa = {x0, x1}, xb = {x2, x3};
    const lvae_f2 za = xa * (lvae_f2)(0.70710678118654752440f), zb = xb * (lvae_f2)(0.70710678118654752440f);
    const lvae_f2 ta = {fminf(fabsf(za[0]), 4.0f), fminf(fabsf(za[1]), 4.0f)}, tb = {fminf(fabsf(zb[0]), 4.0f), fminf(fabsf(zb[1]), 4.0f)};
    lvae_f2 ra = __builtin_elementwise_fma((lvae_f2)(LVAE_GQ7), ta, (lvae_f2)(LVAE_GQ6));
    lvae_f2 rb = __builtin_elementwise_fma((lvae_f2)(LVAE_GQ7), tb, (lvae_f2)(LVAE_GQ6));
#define LVAE_G4_STEP(q) ra = __builtin_elementwise_fma(ra, ta, (lvae_f2)(q)); rb = __builtin_elementwise_fma(rb, tb, (lvae_f2)(q));
    LVAE_G4_STEP(LVAE_GQ5) LVAE_G4_STEP(LVAE_GQ4) LVAE_G4_STEP(LVAE_GQ3) LVAE_G4_STEP(LVAE_GQ2) LVAE_G4_STEP(LVAE_GQ1) LVAE_G4_STEP(LVAE_GQ0)
#undef LVAE_G4_STEP
    const lvae_f2 pa = ra * ta, pb = rb * tb;
    lvae_f2 ea, eb;
    ea[0] = copysignf(1.0f - __builtin_amdgcn_exp2f(pa[0]), za[0]);
    ea[1] = copysignf(1.0f - __builtin_amdgcn_exp2f(pa[1]), za[1]);
    eb[0] = copysignf(1.0f - __builtin_amdgcn_exp2f(pb[0]), zb[0]);
    eb[1] = copysignf(1.0f - __builtin_amdgcn_exp2f(pb[1]), zb[1]);
    const lvae_f2 ha = (lvae_f2)(0.5f) * xa, hb = (lvae_f2)(0.5f) * xb;
    const lvae_f2 ga = __builtin_elementwise_fma(ha, ea, ha), gb = __builtin_elementwise_fma(hb, eb, hb);
    x0 = ga[0]; x1 = ga[1]; x2 = gb[0]; x3 = gb[1];
}

// exact-erf GELU (nn.GELU() default; lvae/models/common.py:124,132)
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
    const float h = 0.5f * x;
    return fmaf(h, lvae_gelu_erf1(x * 0.70710678118654752440f), h);
}
