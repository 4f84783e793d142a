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
// do not overlap on a SIMD, so every VALU instruction in an epilogue is serial time: the polynomial halves.
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
__device__ __forceinline__ void gelu_erf2(float& x0, float& x1) {
#pragma clang fp contract(off)
    const lvae_f2 x = {x0, x1};
    const lvae_f2 e = lvae_erff2(x * (lvae_f2)(0.70710678118654752440f));
    const lvae_f2 g = ((lvae_f2)(0.5f) * x) * ((lvae_f2)(1.0f) + e);
    x0 = g[0]; x1 = g[1];
}

// exact-erf GELU (nn.GELU() default; lvae/models/common.py:124,132)
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
    return 0.5f * x * (1.0f + lvae_erff(x * 0.70710678118654752440f));
}
