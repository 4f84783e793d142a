// Semantics probe of v_cvt_scalef32_pk_fp8_bf16 / v_cvt_scalef32_pk_fp8_f32 (gfx950): which way does `scale` act, rounding, saturation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o, const float* a, const float* b, const float* sc, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    bf16x2 v = {(__bf16)a[i], (__bf16)b[i]};
    s16x2 w = {0, 0};
    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(w, v, sc[i], false);
    s16x2 w2 = {0, 0};
    w2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w2, a[i], b[i], sc[i], false);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a[i], b[i], 0, false);
    o[3 * i] = (unsigned)(unsigned short)w[0];
    o[3 * i + 1] = (unsigned)(unsigned short)w2[0];
    o[3 * i + 2] = (unsigned)p & 0xffff;
}
static float fp8_e4m3(unsigned char x) {
    int s = x >> 7, e = (x >> 3) & 15, m = x & 7;
    float v = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    if (e == 15 && m == 7) v = NAN;
    return s ? -v : v;
}
int main() {
    const int n = 10;
    float ha[n] = {1.f, 3.f, 448.f, 500.f, 0.3f, 17.f, 1.0625f, 1.1875f, -2.5f, 1000.f};
    float hb[n] = {2.f, 5.f, 1.f, 1.f, 0.7f, 19.f, 1.0625f, 1.1875f, 2.5f, 1e-3f};
    float hs[n] = {1.f, 2.f, 1.f, 1.f, 0.25f, 4.f, 1.f, 1.f, 0.5f, 8.f};
    float *a, *b, *s; unsigned* o;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&s, n * 4); hipMalloc(&o, n * 12);
    hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice); hipMemcpy(s, hs, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, a, b, s, n);
    unsigned ho[3 * n];
    hipMemcpy(ho, o, n * 12, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("a=%g b=%g scale=%g | scalef32_bf16: (%g, %g)  scalef32_f32: (%g, %g)  plain cvt_pk: (%g, %g)\n", ha[i], hb[i], hs[i],
               fp8_e4m3(ho[3 * i] & 255), fp8_e4m3(ho[3 * i] >> 8), fp8_e4m3(ho[3 * i + 1] & 255), fp8_e4m3(ho[3 * i + 1] >> 8),
               fp8_e4m3(ho[3 * i + 2] & 255), fp8_e4m3(ho[3 * i + 2] >> 8));
    return 0;
}
