// Do f32 MFMA (v_mfma_f32_32x32x2_f32) and plain f32 VALU work from a CO-RESIDENT wave on the same SIMD overlap?
// 512-thread workgroups (2 waves per SIMD): waves 0-3 run an MFMA-only loop, waves 4-7 run a VALU-only FMA loop (mode 1),
// idle (mode 0) or MFMA too (mode 2).  Reports the MFMA waves' elapsed shader cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(const float* in, float* out, long* cyc, int iters, int mode, int prio) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const int wave = threadIdx.x >> 6;
    float a0 = in[t & 1023], b0 = in[(t + 13) & 1023];
    bf16x8 va, vb;
    for (int e = 0; e < 8; ++e) { va[e] = (__bf16)(a0 + e); vb[e] = (__bf16)(b0 - e); }
    long t0 = clock64();
    float s = 0;
    if ((wave < 4 && mode != 3) || mode == 2) {
        if (prio == 2) __builtin_amdgcn_s_setprio(3);
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, va, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, va, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, vb, c3, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    } else if ((mode == 1 || mode == 3) && wave >= 4) {
        if (prio == 1) __builtin_amdgcn_s_setprio(3);
        float x0 = a0, x1 = b0, x2 = a0 + 1, x3 = b0 + 1, x4 = a0 - 1, x5 = b0 - 1, x6 = a0 * 2, x7 = b0 * 2;
        for (int i = 0; i < iters * 8; ++i) {        // a fixed amount of VALU work, shorter than the MFMA waves' loop
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
                x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f);
            }
        }
        s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    long t1 = clock64();
    out[t] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float *in, *out; long* cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)(i % 17) * 0.01f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int prio = 0; prio < 3; ++prio)
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, cyc, 10, mode, prio);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, in, out, cyc, iters, mode, prio);
        hipDeviceSynchronize();
        long hc[16]; hipMemcpy(hc, cyc + 8 * 100, sizeof(hc), hipMemcpyDeviceToHost);
        double ideal = (double)iters * 32 * 32;
        printf("prio %d mode %d (%s): MFMA wave cycles %ld = %.2fx the MFMA-only ideal (%.0f); partner wave cycles %ld\n", prio, mode,
               mode == 0 ? "partner idle" : mode == 1 ? "partner VALU fma loop" : mode == 2 ? "partner MFMA too" : "VALU wave alone (MFMA waves idle)", hc[0], hc[0] / ideal, ideal, hc[4]);
    }
    return 0;
}
