// Cross-wave overlap on one SIMD, second look: waves 0-3 of a 512-thread workgroup run an f16 MFMA loop (v_mfma_f32_32x32x16_f16), waves 4-7 (their
// SIMD partners) a VALU loop.  mfma_valu_overlap.hip found the partner starved while the MFMA wave had INDEPENDENT MFMAs ready every cycle.
// Question: does the partner get issue slots when the MFMA wave's next MFMA is NOT ready -- a dependent chain on NACC accumulators (1 = fully
// dependent), or an s_nop between MFMAs, or under s_setprio?  Reports cycles of both waves against their solo times.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int GAP>       // GAP: 0 none, 1 = s_nop 7 behind every MFMA, 2 = s_setprio 0 on the MFMA wave / 3 on the VALU wave
__global__ __launch_bounds__(512) void k(const float* in, float* out, long* cyc, int iters, int mode) {   // mode 0 both, 1 MFMA waves only, 2 VALU waves only
    const int t = threadIdx.x + blockIdx.x * blockDim.x, wave = threadIdx.x >> 6;
    float a0 = in[t & 1023], b0 = in[(t + 13) & 1023], s = 0;
    long t0 = clock64();
    if (wave < 4 && mode != 2) {
        if (GAP == 2) __builtin_amdgcn_s_setprio(0);
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(a0 + e); b[e] = (_Float16)(b0 - e); }
        f32x16 c[4] = {{0}, {0}, {0}, {0}};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[j % NACC]) : "v"(a), "v"(b));
                if (GAP == 1) asm volatile("s_nop 7");
            }
        }
        for (int r = 0; r < 16; ++r) s += c[0][r] + c[1][r] + c[2][r] + c[3][r];
    } else if (wave >= 4 && mode != 1) {
        if (GAP == 2) __builtin_amdgcn_s_setprio(3);
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = a0 + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(a0), "v"(b0));     // 64 VALU per 16 MFMAs of the partner
        }
        for (int j = 0; j < 8; ++j) s += x[j];
    }
    long t1 = clock64();
    out[t] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int NACC, int GAP>
void run(const float* in, float* out, long* cyc) {
    const int iters = 4000;
    long r[3][2];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL((k<NACC, GAP>), dim3(256), dim3(512), 0, 0, in, out, cyc, 10, mode);
        hipLaunchKernelGGL((k<NACC, GAP>), dim3(256), dim3(512), 0, 0, in, out, cyc, iters, mode);
        hipDeviceSynchronize();
        long hc[8]; hipMemcpy(hc, cyc + 8 * 100, sizeof(hc), hipMemcpyDeviceToHost);
        r[mode][0] = hc[0]; r[mode][1] = hc[4];
    }
    printf("NACC %d gap %d: MFMA wave alone %6.1f cycles per MFMA, VALU wave alone %5.2f cycles per VALU | together: MFMA wave %6.1f per MFMA, VALU wave %5.2f per VALU  -> sum of solo times %.2f M, together %.2f M cycles\n",
           NACC, GAP, r[1][0] / (iters * 16.0), r[2][1] / (iters * 64.0), r[0][0] / (iters * 16.0), r[0][1] / (iters * 64.0),
           (r[1][0] + r[2][1]) / 1e6, (r[0][0] > r[0][1] ? r[0][0] : r[0][1]) / 1e6);
}
int main() {
    float *in, *out; long* cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)(i % 17) * 0.01f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    run<4, 0>(in, out, cyc); run<2, 0>(in, out, cyc); run<1, 0>(in, out, cyc);
    run<4, 1>(in, out, cyc); run<1, 1>(in, out, cyc); run<4, 2>(in, out, cyc); run<1, 2>(in, out, cyc);
    return 0;
}
