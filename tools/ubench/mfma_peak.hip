// MFMA-only micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain on this chip with random vs zero operands,
// 1 or 2 waves per SIMD?  Reports TFLOP/s and the effective shader clock (s_memtime ticks / wall time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(const float* in, float* out, long* cyc, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    float a0 = in[t & 1023], a1 = in[(t + 7) & 1023], b0 = in[(t + 13) & 1023], b1 = in[(t + 29) & 1023];
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
        }
    }
    long t1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[t] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *in, *out; long* cyc;
    const int blocks_max = 512;
    hipMalloc(&in, 4096); hipMalloc(&out, blocks_max * 256 * 4); hipMalloc(&cyc, blocks_max * 8);
    float h[1024];
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 1024; ++i) h[i] = mode ? (float)rand() / RAND_MAX * 2 - 1 : 0.f;
        hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
        for (int blocks = 256; blocks <= 512; blocks *= 2) {
            const int iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long hc[512]; hipMemcpy(hc, cyc, blocks * 8, hipMemcpyDeviceToHost);
            double flops = (double)blocks * 4 * iters * 64 * 4096.0;
            printf("%s operands, %d blocks (%d wave/SIMD): %.1f TF/s, %.2f ms, s_memtime ticks/ms = %.0f (ticks %ld)\n",
                   mode ? "random" : "zero", blocks, blocks / 256, flops / ms / 1e9, ms, hc[0] / ms, hc[0]);
        }
    }
    return 0;
}
