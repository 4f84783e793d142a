// Can plain VALU instructions issued by the SAME wave between two bf16 MFMAs hide in the MFMA's 32-cycle shadow?
// One or two waves per SIMD; each wave runs {MFMA on 4 independent accumulators, F filler v_fma_f32 after each MFMA}.
// Reports cycles per MFMA for F = 0..10.  (Companion of mfma_valu_overlap*.hip, where the VALU work sat in a DIFFERENT wave.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int F>
__global__ __launch_bounds__(512) void k(const float* in, float* out, long* cyc, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const int wave = threadIdx.x >> 6;
    float a0 = in[t & 1023], b0 = in[(t + 13) & 1023];
    bf16x8 va, vb;
    for (int e = 0; e < 8; ++e) { va[e] = (__bf16)(a0 + e); vb[e] = (__bf16)(b0 - e); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = a0 + j;
    long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#define STEP(C, A, B)                                                                                     \
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(C) : "v"(A), "v"(B));              \
        _Pragma("unroll") for (int f = 0; f < F; ++f)                                                      \
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f & 7]) : "v"(a0), "v"(b0));
        STEP(c0, va, vb) STEP(c1, vb, va) STEP(c2, va, va) STEP(c3, vb, vb)
        STEP(c0, va, vb) STEP(c1, vb, va) STEP(c2, va, va) STEP(c3, vb, vb)
    }
    long t1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int j = 0; j < 8; ++j) s += x[j];
    out[t] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int F>
void run(const float* in, float* out, long* cyc, int nt) {
    const int iters = 20000;
    hipLaunchKernelGGL(k<F>, dim3(256), dim3(nt), 0, 0, in, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<F>, dim3(256), dim3(nt), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 256.0 * (nt / 64) * iters * 8.0 * 32768.0 / (ms * 1e-3) / 1e12;
    long hc[8]; hipMemcpy(hc, cyc + 8 * 100, sizeof(hc), hipMemcpyDeviceToHost);
    long mx = 0; for (int w = 0; w < nt / 64; ++w) mx = hc[w] > mx ? hc[w] : mx;
    printf("waves/SIMD %d  fillers/MFMA %2d : first wave %.1f, slowest wave %.1f cycles per own MFMA -> %.1f cycles per SIMD-MFMA; %.0f TFLOP/s bf16 chip-wide (%.1f us, random operands)\n", nt / 256, F,
           hc[0] / (iters * 8.0), mx / (iters * 8.0), mx / (iters * 8.0) / (nt / 256), tf, ms * 1e3);
}

int main() {
    float *in, *out; long* cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    for (int nt = 256; nt <= 512; nt += 256) {
        run<0>(in, out, cyc, nt); run<1>(in, out, cyc, nt); run<2>(in, out, cyc, nt); run<3>(in, out, cyc, nt);
        run<4>(in, out, cyc, nt); run<5>(in, out, cyc, nt); run<6>(in, out, cyc, nt); run<7>(in, out, cyc, nt);
        run<8>(in, out, cyc, nt); run<10>(in, out, cyc, nt); run<12>(in, out, cyc, nt);
    }
    return 0;
}
