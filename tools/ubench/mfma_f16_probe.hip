// v_mfma_f32_32x32x16_f16 probe for the fp16 hi/lo operand split (round 3):
//   1. are fp16 SUBNORMAL inputs honoured or flushed?
//   2. how are the 16 products of one instruction summed (exactly, or with fp32 roundings in between)?
//   3. throughput + effective clock of an f16 MFMA loop vs the bf16 one (random operands), 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// A[i][k] = av[k] for every row i, B[k][j] = bv[k] for every column j  =>  every C element = sum_k av[k] * bv[k]
__global__ void probe(const float* av, const float* bv, float* out, float c0) {
    const int lane = threadIdx.x, h = lane >> 5;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)av[8 * h + e]; b[e] = (_Float16)bv[8 * h + e]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = c0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
}

template <bool F16>
__global__ __launch_bounds__(256) void rate(const float* in, float* out, long* cyc, int iters) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    f16x8 ha[2], hb[2];
    bf16x8 ba[2], bb[2];
    for (int e = 0; e < 8; ++e)
        for (int i = 0; i < 2; ++i) {
            const float x = in[(t * 8 + e + 131 * i) & 1023], y = in[(t * 8 + e + 577 * i + 64) & 1023];
            ha[i][e] = (_Float16)x; hb[i][e] = (_Float16)y; ba[i][e] = (__bf16)x; bb[i][e] = (__bf16)y;
        }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (F16) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[0], hb[0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[0], hb[1], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[1], hb[0], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[1], hb[1], c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba[0], bb[0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba[0], bb[1], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba[1], bb[0], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba[1], bb[1], c3, 0, 0, 0);
            }
        }
    }
    long t1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[t] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float run_probe(float* dav, float* dbv, float* dout, const float* av, const float* bv, float c0) {
    hipMemcpy(dav, av, 64, hipMemcpyHostToDevice); hipMemcpy(dbv, bv, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dav, dbv, dout, c0);
    float r; hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
    return r;
}

int main() {
    float *dav, *dbv, *dout;
    hipMalloc(&dav, 64); hipMalloc(&dbv, 64); hipMalloc(&dout, 64);
    float av[16], bv[16];
    // 1. subnormal inputs: a = 2^-20 (fp16 subnormal, 16 ulps), b = 1
    for (int k = 0; k < 16; ++k) { av[k] = ldexpf(1.f, -20); bv[k] = 1.f; }
    printf("subnormal A (2^-20) x 1, 16 terms: got %.9e  expected %.9e  (0 => inputs flushed)\n", run_probe(dav, dbv, dout, av, bv, 0.f), 16 * ldexp(1.0, -20));
    for (int k = 0; k < 16; ++k) { av[k] = ldexpf(1.f, -20); bv[k] = ldexpf(1.f, -20); }
    printf("subnormal x subnormal (2^-40 each), 16 terms: got %.9e  expected %.9e\n", run_probe(dav, dbv, dout, av, bv, 0.f), 16 * ldexp(1.0, -40));
    // 2. summation: one product of 1, fifteen of 2^-24 (each alone is a tie that rounds to even = lost in a sequential fp32 sum)
    for (int k = 0; k < 16; ++k) { av[k] = k == 0 ? 1.f : ldexpf(1.f, -12); bv[k] = k == 0 ? 1.f : ldexpf(1.f, -12); }
    printf("1 + 15 * 2^-24 (c0 = 0): got 1 + %.4f * 2^-23   (exact sum would round to 1 + 7.5 -> 8 * 2^-23; sequential fp32 gives 0)\n",
           (run_probe(dav, dbv, dout, av, bv, 0.f) - 1.0) * ldexp(1.0, 23));
    for (int k = 0; k < 16; ++k) { av[k] = ldexpf(1.f, -12); bv[k] = ldexpf(1.f, -12); }
    printf("c0 = 1, 16 products of 2^-24: got 1 + %.4f * 2^-23   (exact: 8)\n", (run_probe(dav, dbv, dout, av, bv, 1.f) - 1.0) * ldexp(1.0, 23));
    for (int k = 0; k < 16; ++k) { av[k] = k == 15 ? 1.f : ldexpf(1.f, -12); bv[k] = k == 15 ? 1.f : ldexpf(1.f, -12); }
    printf("15 * 2^-24 then 1 (last k): got 1 + %.4f * 2^-23\n", (run_probe(dav, dbv, dout, av, bv, 0.f) - 1.0) * ldexp(1.0, 23));
    // products need 22 bits: (1 + 2^-10)^2 = 1 + 2^-9 + 2^-20 exactly
    for (int k = 0; k < 16; ++k) { av[k] = k == 0 ? 1.f + ldexpf(1.f, -10) : 0.f; bv[k] = av[k]; }
    printf("(1 + 2^-10)^2: got 1 + 2^-9 + %.4f * 2^-20   (1 => product exact)\n", (run_probe(dav, dbv, dout, av, bv, 0.f) - 1.0 - ldexp(1.0, -9)) * ldexp(1.0, 20));

    // 3. rates
    float* in; float* out; long* cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX * 2 - 1;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    for (int f16 = 0; f16 < 2; ++f16)
        for (int blocks = 256; blocks <= 512; blocks *= 2) {
            const int iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (f16) hipLaunchKernelGGL(rate<true>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, rep ? iters : 100);
                else hipLaunchKernelGGL(rate<false>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, rep ? iters : 100);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long hc[512]; hipMemcpy(hc, cyc, blocks * 8, hipMemcpyDeviceToHost);
            double flops = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 16;
            printf("%s MFMA 32x32x16, random operands, %d blocks (%d wave/SIMD): %.1f TF/s, %.2f ms, s_memtime ticks/ms = %.0f\n",
                   f16 ? "f16 " : "bf16", blocks, blocks / 256, flops / ms / 1e9, ms, hc[0] / ms);
        }
    return 0;
}
