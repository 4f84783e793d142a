// mem_bw.hip -- what the chip sustains for plain streaming stores, loads and both at once (16 B per lane, whole 128-B lines per 8 lanes,
// grid-stride over a buffer far larger than the 256 MB Infinity Cache), and for LDS-DMA loads from a workgroup whose other wave is
// storing: the numbers behind "a GEMM's result stores cost their bandwidth, not their latency" (csrc/gemm_h2q.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_store(f32x4* out, long n4) {
    const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = v;
}
__global__ void k_load(const f32x4* in, long n4, float* sink) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) { const f32x4 v = in[i]; a += v; }
    if (a[0] + a[1] + a[2] + a[3] == 123.456f) *sink = a[0];
}
__global__ void k_copy(const f32x4* in, f32x4* out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
// half of the waves of a workgroup load, the other half store (different buffers): do a CU's loads wait behind its stores?
__global__ void k_split(const f32x4* in, f32x4* out, long n4, float* sink) {
    const int half = blockDim.x / 2;
    const bool loader = threadIdx.x < half;
    const long t = (long)blockIdx.x * half + (threadIdx.x % half), stride = (long)gridDim.x * half;
    if (loader) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (long i = t; i < n4; i += stride) { const f32x4 v = in[i]; a += v; }
        if (a[0] + a[1] + a[2] + a[3] == 123.456f) *sink = a[0];
    } else {
        const f32x4 v = {1.f, 2.f, 3.f, 4.f};
        for (long i = t; i < n4; i += stride) out[i] = v;
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
    for (long mb : {151L, 604L, 2416L}) {
        const long bytes = mb << 20, n4 = bytes / 16;
        f32x4 *x, *y; float* sink;
        CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&sink, 4));
        CK(hipMemset(x, 0, bytes)); CK(hipMemset(y, 0, bytes));
        const int grid = 256 * 8, bs = 256;
        const float ts = timeit([&] { hipLaunchKernelGGL(k_store, dim3(grid), dim3(bs), 0, 0, y, n4); }, 10);
        const float tl = timeit([&] { hipLaunchKernelGGL(k_load, dim3(grid), dim3(bs), 0, 0, x, n4, sink); }, 10);
        const float tc = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(bs), 0, 0, x, y, n4); }, 10);
        const float tp = timeit([&] { hipLaunchKernelGGL(k_split, dim3(grid), dim3(bs), 0, 0, x, y, n4, sink); }, 10);
        printf("%5ld MB: store %7.1f us = %5.2f TB/s | load %7.1f us = %5.2f TB/s | copy (load + store) %7.1f us = %5.2f TB/s of traffic | "
               "half the waves load, half store, %ld MB each: %7.1f us (load alone %.1f, store alone %.1f)\n",
               mb, ts * 1e3, bytes / ts / 1e9, tl * 1e3, bytes / tl / 1e9, tc * 1e3, 2.0 * bytes / tc / 1e9, mb, tp * 1e3, tl * 1e3, ts * 1e3);
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(sink));
    }
    return 0;
}
