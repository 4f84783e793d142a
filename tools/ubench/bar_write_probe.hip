// bar_write_probe.hip -- round 6: can the HOST write a decode block's symbols straight into DEVICE memory (over the PCIe BAR) while it decodes,
// so that the dequantize launch reads local HBM instead of pulling int32 symbols from pinned host memory at the head of every GPU segment
// (28 us per launch at 4 images, 2.4 MB for a stride-16 block: profiles/r06_kernel_sequence_*_b8)?  Measures, for plain hipMalloc memory and for
// fine-grained device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)):
//   * whether the CPU may store to the pointer at all (a fault ends the probe: run it under `timeout`, one kind per process: argv[1] = 0 / 1),
//   * sequential 4-byte CPU stores of 2.4 MB: GB/s,
//   * a kernel summing the words afterwards (visibility + device-side read time), against the same kernel reading pinned host memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(256) void sum_kernel(const int* p, int n, unsigned long long* out) {
    __shared__ unsigned long long part[256];
    unsigned long long s = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += (unsigned)p[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(out, part[0]);          // (one atomic per workgroup: `out` is pinned host memory)
}

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    const int n = 4 * 147456;                                     // one stride-16 block of four images
    int* dev = nullptr;
    if (kind == 0) CK(hipMalloc((void**)&dev, n * 4));
    else if (kind == 1) CK(hipExtMallocWithFlags((void**)&dev, n * 4, hipDeviceMallocFinegrained));
    else CK(hipExtMallocWithFlags((void**)&dev, n * 4, hipDeviceMallocUncached));
    int* pinned; CK(hipHostMalloc((void**)&pinned, n * 4, hipHostMallocDefault));
    unsigned long long* out; CK(hipHostMalloc((void**)&out, 8, hipHostMallocDefault));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("kind %d (%s): pointer %p\n", kind, kind == 0 ? "hipMalloc" : (kind == 1 ? "fine-grained device memory" : "uncached device memory"), (void*)dev);
    fflush(stdout);
    // device-side read of pinned host memory vs device memory (filled by hipMemcpy)
    for (int i = 0; i < n; ++i) pinned[i] = i & 1023;
    CK(hipMemcpy(dev, pinned, n * 4, hipMemcpyHostToDevice));
    for (int src = 0; src < 2; ++src) {
        double best = 1e9;
        for (int r = 0; r < 10; ++r) {
            *out = 0;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(sum_kernel, dim3(96), dim3(256), 0, st, src ? dev : pinned, n, out);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms * 1e3 < best) best = ms * 1e3;
        }
        printf("kernel reading %d words from %s: %.1f us\n", n, src ? "device memory" : "pinned host memory", best);
    }
    fflush(stdout);
    // CPU stores straight to the device pointer
    printf("CPU stores to the device pointer ...\n"); fflush(stdout);
    volatile int* vd = dev;
    vd[0] = 7;                                                     // faults here if the CPU has no access
    printf("  first store ok\n"); fflush(stdout);
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now_us();
        for (int i = 0; i < n; ++i) dev[i] = (i * 7 + rep) & 1023;
        __builtin_ia32_sfence();
        const double t1 = now_us();
        *out = 0;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(sum_kernel, dim3(96), dim3(256), 0, st, dev, n, out);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float kms; CK(hipEventElapsedTime(&kms, e0, e1));
        printf("  (kernel reading the freshly stored words: %.1f us)\n", kms * 1e3);
        unsigned long long want = 0;
        for (int i = 0; i < n; ++i) want += (unsigned)((i * 7 + rep) & 1023);
        printf("  %d sequential 4-byte stores: %.1f us = %.2f GB/s; kernel sees them: %s\n", n, t1 - t0, n * 4 / (t1 - t0) / 1e3, *out == want ? "yes" : "NO");
    }
    // the same stores to pinned host memory, for scale
    {
        const double t0 = now_us();
        for (int i = 0; i < n; ++i) pinned[i] = (i * 7) & 1023;
        const double t1 = now_us();
        printf("  the same stores to pinned host memory: %.1f us = %.2f GB/s\n", t1 - t0, n * 4 / (t1 - t0) / 1e3);
    }
    return 0;
}
