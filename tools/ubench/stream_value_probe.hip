// stream_value_probe.hip -- round 6: what a host <-> GPU hand-over on the decode chain costs, and whether stream memory operations
// (hipStreamWriteValue32 / hipStreamWaitValue32) can take the host's launch latency and completion polling off that chain.
//   A. GPU -> host: a short kernel, then the host learns that it has finished by (1) hipStreamSynchronize, (2) polling hipStreamQuery,
//      (3) spinning on a pinned host word that hipStreamWriteValue32 writes behind the kernel, (4) ... that the kernel itself writes
//      (system-scope store by its last thread).
//   B. host -> GPU: the next kernel starts (1) by being launched when the host is ready, (2) pre-enqueued behind a
//      hipStreamWaitValue32 on a signal-memory word that the host writes when it is ready.  Measured: host "ready" -> host sees the
//      kernel's completion word.
// Build: bash tools/ubench/build.sh stream_value_probe ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void work_kernel(float* p, int n, volatile uint32_t* host_flag, uint32_t value) {
    float v = p[threadIdx.x];
    for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
    if (host_flag && threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store((uint32_t*)host_flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* buf; CK(hipMalloc(&buf, 4096));
    CK(hipMemset(buf, 0, 4096));
    volatile uint32_t* flag;                      // pinned host memory
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
    flag[0] = 0;
    uint64_t* sig = nullptr;                      // signal memory (the documented operand of the wait)
    hipError_t es = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(es));
    const int N = 200, WORK = 2000;               // ~5 us kernel
    std::vector<double> t;
    uint32_t epoch = 0;
    // warm-up
    for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, buf, WORK, nullptr, 0u); CK(hipStreamSynchronize(st)); }
    // kernel duration by events
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, buf, WORK, nullptr, 0u);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("kernel: %.1f us each (20 back to back)\n", ms * 1e3 / 20);
    }
    // A1: launch + hipStreamSynchronize
    t.clear();
    for (int i = 0; i < N; ++i) { double t0 = now_us(); hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, buf, WORK, nullptr, 0u); CK(hipStreamSynchronize(st)); t.push_back(now_us() - t0); }
    printf("A1 launch -> hipStreamSynchronize returns        : %6.1f us (median)\n", med(t));
    // A2: launch + poll hipStreamQuery
    t.clear();
    for (int i = 0; i < N; ++i) { double t0 = now_us(); hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, buf, WORK, nullptr, 0u); while (hipStreamQuery(st) == hipErrorNotReady) {} t.push_back(now_us() - t0); }
    printf("A2 launch -> hipStreamQuery says ready           : %6.1f us\n", med(t));
    // A3: launch + hipStreamWriteValue32 to pinned host memory, host spins on the word
    t.clear();
    for (int i = 0; i < N; ++i) {
        ++epoch;
        double t0 = now_us();
        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, buf, WORK, nullptr, 0u);
        hipError_t e = hipStreamWriteValue32(st, (void*)flag, epoch, 0);
        if (e != hipSuccess) { printf("hipStreamWriteValue32 on pinned host memory: %s\n", hipGetErrorString(e)); break; }
        while (flag[0] != epoch) {}
        t.push_back(now_us() - t0);
    }
    if (!t.empty()) printf("A3 launch -> word written by hipStreamWriteValue32: %6.1f us\n", med(t));
    CK(hipStreamSynchronize(st));
    // A4: the kernel writes the word itself (system-scope store)
    t.clear();
    for (int i = 0; i < N; ++i) {
        ++epoch;
        double t0 = now_us();
        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, buf, WORK, flag, epoch);
        while (flag[0] != epoch) {}
        t.push_back(now_us() - t0);
    }
    printf("A4 launch -> word written by the kernel itself     : %6.1f us\n", med(t));
    CK(hipStreamSynchronize(st));
    // B1: host ready -> launch -> completion word (kernel writes it)
    t.clear();
    for (int i = 0; i < N; ++i) {
        ++epoch;
        CK(hipStreamSynchronize(st));
        double t0 = now_us();
        hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, buf, WORK, flag, epoch);
        while (flag[0] != epoch) {}
        t.push_back(now_us() - t0);
    }
    printf("B1 host ready -> launch -> kernel's word (idle stream): %6.1f us\n", med(t));
    // B2: pre-enqueued behind hipStreamWaitValue32 on signal memory; the host writes the signal word when ready
    if (es == hipSuccess && can) {
        t.clear();
        *sig = 0;
        bool ok = true;
        for (int i = 0; i < N && ok; ++i) {
            ++epoch;
            hipError_t e = hipStreamWaitValue32(st, (void*)sig, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { printf("hipStreamWaitValue32 on signal memory: %s\n", hipGetErrorString(e)); ok = false; break; }
            hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, buf, WORK, flag, epoch);
            // give the command processor time to reach the wait
            double tw = now_us(); while (now_us() - tw < 50.0) {}
            double t0 = now_us();
            __atomic_store_n((volatile uint32_t*)sig, epoch, __ATOMIC_RELEASE);
            while (flag[0] != epoch) {}
            t.push_back(now_us() - t0);
        }
        if (ok) printf("B2 host ready -> signal word -> pre-enqueued kernel's word: %6.1f us\n", med(t));
        CK(hipStreamSynchronize(st));
    }
    // B3: the same wait on a PINNED HOST word (outside the documented contract)
    {
        volatile uint32_t* go = flag + 8;
        go[0] = 0;
        t.clear();
        bool ok = true;
        for (int i = 0; i < N && ok; ++i) {
            ++epoch;
            hipError_t e = hipStreamWaitValue32(st, (void*)go, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { printf("hipStreamWaitValue32 on pinned host memory: %s\n", hipGetErrorString(e)); ok = false; break; }
            hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, buf, WORK, flag, epoch);
            double tw = now_us(); while (now_us() - tw < 50.0) {}
            double t0 = now_us();
            __atomic_store_n(go, epoch, __ATOMIC_RELEASE);
            while (flag[0] != epoch) { if (now_us() - t0 > 2e6) { printf("B3 timed out\n"); __atomic_store_n(go, 0xFFFFFFFFu, __ATOMIC_RELEASE); ok = false; break; } }
            t.push_back(now_us() - t0);
        }
        if (ok) printf("B3 host ready -> pinned word -> pre-enqueued kernel's word: %6.1f us\n", med(t));
        (void)hipStreamSynchronize(st);
    }
    return 0;
}
