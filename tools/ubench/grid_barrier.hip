// grid_barrier.hip -- round 6: what a GRID-WIDE barrier inside a persistent kernel costs on MI355X, against the kernel boundary it would replace.
// VERDICT r05 item 2 asks for one persistent kernel per latent-block segment on the small maps (phases = today's launches, handing their
// activations over through L2 behind grid barriers).  A launch boundary costs ~3 us here (profiles/r06_wg_dispatch.txt: empty kernels back to
// back 2.7-3.4 us); this measures the alternative:
//   bar   : G workgroups (one per CU, or two), R rounds of { atomic add (release, agent scope) ; spin on the counter (acquire, agent scope) }
//   bar+d : the same with a hand-over in every round -- each workgroup writes 4 KB, and behind the barrier reads the 4 KB that the workgroup
//           (b + G / 2) % G wrote (another XCD: workgroups are dealt to the XCDs round robin) and folds it into what it writes next
//   bar2 / bar2+d: the two-level barrier below (per-XCD counters and flags)
//   launch: the same hand-over as R back-to-back launches of a one-round kernel (stream order instead of the barrier)
// Output: us per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// two-level form: arrive on the counter of the workgroup's XCD (b % 8: its own 128-B line); the last arriver of an XCD arrives on the global
// counter; the last of those publishes the round in eight per-XCD flag lines, and a workgroup spins on its XCD's flag only
__device__ __forceinline__ void grid_barrier2(unsigned* st, int b, int G, unsigned round1) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int x = b & 7, per = (G >> 3) + ((G & 7) > x ? 1 : 0);
        unsigned* cx = st + 32 * x;            // [0 .. 255]: per-XCD counters, one line each
        unsigned* cg = st + 32 * 8;            // global counter
        unsigned* fl = st + 32 * (9 + x);      // per-XCD flags
        const unsigned o = __hip_atomic_fetch_add(cx, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (o == per * round1 - 1) {
            const int nx = G < 8 ? G : 8;
            const unsigned og = __hip_atomic_fetch_add(cg, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (og == nx * round1 - 1)
                for (int i = 0; i < nx; ++i) __hip_atomic_store(st + 32 * (9 + i), round1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round1) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

template <bool DATA>
__global__ __launch_bounds__(256) void k_persistent2(unsigned* st, float* buf, int rounds, float* sink) {
    const int G = gridDim.x, b = blockIdx.x;
    float4 v = {1.f, 2.f, 3.f, (float)b};
    for (int r = 0; r < rounds; ++r) {
        if (DATA) ((float4*)(buf + ((size_t)(r & 1) * G + b) * 1024))[threadIdx.x] = v;
        grid_barrier2(st, b, G, (unsigned)(r + 1));
        if (DATA) {
            const float4 o = ((const float4*)(buf + ((size_t)(r & 1) * G + (b + G / 2) % G) * 1024))[threadIdx.x];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
    }
    if (v.x == 123.456f) *sink = v.y;
}

template <bool DATA>
__global__ __launch_bounds__(256) void k_persistent(unsigned* cnt, float* buf, int rounds, float* sink) {
    const int G = gridDim.x, b = blockIdx.x;
    float4 v = {1.f, 2.f, 3.f, (float)b};
    for (int r = 0; r < rounds; ++r) {
        if (DATA) ((float4*)(buf + ((size_t)(r & 1) * G + b) * 1024))[threadIdx.x] = v;
        grid_barrier(cnt, (unsigned)G * (r + 1));
        if (DATA) {
            const float4 o = ((const float4*)(buf + ((size_t)(r & 1) * G + (b + G / 2) % G) * 1024))[threadIdx.x];
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
    }
    if (v.x == 123.456f) *sink = v.y;
}

__global__ __launch_bounds__(256) void k_round(float* buf, int r, float* sink) {
    const int G = gridDim.x, b = blockIdx.x;
    float4 v = {1.f, 2.f, 3.f, (float)b};
    if (r > 0) {
        const float4 o = ((const float4*)(buf + ((size_t)((r - 1) & 1) * G + (b + G / 2) % G) * 1024))[threadIdx.x];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    ((float4*)(buf + ((size_t)(r & 1) * G + b) * 1024))[threadIdx.x] = v;
    if (v.x == 123.456f) *sink = v.y;
}

int main() {
    unsigned* cnt; float *buf, *sink;
    CK(hipMalloc(&cnt, 4096)); CK(hipMalloc(&buf, 2 * 512 * 4096)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 200;
    printf("%-8s %5s | %12s\n", "kind", "WGs", "us per round");
    for (int G : {8, 48, 96, 256, 512}) {
        for (int kind = 0; kind < 5; ++kind) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(cnt, 0, 4096));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k_persistent<false>, dim3(G), dim3(256), 0, 0, cnt, buf, R, sink);
                else if (kind == 1) hipLaunchKernelGGL(k_persistent<true>, dim3(G), dim3(256), 0, 0, cnt, buf, R, sink);
                else if (kind == 3) hipLaunchKernelGGL(k_persistent2<false>, dim3(G), dim3(256), 0, 0, cnt, buf, R, sink);
                else if (kind == 4) hipLaunchKernelGGL(k_persistent2<true>, dim3(G), dim3(256), 0, 0, cnt, buf, R, sink);
                else for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k_round, dim3(G), dim3(256), 0, 0, buf, r, sink);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%-8s %5d | %12.2f\n", kind == 0 ? "bar" : (kind == 1 ? "bar+d" : (kind == 2 ? "launch" : (kind == 3 ? "bar2" : "bar2+d"))), G, best * 1e3f / R);
        }
    }
    return 0;
}
