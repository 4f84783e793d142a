// wg_dispatch.hip -- round 6: how fast the chip STARTS workgroups.  The pre-split MLP GEMMs run 128 x 64 tiles, one workgroup of four
// waves per tile: a stride-8 layer of a 4-image group is 2 304 workgroups that live ~2 us each (profiles/r06_pmc_gemm_h2p_wave_states.txt:
// 467 waves resident on average in a 37 us launch with 3 072 wave slots).  If workgroups cannot be started faster than they finish, the
// launch time is the dispatch time whatever the kernel does inside.  This measures the time of a launch of G workgroups that do (almost)
// nothing, by threads per workgroup, LDS allocation and register allocation, and the time when each workgroup additionally stays for a
// fixed number of cycles (s_sleep): dispatch cost = what does not shrink with the work.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VGPRS>
__global__ __launch_bounds__(256) void k_empty(float* out, int stay) {
    extern __shared__ float smem[];
    if (VGPRS > 128) asm volatile("v_mov_b32 v163, 0" ::: "v163");      // makes the kernel allocate 164 registers per lane
    else if (VGPRS > 64) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    for (int i = 0; i < stay; ++i) __builtin_amdgcn_s_sleep(8);            // ~512 cycles each
    if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = smem[0];
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    float* out; CK(hipMalloc(&out, 4));
    CK(hipFuncSetAttribute((const void*)k_empty<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_empty<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_empty<164>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("%7s %8s %7s %6s %6s | %10s %14s\n", "WGs", "threads", "LDS KB", "VGPRs", "stay", "launch us", "ns per WG");
    for (int G : {256, 768, 2304, 9216}) {
        for (int T : {256}) {
            for (int lds : {0, 48, 72}) {
                for (int v : {32, 164}) {
                    for (int stay : {0, 4, 16}) {                   // 0, ~2k, ~8k cycles of residence
                        if (T == 512 && (lds == 0 || v == 32)) continue;
                        float t;
                        if (v == 32) t = timeit([&] { hipLaunchKernelGGL(k_empty<32>, dim3(G), dim3(T), lds * 1024, 0, out, stay); }, 20);
                        else t = timeit([&] { hipLaunchKernelGGL(k_empty<164>, dim3(G), dim3(T), lds * 1024, 0, out, stay); }, 20);
                        printf("%7d %8d %7d %6d %6d | %10.1f %14.1f\n", G, T, lds, v, stay, t * 1e3, t * 1e6 / G);
                    }
                }
            }
        }
    }
    return 0;
}
