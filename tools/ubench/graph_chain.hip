// graph_chain.hip -- round 6: per-kernel cost of a dependent chain of small kernels, launched one by one on a stream against the same chain
// replayed as ONE hipGraph (captured once).  The decode segments are chains of 10-20 kernels of 5-20 us; rounds 1-2 found no gain from graphs
// at ~640 launches of the Python-driven plans -- this re-measures the GPU-side cost per kernel with today's short native launch loop.
//   chain of N kernels, each `wgs` workgroups x 256 threads doing `work` dependent FMAs and reading what its predecessor wrote.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void step(const float* in, float* out, int work) {
    float v = in[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *a, *b; CK(hipMalloc(&a, 512 * 256 * 4)); CK(hipMalloc(&b, 512 * 256 * 4));
    CK(hipMemset(a, 0, 512 * 256 * 4)); CK(hipMemset(b, 0, 512 * 256 * 4));
    printf("%5s %5s %6s | %14s %14s %16s | %s\n", "N", "WGs", "work", "stream us/kern", "graph us/kern", "graph launch us", "host us per stream launch");
    for (int N : {16, 64}) {
        for (int wgs : {2, 48, 256}) {
            for (int work : {0, 2000}) {
                auto chain = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(step, dim3(wgs), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, work); };
                chain(); CK(hipStreamSynchronize(st));
                const int reps = 30;
                double t0 = now_us(), host = 0;
                for (int r = 0; r < reps; ++r) { double h0 = now_us(); chain(); host += now_us() - h0; CK(hipStreamSynchronize(st)); }
                const double t_stream = (now_us() - t0) / reps;
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                chain();
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
                double gl = 0;
                t0 = now_us();
                for (int r = 0; r < reps; ++r) { double h0 = now_us(); CK(hipGraphLaunch(ge, st)); gl += now_us() - h0; CK(hipStreamSynchronize(st)); }
                const double t_graph = (now_us() - t0) / reps;
                printf("%5d %5d %6d | %14.2f %14.2f %16.1f | %.2f\n", N, wgs, work, t_stream / N, t_graph / N, gl / reps, host / reps / N);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
        }
    }
    return 0;
}
