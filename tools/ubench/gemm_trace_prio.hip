// In-kernel timeline of the GEMM main loop (one wave of a mid-grid block): s_memtime stamps per k-tile.
#define LVAE_GEMM_TRACE 1
#define LVAE_EPI_PRIO 3
extern "C" { __device__ long* lvae_trace_buf; }
#include "../../lossy-vae_amd/csrc/gemm_f32.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 49152, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 1024;
    float *A, *W, *b, *o; long* tb;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&b, N * 4); hipMalloc(&o, (size_t)M * N * 4);
    hipMalloc(&tb, 16 * 8 * 8); hipMemset(tb, 0, 16 * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(lvae_trace_buf), &tb, sizeof(tb));
    std::vector<float> h((size_t)M * K); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, N * 4);
    lvae_gemm_desc d = {};
    d.A0 = A; d.lda0 = K; d.K0 = K; d.Wt = W; d.ldw = K; d.bias = b; d.out = o; d.ldo = N; d.M = M; d.N = N; d.K = K;
    d.epi = argc > 4 ? atoi(argv[4]) : 0; d.cfg = argc > 5 ? atoi(argv[5]) : 0; d.gamma = b; d.res = o; d.ldres = N;
    {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gemm_kernel<CfgD192, 0>, 256, CfgD192::LDS_BYTES);
        hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)gemm_kernel<CfgD192, 0>);
        printf("occupancy API: %d blocks/CU (err %d), regs %d, static smem %zu, dyn LDS %d, maxDyn %d\n", nb, (int)e, fa.numRegs, fa.sharedSizeBytes, CfgD192::LDS_BYTES, fa.maxDynamicSharedSizeBytes);
    }
    for (int i = 0; i < 3; ++i) lvae_gemm_f32(&d, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); lvae_gemm_f32(&d, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%d N=%d K=%d: %.1f us, %.1f TF/s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    long t[128]; hipMemcpy(t, tb, sizeof(t), hipMemcpyDeviceToHost);
    printf("kt : loads_issue  reads+mfma  vmcnt+lds_write  barrier  | iter_total (s_memtime ticks)\n");
    for (int kt = 0; kt < 16 && kt < K / 32 - 1; ++kt)
        printf("%2d : %6ld %8ld %8ld %8ld | %8ld\n", kt, t[kt*8+1]-t[kt*8+0], t[kt*8+2]-t[kt*8+1], t[kt*8+3]-t[kt*8+2],
               t[kt*8+4]-t[kt*8+3], t[(kt+1)*8+0]-t[kt*8+0]);
    printf("tile lifetime (ticks): prologue %ld, main loop %ld, epilogue %ld, total %ld\n", t[121] - t[120], t[122] - t[121], t[123] - t[122], t[123] - t[120]);
    return 0;
}
