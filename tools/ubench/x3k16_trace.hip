// In-kernel timeline of gemm_x3k16_kernel (wave 0 of a mid-grid block): s_memtime stamps per k16 stage.
//   x3k16_trace M N K epi TN [lds_pad_KiB]   (pad >= 20 forces one workgroup per CU)
#define LVAE_X3V2_TRACE 1
extern "C" { __device__ long* lvae_trace_buf; }
#include "../../lossy-vae_amd/csrc/gemm_x3v2.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 49152, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 384;
    int epi = argc > 4 ? atoi(argv[4]) : 0, tn = argc > 5 ? atoi(argv[5]) : 3;
    g_x3v2_lds_pad = argc > 6 ? atoi(argv[6]) * 1024 : 0;
    float *A, *b, *o; unsigned short* W; long* tb;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)6 * N * K * 2); hipMalloc(&b, N * 4); hipMalloc(&o, (size_t)M * N * 4);
    hipMalloc(&tb, 128 * 8); hipMemset(tb, 0, 128 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(lvae_trace_buf), &tb, sizeof(tb));
    std::vector<float> h((size_t)M * K); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    std::vector<unsigned short> hw((size_t)6 * N * K); for (auto& v : hw) v = 0x3c00 + (rand() & 0xff);
    hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(b, 0, N * 4);
    lvae_gemm_desc d = {};
    d.A0 = A; d.lda0 = K; d.K0 = K; d.Wt16 = W; d.ldw = K; d.bias = b; d.out = o; d.ldo = N; d.M = M; d.N = N; d.K = K;
    d.epi = epi; d.gamma = b; d.res = o; d.ldres = N; d.prec = 2;
    int rc = 0;
    for (int i = 0; i < 3; ++i) lvae_gemm_x3v2_try(&d, 0, tn, &rc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); lvae_gemm_x3v2_try(&d, 0, tn, &rc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%d N=%d K=%d epi=%d TN=%d pad=%d: %.1f us, %.1f TF/s (rc %d)\n", M, N, K, epi, tn, g_x3v2_lds_pad, ms * 1e3, 2.0 * M * N * K / ms / 1e9, rc);
    long t[128]; hipMemcpy(t, tb, sizeof(t), hipMemcpyDeviceToHost);
    printf("k16 stage : reads+grp0   rest  barrier | total   (s_memtime ticks; 12*TN MFMAs per stage = %d ticks of MFMA)\n", 12 * tn * 32);
    for (int q = 0; q < 28 && q < K / 16; ++q)
        printf("%2d : %6ld %8ld %8ld | %8ld\n", q, t[q*4+1]-t[q*4+0], t[q*4+2]-t[q*4+1], t[q*4+3]-t[q*4+2], t[q*4+3]-t[q*4+0]);
    printf("main loop %ld, epilogue %ld ticks\n", t[121] - t[120], t[122] - t[121]);
    return 0;
}
