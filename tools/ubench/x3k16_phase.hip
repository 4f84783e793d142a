// Phase relation of the two co-resident workgroups of a CU in gemm_x3k16_kernel: every workgroup's wave 0 stamps each k16 stage
// (s_memtime) and records HW_ID / XCC_ID / LDS_ALLOC; the host groups workgroups by CU and prints, for one CU, the per-stage
// timeline statistics and where workgroup A's barriers fall inside workgroup B's stages.
//   x3k16_phase M N K epi TN
#define LVAE_X3V2_TRACE 2
extern "C" { __device__ long* lvae_trace_buf; }
#include "../../lossy-vae_amd/csrc/gemm_x3v2.hip"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>
int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 49152, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 4096;
    int epi = argc > 4 ? atoi(argv[4]) : 0, tn = argc > 5 ? atoi(argv[5]) : 3;
    const int nq = K / 16 > 256 ? 256 : K / 16;
    const int tiles = ((M + 127) / 128) * ((N + 64 * tn - 1) / (64 * tn));
    const size_t REC = 8 + 4 * 256;
    float *A, *b, *o; unsigned short* W; long* tb;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)6 * N * K * 2); hipMalloc(&b, N * 4); hipMalloc(&o, (size_t)M * N * 4);
    hipMalloc(&tb, tiles * REC * 8); hipMemset(tb, 0, tiles * REC * 8);
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
    for (int i = 0; i < 2; ++i) lvae_gemm_x3v2_try(&d, 0, tn, &rc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); lvae_gemm_x3v2_try(&d, 0, tn, &rc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("M=%d N=%d K=%d epi=%d TN=%d: %.1f us, %.1f TF/s (rc %d), %d tiles\n", M, N, K, epi, tn, ms * 1e3, 2.0 * M * N * K / ms / 1e9, rc, tiles);
    std::vector<long> t(tiles * REC); hipMemcpy(t.data(), tb, tiles * REC * 8, hipMemcpyDeviceToHost);
    std::map<long, std::vector<int>> by_cu;
    for (int i = 0; i < tiles; ++i) {
        const long hwid = t[i * REC + 2], xcc = t[i * REC + 3] & 0xf;
        const long cu = (hwid >> 8) & 0xf, sh = (hwid >> 12) & 1, se = (hwid >> 13) & 7;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(i);
    }
    printf("%zu distinct CUs seen\n", by_cu.size());
    int shown = 0;
    double sum_tot = 0, sum_g0 = 0, sum_rest = 0, sum_bar = 0; long nst = 0;
    std::vector<int> hist(10, 0);
    for (auto& kv : by_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int c) { return t[a * REC] < t[c * REC]; });
        const long t0 = t[v[0] * REC];
        if (shown < 2) {
            printf("CU key %lx: %zu workgroups\n", kv.first, v.size());
            for (int i : v)
                printf("  block %4d  start %8ld  loop_end %8ld  end %8ld  wave_slot %ld simd %ld  lds_alloc %#lx\n", i, t[i * REC] - t0, t[i * REC + 1] - t0,
                       t[i * REC + 5] - t0, t[i * REC + 2] & 0xf, (t[i * REC + 2] >> 4) & 3, t[i * REC + 4]);
            ++shown;
        }
        for (int i : v)
            for (int q = 2; q < nq - 1; ++q) {
                const long* s = &t[i * REC + 8 + q * 4];
                sum_g0 += s[1] - s[0]; sum_rest += s[2] - s[1]; sum_bar += s[3] - s[2]; sum_tot += s[4] - s[0]; ++nst;
            }
        // phase: for each pair of time-overlapping workgroups, position of A's barrier-exit stamps inside B's stages
        for (size_t a = 0; a < v.size(); ++a)
            for (size_t c = 0; c < v.size(); ++c) {
                if (a == c) continue;
                const long* sa = &t[v[a] * REC + 8];
                const long* sb = &t[v[c] * REC + 8];
                for (int q = 2; q < nq - 1; ++q) {
                    const long x = sa[q * 4 + 3];
                    for (int r = 2; r < nq - 1; ++r)
                        if (sb[r * 4] <= x && x < sb[(r + 1) * 4]) {
                            const int bin = (int)(10.0 * (x - sb[r * 4]) / (sb[(r + 1) * 4] - sb[r * 4]));
                            ++hist[bin < 10 ? bin : 9];
                            break;
                        }
                }
            }
    }
    printf("per stage (ticks, all workgroups, stages 2..%d): stage period %.0f = reads+grp0 %.0f + rest %.0f + barrier %.0f + loop overhead; MFMA %d\n", nq - 2,
           sum_tot / nst, sum_g0 / nst, sum_rest / nst, sum_bar / nst, 12 * tn * 32);
    printf("phase of a workgroup's barrier exit inside its CU-mate's stage (10 bins over the stage; in phase = bins 0 and 9 heavy):\n ");
    long tot = 0; for (int x : hist) tot += x;
    for (int x : hist) printf(" %5.1f%%", 100.0 * x / (tot ? tot : 1));
    printf("\n");
    return 0;
}
