// lds_slot_probe.hip -- round 6: which values of HW_REG_LDS_ALLOC (id 6) the three co-resident 48 KB workgroups of a CU see, and in which
// order the dispatcher fills the chip (is workgroup b's layer on its CU = (b / 8) / 32 ?).  Needed by the three-phase stagger study of
// gemm_h2p (tools/r6_stagger3.sh): a workgroup must know which of its CU's slots it occupies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void probe(unsigned* out, int stay) {
    extern __shared__ float smem[];
    const unsigned lds = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 6);          // LDS_ALLOC bits 15:0
    const unsigned lds_hi = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (16 << 6) | 6);      // bits 31:16
    const unsigned hwid = __builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4);         // HW_ID bits 15:0: wave, simd, pipe, cu, sh, se
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int i = 0; i < stay; ++i) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[4 * blockIdx.x] = lds; out[4 * blockIdx.x + 1] = lds_hi; out[4 * blockIdx.x + 2] = hwid; out[4 * blockIdx.x + 3] = xcc; }
    if (smem[threadIdx.x] == 123.f) out[0] = 1;
}

int main() {
    const int G = 768;
    unsigned* d; CK(hipMalloc(&d, G * 16));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024));
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 48 * 1024, 0, d, 40);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(G * 4); CK(hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost));
    std::map<unsigned, int> lds_vals;
    std::map<unsigned long, std::vector<std::pair<int, unsigned>>> per_cu;
    for (int b = 0; b < G; ++b) {
        lds_vals[h[4 * b]]++;
        const unsigned hw = h[4 * b + 2], cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = h[4 * b + 3] & 15;
        per_cu[((unsigned long)xcc << 16) | (se << 8) | (sh << 4) | cu].push_back({b, h[4 * b]});
    }
    printf("distinct LDS_ALLOC[15:0] values:");
    for (auto& kv : lds_vals) printf("  0x%04x x%d", kv.first, kv.second);
    printf("\nLDS_ALLOC[31:16] of block 0: 0x%04x\n", h[1]);
    printf("CUs seen: %zu\n", per_cu.size());
    int shown = 0, agree = 0, total = 0;
    for (auto& kv : per_cu) {
        if (shown < 6) {
            printf("xcc %lu se %lu sh %lu cu %2lu:", kv.first >> 16, (kv.first >> 8) & 255, (kv.first >> 4) & 15, kv.first & 15);
            for (auto& p : kv.second) printf("  b=%3d (b/8)/32=%d lds=0x%04x", p.first, (p.first / 8) / 32, p.second);
            printf("\n");
            ++shown;
        }
        std::set<int> layers;
        for (auto& p : kv.second) layers.insert((p.first / 8) / 32);
        total++; agree += layers.size() == kv.second.size();
    }
    printf("CUs whose resident workgroups all have distinct (b/8)/32: %d of %d\n", agree, total);
    return 0;
}
