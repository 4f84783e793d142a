// Does a packed f32 op whose LOW lane selects the HIGH dword of an operand (op_sel bit = 1) return wrong results while MFMA kernels share
// the CUs?  (DESIGN.md 5b called this an "erratum": csrc/dwconv_cl.hip saw wrong even-column pixels with `v_pk_fma_f32 ... op_sel:[0,1,0]`
// beside split-K GEMMs and has avoided the selection since.)  Probe: every lane runs chains of v_pk_fma_f32 with the weight taken from the
// high dword of src1 (op_sel:[0,1,0] op_sel_hi:[1,1,1]) -- the dwconv form -- and from src0 (op_sel:[1,0,0]), against the same chain in
// scalar v_fma_f32; a second stream runs an MFMA-dense kernel on the same CUs (both grids are 4 workgroups per CU of 256 threads).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// bad[0..1]: src1 form, low / high lane; bad[2..3]: src0 form; bad[4..5]: CONTROL (weight in the low dword, broadcast by op_sel_hi only:
// the form hipcc emits for scalar broadcasts and csrc/dwconv_cl.hip uses)
__global__ __launch_bounds__(256) void probe(const float* in, unsigned* bad, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x2 x = {in[t & 4095], in[(t * 7 + 1) & 4095]}, w = {in[(t * 3 + 2) & 4095] * 0.01f, in[(t * 5 + 3) & 4095] * 0.01f};
    unsigned nb[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        f32x2 a1 = {0.5f, 0.25f}, a0 = {0.5f, 0.25f}, ac = {0.5f, 0.25f};
        float s0 = 0.5f, s1 = 0.25f, c0 = 0.5f, c1 = 0.25f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a1) : "v"(x), "v"(w));   // both lanes x w.hi (src1)
            asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a0) : "v"(x), "v"(w));   // the same with w as src0
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ac) : "v"(x), "v"(w));                  // control: both lanes x w.lo
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(x[0]), "v"(w[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(x[1]), "v"(w[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(x[0]), "v"(w[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(x[1]), "v"(w[0]));
        }
        nb[0] += a1[0] != s0; nb[1] += a1[1] != s1; nb[2] += a0[0] != s0; nb[3] += a0[1] != s1; nb[4] += ac[0] != c0; nb[5] += ac[1] != c1;
        x[0] += 1e-3f; w[1] += 1e-5f;
    }
    for (int k = 0; k < 6; ++k) if (nb[k]) atomicAdd(bad + k, nb[k]);
}
__global__ __launch_bounds__(256) void mfma_load(float* out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.01f + e); b[e] = (_Float16)(1.0f - e * 0.1f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float *in, *out; unsigned* bad;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&bad, 24);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int mode = 0; mode < 2; ++mode) {                       // 0: probe alone; 1: beside the MFMA kernel
        unsigned long long total[6] = {0, 0, 0, 0, 0, 0}; long checks = 0;
        for (int rep = 0; rep < 200; ++rep) {
            hipMemsetAsync(bad, 0, 24, s1);
            if (mode) hipLaunchKernelGGL(mfma_load, dim3(1024), dim3(256), 0, s2, out, 4000);
            hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, s1, in, bad, 2000);
            unsigned hb[6]; hipMemcpyAsync(hb, bad, 24, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); hipStreamSynchronize(s2);
            for (int k = 0; k < 6; ++k) total[k] += hb[k];
            checks += 1024L * 256 * 2000;
        }
        printf("%s (200 launches, %ld chains of 16 packed FMAs per form and lane):\n  wrong results  op_sel on src1: low lane %llu, high lane %llu | op_sel on src0: low %llu, high %llu | control (no low-lane high-dword select): low %llu, high %llu\n",
               mode ? "beside an MFMA-dense kernel on a second stream" : "probe alone", checks, total[0], total[1], total[2], total[3], total[4], total[5]);
    }
    return 0;
}
