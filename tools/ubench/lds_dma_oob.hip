// Does an out-of-range lane of an LDS-DMA load (buffer_load_dwordx4 ... lds) write ZERO to LDS or leave LDS untouched?
// Kernel A fills every CU's LDS with a NaN pattern; kernel B loads through an EMPTY descriptor (num_records = 0), through a
// descriptor that covers only the first half of the lanes, and through a full one, and copies its LDS out.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void fill(float* sink) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = __uint_as_float(0x7fc01234u);
    __syncthreads();
    if (sink && lds[threadIdx.x] == 1.f) sink[0] = 1.f;
}
__global__ void probe(const float* src, float* out, int mode) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const int recs = mode == 0 ? 0 : (mode == 1 ? 32 * 16 : 64 * 16);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, recs, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[(blockIdx.x * 64 + lane) * 4 + i] = lds[lane * 4 + i];
}
int main() {
    float *src, *out, *sink;
    const int NB = 2048;
    hipMalloc(&src, 4096); hipMalloc(&out, NB * 64 * 16); hipMalloc(&sink, 4);
    std::vector<float> h(1024, 5.0f); hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)fill, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(fill, dim3(NB), dim3(256), 65536, 0, (float*)nullptr);
        hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 65536, 0, src, out, mode);
        hipDeviceSynchronize();
        std::vector<float> o(NB * 64 * 4); hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost);
        long zeros = 0, fives = 0, nans = 0, other = 0, oob_zero = 0, oob_nan = 0;
        for (int b = 0; b < NB; ++b)
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 4; ++i) {
                    const float v = o[(b * 64 + l) * 4 + i];
                    const bool oob = mode == 0 || (mode == 1 && l >= 32);
                    if (v == 0.f) { ++zeros; if (oob) ++oob_zero; } else if (v == 5.f) ++fives; else if (v != v) { ++nans; if (oob) ++oob_nan; } else ++other;
                }
        printf("mode %d (records %s): zeros %ld, loaded values %ld, stale NaNs %ld, other %ld | out-of-range lanes: %ld zero, %ld stale\n", mode,
               mode == 0 ? "0" : (mode == 1 ? "first 32 lanes" : "all 64 lanes"), zeros, fives, nans, other, oob_zero, oob_nan);
    }
    return 0;
}
