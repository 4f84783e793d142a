// cu_ingest.hip -- round 6: how many bytes per clock ONE compute unit can take in, by instruction kind and number of issuing waves.
// The f16x2 GEMMs stream 4 bytes per operand element; with 128 x 64 tiles a CU needs 0.094 B per multiply-accumulate, i.e. 64 B/clk to
// keep its matrix pipe busy (683 f16x2 MAC/clk/CU).  This measures what the memory path of a CU delivers:
//   dma  : buffer_load_dwordx4 ... lds (1 KiB per wave instruction, whole 128-B lines, lane-linear LDS image) into a ring of LDS slots,
//   vgpr : buffer_load_dwordx4 into registers (whole lines per 8 lanes), values folded into an accumulator,
//   stage: buffer_load_dwordx4 into registers, then ds_write_b128 into the same LDS ring (the pre-LDS-DMA way of filling LDS),
// W issuing waves per workgroup, one workgroup per CU on G of the chip's CUs, each workgroup re-reading its own region of R KiB `passes` times
// (R = 128: L2 hits after the first pass; R = 8192: every line from HBM / Infinity Cache).
// Output: GB/s per CU and B/clk/CU at the measured kernel time (clock from hipDeviceAttributeClockRate, reported).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ __launch_bounds__(1024, 1) void k_dma(const char* base, long region, int passes, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* mine = base + (long)blockIdx.x * region;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, (int)region, 0x00020000);
    char* slot = (char*)smem + wave * INFLIGHT * 1024;
    const int pieces = (int)(region / 1024);                       // 1 KiB pieces, dealt to the waves round-robin
    for (int p = 0; p < passes; ++p) {
        int k = 0;
        for (int i = wave; i < pieces; i += nw, ++k) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + (k % INFLIGHT) * 1024), 16, lane * 16, i * 1024, 0, 0);
            if (k % INFLIGHT == INFLIGHT - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT / 2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (smem[threadIdx.x] == 123.456f) *sink = 1.f;
}

template <int INFLIGHT>
__global__ __launch_bounds__(1024, 1) void k_vgpr(const char* base, long region, int passes, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* mine = base + (long)blockIdx.x * region;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, (int)region, 0x00020000);
    const int pieces = (int)(region / 1024);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < passes; ++p) {
        for (int i0 = wave; i0 < pieces; i0 += nw * INFLIGHT) {
            u32x4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (i0 + j * nw) * 1024, 0);
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) acc += __builtin_bit_cast(f32x4, v[j]);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) *sink = acc[0];
}

// the same bytes into the same LDS ring through registers: buffer_load_dwordx4 -> VGPR -> ds_write_b128
template <int INFLIGHT>
__global__ __launch_bounds__(1024, 1) void k_stage(const char* base, long region, int passes, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char* mine = base + (long)blockIdx.x * region;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, (int)region, 0x00020000);
    const unsigned slot = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)((char*)smem + wave * INFLIGHT * 1024) + lane * 16;
    const int pieces = (int)(region / 1024);
    for (int p = 0; p < passes; ++p) {
        for (int i0 = wave; i0 < pieces; i0 += nw * INFLIGHT) {
            u32x4 v[INFLIGHT];
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (i0 + j * nw) * 1024, 0);
#pragma unroll
            for (int j = 0; j < INFLIGHT; ++j) asm volatile("ds_write_b128 %0, %1" ::"v"(slot + j * 1024), "v"(v[j]) : "memory");     // (asm: a plain store is dead but for the last pass)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (smem[threadIdx.x] == 123.456f) *sink = 1.f;
}

template <class F> float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double ghz = khz / 1e6;
    printf("clock rate attribute %.2f GHz (B/clk figures use it; the chip may run below it under load)\n", ghz);
    const long total = 2048L << 20;
    char* buf; float* sink;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, total));
    CK(hipFuncSetAttribute((const void*)k_dma<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_stage<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("%-5s %4s %7s %6s | %10s %10s %9s\n", "kind", "CUs", "regionK", "waves", "GB/s/CU", "B/clk/CU", "chip TB/s");
    for (long regionK : {128L, 512L, 8192L}) {
        const long region = regionK << 10;
        const int passes = regionK == 128 ? 32 : (regionK == 512 ? 8 : 1);
        for (int G : {32, 48, 128, 256}) {
            if ((long)G * region > total) continue;
            for (int W : {1, 2, 4, 8, 16}) {
                const double bytes = (double)region * passes;
                if (W <= 8) {
                    const float t = timeit([&] { hipLaunchKernelGGL(k_dma<16>, dim3(G), dim3(64 * W), W * 16 * 1024, 0, buf, region, passes, sink); }, 5);
                    printf("%-5s %4d %7ld %6d | %10.1f %10.1f %9.2f\n", "dma", G, regionK, W, bytes / t / 1e6, bytes / t / 1e6 / ghz, bytes * G / t / 1e9);
                } else {
                    const float t = timeit([&] { hipLaunchKernelGGL(k_dma<8>, dim3(G), dim3(64 * W), W * 8 * 1024, 0, buf, region, passes, sink); }, 5);
                    printf("%-5s %4d %7ld %6d | %10.1f %10.1f %9.2f\n", "dma", G, regionK, W, bytes / t / 1e6, bytes / t / 1e6 / ghz, bytes * G / t / 1e9);
                }
                const float t2 = timeit([&] { hipLaunchKernelGGL(k_vgpr<8>, dim3(G), dim3(64 * W), 0, 0, buf, region, passes, sink); }, 5);
                printf("%-5s %4d %7ld %6d | %10.1f %10.1f %9.2f\n", "vgpr", G, regionK, W, bytes / t2 / 1e6, bytes / t2 / 1e6 / ghz, bytes * G / t2 / 1e9);
                const float t3 = timeit([&] { hipLaunchKernelGGL(k_stage<8>, dim3(G), dim3(64 * W), W * 8 * 1024, 0, buf, region, passes, sink); }, 5);
                printf("%-5s %4d %7ld %6d | %10.1f %10.1f %9.2f\n", "stage", G, regionK, W, bytes / t3 / 1e6, bytes / t3 / 1e6 / ghz, bytes * G / t3 / 1e9);
            }
        }
    }
    // in-flight depth sweep: is a wave's LDS-DMA rate its issue rate or (pieces in flight) / latency?
    printf("\n%-5s %4s %7s %6s %9s | %10s %10s\n", "kind", "CUs", "regionK", "waves", "in flight", "GB/s/CU", "B/clk/CU");
    CK(hipFuncSetAttribute((const void*)k_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_dma<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_dma<60>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (long regionK : {128L, 8192L}) {
        const long region = regionK << 10;
        const int passes = regionK == 128 ? 32 : 1;
        const double bytes = (double)region * passes;
        for (int G : {48, 256}) {
            for (int W : {1, 2, 4}) {
#define SWEEP(NF) if (W * NF <= 150) { const float t = timeit([&] { hipLaunchKernelGGL(k_dma<NF>, dim3(G), dim3(64 * W), W * NF * 1024, 0, buf, region, passes, sink); }, 5); \
                    printf("%-5s %4d %7ld %6d %9d | %10.1f %10.1f\n", "dma", G, regionK, W, NF, bytes / t / 1e6, bytes / t / 1e6 / ghz); }
                SWEEP(4) SWEEP(8) SWEEP(16) SWEEP(32) SWEEP(60)
#undef SWEEP
            }
        }
    }
    return 0;
}
