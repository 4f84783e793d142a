// gemm_h2e.hip -- fc1 of a ConvNeXt block (pre-split operands in, bias + exact-erf GELU, pre-split result out: lvae/models/common.py:131-132,154)
// as a PERSISTENT f16x2 GEMM whose epilogue is issued by the same waves BETWEEN the MFMAs of the next tile's main loop.
//
// Why.  In gemm_h2p_kernel the fused epilogue is 55-60 % of an MLP launch (profiles/r04_gemm_h2p_ablations.txt: 49152 x 768 x 384 with
// GELU 124.6 us, main loop alone 52.4 us): 30 VALU instructions per element -- GELU, f16x2 split, quad transpose -- at ~4 cycles per
// wave-instruction and SIMD, issued while no MFMA of that wave is in flight; the CU's other workgroup does not fill the gap (priorities,
// stagger, MFMA order: measured).  What does work on this chip is SAME-wave interleaving: six VALU instructions per MFMA issue
// for free, every further one costs ~4.3 cycles (tools/ubench/mfma_valu_interleave.hip, two waves per SIMD).  So:
//   * a workgroup (4 waves, 128 x 64 tiles, two workgroups per CU) walks a static list of tiles; its LDS-DMA ring (3 x 24 KB) runs flat
//     across tile boundaries (the next tile's first two stages are in flight while this tile ends);
//   * when a tile's K loop ends, its accumulators are folded into 32 registers (fma(accX, 2^-11, accH)) and the wave moves on; the
//     epilogue of THAT tile -- 8 units of 4 elements per lane: + bias, GELU, quad transpose, split -- is cut into six slices per unit and
//     one slice follows each MFMA pair of the NEXT tile's first eight stages (K >= 256: at least eight stages);
//   * the 16 stores of a finished tile are issued at the next tile boundary, behind a vmcnt(0) that the ring needs there anyway (gfx950 has
//     ONE counter for loads and stores: stores interleaved with the DMA-pipelined loop would turn every counted wait into a wait for them).
// Arithmetic: per accumulator the MFMA sequence of gemm_h2p_kernel, the same fold, gemm_epilogue's operations in its order (device_math.h's
// erf polynomial written out step by step: same operations, same roundings, no contraction) -- every output bit equals gemm_h2p's
// (tools/h2e_equal.py: 0 words differ on seven shapes).
// STUDY, NOT PART OF liblvae_hip.so (like gemm_h2pp.hip): it lost.  EXTRA_SRC=gemm_h2e.hip tools/build_exp.sh h2e gemm_h2p.hip -DLVAE_EXP_H2E
// links it behind d.cfg = 51.  Measured (profiles/r04_gemm_h2e_interleaved_epilogue_study.txt): 49152 x 768 x 384 157-161 us against
// gemm_h2p's 127-131 (128 x 128 tiles).  Its ablations say why: the persistent 128 x 64 main loop alone is 94 us (gemm_h2p's 128 x 64:
// 88, its 128 x 128: 52 -- with two accumulator pairs per wave the loop is bound by LDS fill and fragment reads, and 128 x 128 per four
// waves + eight units of epilogue state do not fit 256 registers); the interleaved slices add 32 us to it -- ten VALU instructions per
// MFMA where six are free -- and the stores behind the tile-boundary vmcnt(0) another 35.
#include "../../lossy-vae_amd/csrc/gemm_common.h"

#include <type_traits>
#include <utility>

#if !defined(LVAE_EXPERIMENTAL_BUILD) && (defined(H2E_EXP_NOSLICE) || defined(H2E_EXP_NOSTORE) || defined(H2E_EXP_NOMFMA))
#error "H2E_EXP_* timing ablations (wrong results) need -DLVAE_EXPERIMENTAL_BUILD (tools/build_exp.sh)"
#endif
#ifdef H2E_EXP_NOMFMA
#define H2E_MFMA(a, b, c) (c)
#else
#define H2E_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LVAE_FENCE() __builtin_amdgcn_sched_barrier(0)
#define H2E_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))

template <int N, class F, int... I>
__device__ __forceinline__ void h2e_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void h2e_static_for(F&& f) { h2e_static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// one epilogue unit (4 elements per lane = two packed pairs) in flight between its six slices
struct H2EUnit {
    lvae_f2 x[2], a[2], t[2], s[2], r[2], q[2];
    float ex[4];
};

template <int NQ>                     // K / 32: 8, 12 or 16 (fc1 of the C = 256 / 384 / 512 blocks)
__global__ __launch_bounds__(256, 2) void gemm_h2e_kernel(const lvae_gemm_desc d, int tiles_n, int n_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang fp contract(off)
    constexpr int BM = 128, BN = 64, ROWS = BM + BN, STAGE = ROWS * 128, NBUF = 3, NI = ROWS / 8 / 4, NU = 8;
    constexpr int ROWB = NQ * 128;                                   // bytes of one H2K32 row (A and W alike)
    static_assert(NQ >= NU && NI == 6, "one epilogue unit per stage; one DMA instruction per MFMA pair");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)((char*)smem);

    // ---- tiles of this workgroup: linear index blockIdx.x + k * gridDim.x, mapped so that consecutive tiles (n fastest) stay on one XCD
    // (gridDim.x is a multiple of 8 whenever there is more than one tile per workgroup: the linear index keeps the workgroup's XCD)
    auto tile_of = [&](int k) -> int {
        const int b = blockIdx.x + k * gridDim.x;
        if (b >= n_tiles) return -1;
        const int q = n_tiles / 8, r = n_tiles % 8, xcd = b % 8, loc = b / 8;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    };

    // ---- DMA side (gemm_h2p.hip): instruction g of a stage covers its rows 8g .. 8g + 7; wave w issues g = i * 4 + w
    const int r_in = lane >> 3, pp = lane & 7;
    const int dvoff = r_in * ROWB + ((pp ^ ((4 * (wave & 1) + (r_in >> 1)) & 7)) << 4);
    auto dma = [&](int i, int t, int q, int slot) __attribute__((always_inline)) {       // instruction i of stage q of tile t (t < 0: nothing to fetch)
        int wv = wave;
        asm volatile("" : "+s"(wv));
        const int g = i * 4 + wv;
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        char* dst = (char*)smem + slot * STAGE + g * 1024;
        if (i < 4) {                                                   // A rows
            const int m0 = tm * BM;
            const int rows = t < 0 ? 0 : ((d.M - m0) < BM ? (d.M - m0) : BM);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.A0 + (long)(t < 0 ? 0 : m0) * ROWB), 0, rows * ROWB, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, dvoff, 8 * g * ROWB + q * 128, 0, 0);
        } else {                                                       // W rows
            const int n0 = tn * BN;
            const int rows = t < 0 ? 0 : ((d.N - n0) < BN ? (d.N - n0) : BN);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.Wt16 + (long)(t < 0 ? 0 : n0) * ROWB), 0, rows * ROWB, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, dvoff, (8 * g - BM) * ROWB + q * 128, 0, 0);
        }
    };

    // ---- fragment side: piece (plane p, k16 step t, lane half) = 4p + 2t + lh at ((piece ^ x) << 4) of the lane's row
    const int xr = (li >> 1) & 7;
    unsigned po[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) po[pt] = (unsigned)(((4 * (pt >> 1) + 2 * (pt & 1) + lh) ^ xr) << 4);
    const unsigned a_row = lds0 + (wave_m * 64 + li) * 128, b_row = lds0 + (BM + wave_n * 32 + li) * 128;

    f32x16 accH[2], accX[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accH[a][r] = 0.f; accX[a][r] = 0.f; }
    // the PREVIOUS tile: folded accumulators, bias of its columns, its position, and the finished results (hi | lo' pairs per unit)
    float pv[2][16];
    float pbias = 0.f;
    int p_m0 = 0, p_n0 = 0;
    u32x2_t res[NU][2];
    bool have_prev = false;
    H2EUnit U;

    // the six slices of epilogue unit u (a = u >> 2: 32-row block, g = u & 3: rows 8g .. 8g + 3 of a lane half) -- gemm_epilogue's
    // "+ bias, gelu_erf2 x 2, quad_transpose, split_pair_h2 x 2" with lvae_erff2 written out step by step
    auto slice = [&](auto utag, auto stag) __attribute__((always_inline)) {
        constexpr int u = decltype(utag)::value, sl = decltype(stag)::value, a = u >> 2, g = u & 3;
        if constexpr (sl == 0) {
            U.x[0] = (lvae_f2){pv[a][4 * g + 0] + pbias, pv[a][4 * g + 1] + pbias};
            U.x[1] = (lvae_f2){pv[a][4 * g + 2] + pbias, pv[a][4 * g + 3] + pbias};
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                U.a[p] = U.x[p] * (lvae_f2)(0.70710678118654752440f);
                U.t[p] = (lvae_f2){fabsf(U.a[p][0]), fabsf(U.a[p][1])};
                U.s[p] = U.a[p] * U.a[p];
            }
        } else if constexpr (sl == 1) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                lvae_f2 r = __builtin_elementwise_fma((lvae_f2)(-1.72853470e-5f), U.t[p], (lvae_f2)(3.83197126e-4f));
                const lvae_f2 uu = __builtin_elementwise_fma((lvae_f2)(-3.88396438e-3f), U.t[p], (lvae_f2)(2.42546219e-2f));
                r = __builtin_elementwise_fma(r, U.s[p], uu);
                U.r[p] = __builtin_elementwise_fma(r, U.t[p], (lvae_f2)(-1.06777877e-1f));
                lvae_f2 q = __builtin_elementwise_fma((lvae_f2)(-5.96761703e-4f), U.s[p], (lvae_f2)(4.99119423e-3f));
                U.q[p] = __builtin_elementwise_fma(q, U.s[p], (lvae_f2)(-2.67681349e-2f));
            }
        } else if constexpr (sl == 2) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                lvae_f2 r = __builtin_elementwise_fma(U.r[p], U.t[p], (lvae_f2)(-6.34846687e-1f));
                r = __builtin_elementwise_fma(r, U.t[p], (lvae_f2)(-1.28717512e-1f));
                U.r[p] = __builtin_elementwise_fma(r, U.t[p], -U.t[p]);
                lvae_f2 q = __builtin_elementwise_fma(U.q[p], U.s[p], (lvae_f2)(1.12819925e-1f));
                q = __builtin_elementwise_fma(q, U.s[p], (lvae_f2)(-3.76125336e-1f));
                U.q[p] = __builtin_elementwise_fma(q, U.s[p], (lvae_f2)(1.28379166e-1f));
            }
        } else if constexpr (sl == 3) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                U.q[p] = __builtin_elementwise_fma(U.q[p], U.a[p], U.a[p]);
                U.ex[2 * p + 0] = __expf(U.r[p][0]);
                U.ex[2 * p + 1] = __expf(U.r[p][1]);
            }
        } else if constexpr (sl == 4) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                lvae_f2 o;
                o[0] = U.t[p][0] > 0.927734375f ? copysignf(1.0f - U.ex[2 * p + 0], U.a[p][0]) : U.q[p][0];
                o[1] = U.t[p][1] > 0.927734375f ? copysignf(1.0f - U.ex[2 * p + 1], U.a[p][1]) : U.q[p][1];
                U.x[p] = ((lvae_f2)(0.5f) * U.x[p]) * ((lvae_f2)(1.0f) + o);
            }
        } else {
            int lio = li;
            asm volatile("" : "+v"(lio));
            float v0 = U.x[0][0], v1 = U.x[0][1], v2 = U.x[1][0], v3 = U.x[1][1];
            quad_transpose(v0, v1, v2, v3, lio & 3);
            unsigned h0, l0, h1, l1;
            split_pair_h2(v0, v1, h0, l0);
            split_pair_h2(v2, v3, h1, l1);
            res[u][0] = (u32x2_t){h0, h1};
            res[u][1] = (u32x2_t){l0, l1};
        }
    };
    // the 2 * NU stores of the previous tile (gemm_epilogue's addresses: H2K32, ldo = N)
    auto store_prev = [&]() __attribute__((always_inline)) {
        int lio = li, lho = lh;
        asm volatile("" : "+v"(lio), "+v"(lho));
        const int c4 = p_n0 + wave_n * 32 + (lio & ~3);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int row = p_m0 + wave_m * 64 + (u >> 2) * 32 + 4 * lho + 8 * (u & 3) + (lio & 3);
#ifdef H2E_EXP_NOSTORE
            if (row < d.M && c4 < d.N && res[u][0][0] == 0x12345678u) {
#else
            if (row < d.M && c4 < d.N) {
#endif
                char* q = (char*)d.out + (((long)row * d.ldo) << 2) + ((c4 >> 5) << 7) + ((c4 & 31) << 1);
                *(u32x2_t*)q = res[u][0];
                *(u32x2_t*)(q + 64) = res[u][1];
            }
        }
    };

    int k = 0;
    int tile = tile_of(0);
    if (tile < 0) return;
    int s0 = 0, s1 = 1, s2 = 2;                                        // ring slots of this stage, the next one, the one after
    // prologue: stages 0 and 1 of the first tile
#pragma unroll
    for (int i = 0; i < NI; ++i) dma(i, tile, 0, 0);
#pragma unroll
    for (int i = 0; i < NI; ++i) dma(i, tile, 1, 1);
    bool first = true;
    while (tile >= 0) {
        const int nxt = tile_of(k + 1);
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
        const float cbias = d.bias ? d.bias[(n0 + wave_n * 32 + li) < d.N ? (n0 + wave_n * 32 + li) : 0] : 0.f;

        h2e_static_for<NQ>([&](auto ptag) {
            constexpr int P = decltype(ptag)::value;
            // my DMA instructions of this stage have landed once only the next stage's NI are outstanding; stages 0 and 1 of a tile that
            // follows another one were waited for before that tile's stores (below)
            if (P >= 2 || first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            asm volatile("s_barrier" ::: "memory");
            LVAE_FENCE();
            unsigned aq = a_row + s0 * STAGE, bq = b_row + s0 * STAGE;
            asm volatile("" : "+v"(aq), "+v"(bq));
            f16x8 af[2][2][2], bf[2][2];                                 // [t][a][plane], [t][plane]
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    H2E_DSR(af[tt][a][0], aq + po[0 + tt], a * 4096);
                    H2E_DSR(af[tt][a][1], aq + po[2 + tt], a * 4096);
                }
                H2E_DSR(bf[tt][0], bq + po[0 + tt], 0);
                H2E_DSR(bf[tt][1], bq + po[2 + tt], 0);
            }
            // stage P + 2: of this tile, or of the next one
            const int t2 = P + 2 < NQ ? tile : nxt;
            constexpr int q2 = (P + 2) % NQ;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                if (tt == 0) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(af[0][0][0]), "+v"(af[0][0][1]), "+v"(af[0][1][0]), "+v"(af[0][1][1]), "+v"(bf[0][0]), "+v"(bf[0][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[1][0][0]), "+v"(af[1][0][1]), "+v"(af[1][1][0]), "+v"(af[1][1][1]), "+v"(bf[1][0]), "+v"(bf[1][1]));
                LVAE_FENCE();
                h2e_static_for<3>([&](auto jtag) {
                    constexpr int j = decltype(jtag)::value;
                    if constexpr (j == 0) {
                        accX[0] = H2E_MFMA(af[tt][0][1], bf[tt][0], accX[0]);
                        accX[1] = H2E_MFMA(af[tt][1][1], bf[tt][0], accX[1]);
                    } else if constexpr (j == 1) {
                        accX[0] = H2E_MFMA(af[tt][0][0], bf[tt][1], accX[0]);
                        accX[1] = H2E_MFMA(af[tt][1][0], bf[tt][1], accX[1]);
                    } else {
                        accH[0] = H2E_MFMA(af[tt][0][0], bf[tt][0], accH[0]);
                        accH[1] = H2E_MFMA(af[tt][1][0], bf[tt][0], accH[1]);
                    }
                    dma(tt * 3 + j, t2, q2, s2);                       // one DMA instruction behind each MFMA pair ...
#ifndef H2E_EXP_NOSLICE
                    if constexpr (P < NU) {                            // ... and one slice of the previous tile's epilogue unit P
                        if (have_prev) {
                            if (tt == 0) slice(std::integral_constant<int, P>{}, std::integral_constant<int, j>{});
                            else slice(std::integral_constant<int, P>{}, std::integral_constant<int, 3 + j>{});
                        }
                    }
#endif
                    LVAE_FENCE();
                });
            }
            const int sn = s0; s0 = s1; s1 = s2; s2 = sn;
        });
        first = false;

        // ---- tile boundary: the next tile's stages 0 and 1 (in flight since this tile's last two stages) are waited for HERE, before the
        // stores of the tile before this one; then this tile's accumulators are folded and become "previous"
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (have_prev) store_prev();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                pv[a][r] = __builtin_fmaf(accX[a][r], 1.0f / 2048.0f, accH[a][r]);
                accH[a][r] = 0.f; accX[a][r] = 0.f;
            }
        pbias = cbias; p_m0 = m0; p_n0 = n0; have_prev = true;
        tile = nxt;
        ++k;
    }
    // ---- the last tile's epilogue, on its own
    h2e_static_for<NU>([&](auto utag) {
        h2e_static_for<6>([&](auto stag) { slice(utag, stag); });
    });
    store_prev();
#endif
}

template <int NQ>
int launch_h2e(const lvae_gemm_desc* d, hipStream_t st) {
    constexpr int LDS = 3 * 192 * 128;
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)gemm_h2e_kernel<NQ>, LDS)) return ae;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int tiles_m = (d->M + 127) / 128, tiles_n = (d->N + 63) / 64, n_tiles = tiles_m * tiles_n;
    int grid = 2 * n_cu;                                               // two persistent workgroups per CU
    grid -= grid % 8;
    if (n_tiles < grid) grid = n_tiles;
    hipLaunchKernelGGL((gemm_h2e_kernel<NQ>), dim3(grid), dim3(256), LDS, st, *d, tiles_n, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace

// Entry point for gemm_h2p.hip's chooser: -> 1 when this file takes the GEMM (fc1 form: pre-split in and out, bias + GELU, no split-K).
int lvae_gemm_h2e_try(const lvae_gemm_desc* d, hipStream_t st, int* rc) {
    if (d->prec != 4 || !d->a_h2 || !d->out_h2 || d->epi != LVAE_EPI_BIAS_GELU || d->store != LVAE_ST_ROWMAJOR || d->ksplit > 1 || d->ldo != d->N ||
        (d->N & 31) || (long)d->M * d->K * 4 > 0x7fffffffL)
        return 0;
    switch (d->K) {
        case 256: *rc = launch_h2e<8>(d, st); return 1;
        case 384: *rc = launch_h2e<12>(d, st); return 1;
        case 512: *rc = launch_h2e<16>(d, st); return 1;
        default: return 0;
    }
}
