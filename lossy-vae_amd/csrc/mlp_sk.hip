// mlp_sk.hip -- the MLP of a ConvNeXt block on the SMALL maps (stride 32 / 64 of the codec: 96 ... 3072 rows per launch, C = 512,
// hidden = 1024 ... 2048), fc1 -> GELU -> fc2 partial sums as ONE launch + the split-K reduce launch
// (reference: lvae/models/common.py:131-132,154-158 -- `mlp(x)` of ConvNeXtBlockAdaLN, one nn.Module there).
//
// Why.  On these maps a launch has a handful of row tiles and megabytes of weights, so both GEMMs run split-K (lvae/engine.py::
// auto_ksplit: S1 = 2 ... 4 slices of fc1's K = C, S2 = 8 ... 16 slices of fc2's K = hidden; the slice counts fix the summation order and
// with it every bit of the bitstream contract).  Rounds 3-5 ran them as two or three launches -- fc1 (serial split-K, gemm_h2p FOLD), fc2
// (parallel split-K, gemm_h2) + reduce, or fc2 serial -- 35-45 us per block on the dependency chain of BOTH the encoder and the decoder,
// although a block is 1.6-4.8 GFLOP and 6-8 MB of weights: no launch of the chain takes less than ~5 us and the long-K fc2 tiles cannot
// fill the chip.  The observation behind this kernel: fc2's K slice c IS a contiguous range of hidden columns [c CH, (c + 1) CH),
// CH = hidden / S2 -- so the workgroup (row tile, slice c) can compute exactly those hidden columns itself (fc1 restricted to CH output
// columns: all S1 slices of its K = C, folded in slice order like gemm_h2p's FOLD form), apply bias / GELU / the f16x2 split in
// registers, keep the 32 x CH hidden tile in LDS, and multiply it with slice c of W2: the partial sums of fc2's slice c for its rows --
// the value the parallel form writes to plane c of the workspace.  The reduce launch (gemm_f32.hip: splitk_reduce_kernel, unchanged) then
// adds the planes in slice order and applies bias / gamma / residual.  Per accumulator the MFMA sequence, the fold and every
// elementwise operation are those of the launches replaced, hence the same bits (tests/test_gpu_f16x2.py::test_mlp_sk_equals_split_k_gemms):
// which form a plan takes is a question of speed only and may depend on the batch.
//
// Shape of a workgroup: 32 rows x one slice; NW = CH / 32 waves, each owning ONE 32-column block of the hidden chunk in fc1 and, in
// fc2, one 32-column block of every group of CH output columns (NGRP = ceil(C / CH) accumulator pairs).  Everything streams through one
// LDS ring of 3 slots by LDS-DMA in whole 128-B lines (`buffer_load ... lds`: gemm_h2p.hip's scheme -- lane-linear LDS image, bank
// conflicts removed by permuting the SOURCE address, one raw s_barrier + counted vmcnt per unit): fc1 unit = one k32 stage of the 32
// y rows + the CH rows of W1; fc2 unit = one k32 stage of CH rows of W2.  A workgroup moves (32 + CH) C 4 + C CH 4 bytes (0.55-1.1 MB, L2
// hits for all but the first row tile of a slice: the grid is slice-major per XCD) for 12 CH C 32 MFMA-flop: DMA-bound by design --
// the point is the number of dependent launches, not the matrix pipe.
#include "gemm_common.h"

#include <cstdlib>

int lvae_splitk_reduce_launch(const lvae_gemm_desc* d, int S, hipStream_t st);        // gemm_f32.hip

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define SK_FENCE() __builtin_amdgcn_sched_barrier(0)
#define SK_DSR(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define SK_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// Roles.  NW = CH / 32 COMPUTE waves (fragment reads + MFMAs + the two epilogues) and NL LOADER waves that do nothing but issue the
// LDS-DMA pieces (8 rows x 128 B = 1 KiB per wave instruction) and wait for them.  Why loaders, and why eight of them
// (profiles/r06_cu_ingest.txt, tools/ubench/cu_ingest.hip): ONE wave moves 5.7 B/clk by LDS-DMA whatever it waits on (13.8 GB/s: ~180
// cycles per piece), W waves W times that up to the CU's ~54 B/clk -- so a workgroup's streaming rate is its number of ISSUING waves.
// First form of this kernel: the four compute waves issued the pieces between their MFMAs (gemm_h2p's scheme): 12.8 us per workgroup
// for 576 KB at any ring depth = 21 B/clk; four dedicated loaders: 11.2 us (the same four issuers); eight loaders: see the profile.
// A unit = SPU k32 stages; one s_barrier per unit joins all waves: the loaders arrive when the unit has landed (counted vmcnt), the
// compute waves when they are done with the unit before; behind it the loaders refill that unit's slot.  The loaders' unit sequence is
// FLAT over both phases (fc1 units, then fc2 units: fully unrolled, every vmcnt allowance an immediate), so fc2's first units are on
// their way while the compute waves are still in fc1 and its GELU epilogue.
template <int CH, int SPU, int NRB> struct SkRing {
    static constexpr int USZ = SPU * (32 * NRB + CH) * 128;
    static constexpr int FIT = (160 * 1024 - 32 * NRB * CH * 4) / USZ;
    static constexpr int N = FIT < 7 ? FIT : 7;
};

// NRB = 32-row blocks per workgroup (1 or 2).  With two, every streamed weight line serves 64 rows; the second row block is a second SET of
// compute waves (wave = (row block, column block): registers per wave as with one), not a second accumulator chain per wave.
template <int CH, int NGRP, int NQ1, int NL, int SPU, int NRB>
__global__ __launch_bounds__(64 * (NRB * CH / 32 + NL), 1) void mlp_sk_kernel(const lvae_mlp_sk_desc d) {
#pragma clang fp contract(off)
    constexpr int RT = 32 * NRB;
    constexpr int NWC = CH / 32, NW = NRB * NWC, KS = CH / 32, NBUF = SkRing<CH, SPU, NRB>::N;
    constexpr int ST1 = (RT + CH) * 128, ST2 = CH * 128;            // bytes of one k32 stage inside a slot: fc1 (32 y rows + CH W1 rows) / fc2 (CH W2 rows)
    constexpr int USZ = SkRing<CH, SPU, NRB>::USZ;                       // bytes of a ring slot
    constexpr int NG1 = RT / 8 + CH / 8, NG2 = CH / 8;                   // DMA pieces of one stage
    constexpr int P1 = SPU * NG1 / NL, P2 = SPU * NG2 / NL;         // ... per loader wave and unit
    constexpr int NU1 = NQ1 / SPU, NU2 = NGRP * KS / SPU, NUT = NU1 + NU2;
    static_assert(NBUF >= 2 && (SPU * NG1) % NL == 0 && (SPU * NG2) % NL == 0 && NQ1 % SPU == 0 && KS % SPU == 0, "whole pieces per loader wave, whole units");
    static_assert((NBUF - 2) * P1 <= 63, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const ring = (char*)smem;
    char* const hid_lds = ring + NBUF * USZ;                        // [KS][RT rows][128 B]: the GELU'd hidden tile in H2K32 stage form
    const int S1 = d.S1 > 1 ? d.S1 : 1, S2 = d.S2;
    const int t = blockIdx.x, c = t % S2, rt = t / S2;             // consecutive workgroups (one per XCD in turn) take consecutive slices
    const int m0 = rt * RT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = NQ1 * 32, hid = d.hid;
    const int rowb1 = C * 4, rowb2 = hid * 4;                       // bytes of an H2K32 row of y / W1, of W2

    if (wave >= NW) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wave - NW;
        const int rows_a = (d.M - m0) < RT ? (d.M - m0) : RT;
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.y + (long)m0 * rowb1), 0, rows_a * rowb1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.w1 + (long)c * CH * rowb1), 0, CH * rowb1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w2, 0, C * rowb2, 0x00020000);
        const int r_in = lane >> 3, pp = lane & 7;
        // lane -> (row within the piece's 8, physical 16-B chunk); logical chunk = physical ^ ((stage row >> 1) & 7), stage row = 8 g + r_in
        const int sw0 = (pp ^ ((r_in >> 1) & 7)) << 4, sw1 = (pp ^ ((4 + (r_in >> 1)) & 7)) << 4;      // piece index g even / odd
        auto issue = [&](int idx, int slot) {                       // idx, slot: compile-time after unrolling
            if (idx < NU1) {
#pragma unroll
                for (int i = 0; i < P1; ++i) {
                    const int q = i * NL + lw, st = q / NG1, g = q - st * NG1;      // uniform
                    const bool isA = g < RT / 8;                    // the stage's first RT rows are y's
                    const int soff = (isA ? 8 * g : 8 * g - RT) * rowb1 + (idx * SPU + st) * 128;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsY : rsW1, (__attribute__((address_space(3))) void*)(ring + slot * USZ + st * ST1 + g * 1024), 16,
                                                             r_in * rowb1 + ((g & 1) ? sw1 : sw0), soff, 0, 0);
                }
            } else {
                const int u = idx - NU1, grp = u / (KS / SPU), ks0 = (u % (KS / SPU)) * SPU;
#pragma unroll
                for (int i = 0; i < P2; ++i) {                      // rows grp * CH + 8 g ... of W2 (beyond C: out of range = zeros), k32 stage c * KS + ks
                    const int q = i * NL + lw, st = q / NG2, g = q - st * NG2;
                    const int soff = (grp * CH + 8 * g) * rowb2 + (c * KS + ks0 + st) * 128;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (__attribute__((address_space(3))) void*)(ring + slot * USZ + st * ST2 + g * 1024), 16,
                                                             r_in * rowb2 + ((g & 1) ? sw1 : sw0), soff, 0, 0);
                }
            }
        };
#pragma unroll
        for (int idx = 0; idx < NBUF - 1; ++idx) issue(idx, idx);
#pragma unroll
        for (int idx = 0; idx < NUT; ++idx) {
            // unit idx has landed when at most the pieces of the units behind it that are already issued (idx + 1 ... idx + NBUF - 2) are outstanding
            int allow = 0;
#pragma unroll
            for (int j = idx + 1; j <= idx + NBUF - 2 && j < NUT; ++j) allow += j < NU1 ? P1 : P2;
            switch (allow) {                                        // (an immediate: `allow` is a constant of the unrolled iteration)
#define SK_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory"); break;
                SK_W(0) SK_W(4) SK_W(5) SK_W(6) SK_W(7) SK_W(8) SK_W(9) SK_W(10) SK_W(12) SK_W(13) SK_W(14) SK_W(15) SK_W(16) SK_W(17) SK_W(18) SK_W(19)
                SK_W(20) SK_W(21) SK_W(22) SK_W(23) SK_W(24) SK_W(25)
#undef SK_W
                default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
            }
            if (idx + NBUF - 1 < NUT) issue(idx + NBUF - 1, (idx + NBUF - 1) % NBUF);
            if (idx == NU1 - 1) asm volatile("s_barrier" ::: "memory");      // the compute waves' "hidden tile complete" barrier
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves: wave = (row block rb, column block cb)
    const int rb = wave / NWC, cb = wave - rb * NWC;
    const int li = lane & 31, lh = lane >> 5, lj = li & 3;
    const int per1 = NQ1 / S1;
    // fragment addresses: chunk (plane p, k16 step tt, lane half lh) = 4 p + 2 tt + lh of the lane's row, at ((chunk ^ xr) << 4)
    const int xr = (li >> 1) & 7;
    unsigned fo[4];                                                 // [2 p + tt]
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) fo[pt] = (unsigned)(((4 * (pt >> 1) + 2 * (pt & 1) + lh) ^ xr) << 4);
    const unsigned ring_a = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)ring;
    const unsigned a_row = ring_a + (rb * 32 + li) * 128, b1_row = ring_a + (RT + cb * 32 + li) * 128, b2_row = ring_a + (cb * 32 + li) * 128;
    const unsigned h_row = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)hid_lds + (rb * 32 + li) * 128;

    // fc1: P[32 x 32 of this wave] = y W1c^T, S1 slices folded in order
    f32x16 accH, accX, tot;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accH[r] = 0.f; accX[r] = 0.f; tot[r] = 0.f; }
    int in_slice = 0, slice = 0, buf = 0;
    for (int un = 0; un < NU1; ++un) {
        asm volatile("s_barrier" ::: "memory");
        SK_FENCE();
#pragma unroll
        for (int st = 0; st < SPU; ++st) {
            f16x8 af[2][2], bf[2][2];                               // [tt][plane]
            const unsigned ua = a_row + buf * USZ + st * ST1, ub = b1_row + buf * USZ + st * ST1;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                SK_DSR(af[tt][0], ua + fo[0 + tt]); SK_DSR(af[tt][1], ua + fo[2 + tt]);
                SK_DSR(bf[tt][0], ub + fo[0 + tt]); SK_DSR(bf[tt][1], ub + fo[2 + tt]);
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                if (tt == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(bf[0][0]), "+v"(bf[0][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bf[1][0]), "+v"(bf[1][1]));
                SK_FENCE();
                accX = SK_MFMA(af[tt][1], bf[tt][0], accX);
                accX = SK_MFMA(af[tt][0], bf[tt][1], accX);
                accH = SK_MFMA(af[tt][0], bf[tt][0], accH);
                SK_FENCE();
            }
            if (++in_slice == per1) {                               // end of a K slice of fc1: the partial sum as the parallel form stores it
                in_slice = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pr = __builtin_fmaf(accX[r], 1.0f / 2048.0f, accH[r]);
                    if (S1 > 1) pr = pr + 0.0f;                     // (the slab store's "+ bias" with no bias: -0 -> +0)
                    tot[r] = slice == 0 ? pr : tot[r] + pr;
                    accH[r] = 0.f; accX[r] = 0.f;
                }
                ++slice;
            }
        }
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
    // + bias -> GELU -> f16x2 split -> the hidden tile in LDS: [KS][RT rows][128 B], stage ks = this wave's 32 columns; rows 4 lh + 8 g + lj
    // of its row block after the quad transpose
    {
        const f32x4 vb = *(const f32x4*)(d.b1 + c * CH + cb * 32 + (li & ~3));
        const int col32 = li & ~3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0 = tot[4 * g + 0], v1 = tot[4 * g + 1], v2 = tot[4 * g + 2], v3 = tot[4 * g + 3];
            quad_transpose(v0, v1, v2, v3, lj);
            v0 += vb[0]; v1 += vb[1]; v2 += vb[2]; v3 += vb[3];
            gelu_erf4(v0, v1, v2, v3);
            unsigned h0, l0, h1, l1;
            split_pair_h2(v0, v1, h0, l0);
            split_pair_h2(v2, v3, h1, l1);
            const int r = 4 * lh + 8 * g + lj;
            char* q = hid_lds + (cb * RT + rb * 32 + r) * 128 + ((col32 & 4) << 1);
            const int sw = (r >> 1) & 7;
            *(u32x2_t*)(q + ((((col32 >> 3)) ^ sw) << 4)) = (u32x2_t){h0, h1};
            *(u32x2_t*)(q + (((4 + (col32 >> 3)) ^ sw) << 4)) = (u32x2_t){l0, l1};
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // hidden tile complete (the loaders join this barrier too)
    SK_FENCE();

    // fc2: O[32 x C of this wave's row block] partial over k = the CH hidden columns of slice c, one group of CH output columns after the other
    // (a group's accumulators live only while its units run: its partial sums leave for plane c of the workspace -- fma(X, 2^-11, H) + 0.0f,
    // row-major [M][C], what gemm_h2_kernel's slice c stores (gemm_finish, SLAB) -- as soon as its last unit is done)
    float* const plane = d.ws + (long)c * d.M * C;
    f32x16 oH, oX;
#pragma unroll
    for (int u = 0; u < NU2; ++u) {
        const int grp = u / (KS / SPU), ks0 = (u % (KS / SPU)) * SPU, ubuf = (NU1 + u) % NBUF;
        if (ks0 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { oH[r] = 0.f; oX[r] = 0.f; }
        }
        asm volatile("s_barrier" ::: "memory");
        SK_FENCE();
#pragma unroll
        for (int st = 0; st < SPU; ++st) {
            f16x8 af[2][2], bf[2][2];
            const unsigned ua = h_row + (ks0 + st) * RT * 128, ub = b2_row + ubuf * USZ + st * ST2;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                SK_DSR(af[tt][0], ua + fo[0 + tt]); SK_DSR(af[tt][1], ua + fo[2 + tt]);
                SK_DSR(bf[tt][0], ub + fo[0 + tt]); SK_DSR(bf[tt][1], ub + fo[2 + tt]);
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                if (tt == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(bf[0][0]), "+v"(bf[0][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bf[1][0]), "+v"(bf[1][1]));
                SK_FENCE();
                oX = SK_MFMA(af[tt][1], bf[tt][0], oX);
                oX = SK_MFMA(af[tt][0], bf[tt][1], oX);
                oH = SK_MFMA(af[tt][0], bf[tt][0], oH);
                SK_FENCE();
            }
        }
        if (ks0 + SPU == KS) {                                      // the group's last unit
            const int n0 = grp * CH + cb * 32;
            if (n0 < C) {                                           // uniform: the last group of a C that is no multiple of CH
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0 = __builtin_fmaf(oX[4 * q + 0], 1.0f / 2048.0f, oH[4 * q + 0]) + 0.0f;
                    float v1 = __builtin_fmaf(oX[4 * q + 1], 1.0f / 2048.0f, oH[4 * q + 1]) + 0.0f;
                    float v2 = __builtin_fmaf(oX[4 * q + 2], 1.0f / 2048.0f, oH[4 * q + 2]) + 0.0f;
                    float v3 = __builtin_fmaf(oX[4 * q + 3], 1.0f / 2048.0f, oH[4 * q + 3]) + 0.0f;
                    quad_transpose(v0, v1, v2, v3, lj);
                    const int row = m0 + rb * 32 + 4 * lh + 8 * q + lj;
                    if (row < d.M) *(f32x4*)(plane + (long)row * C + n0 + (li & ~3)) = (f32x4){v0, v1, v2, v3};
                }
            }
        }
    }
}

template <int CH, int NGRP, int NQ1, int NL, int SPU, int NRB>
int launch_sk(const lvae_mlp_sk_desc* d, hipStream_t st) {
    constexpr int LDS = SkRing<CH, SPU, NRB>::N * SkRing<CH, SPU, NRB>::USZ + 32 * NRB * CH * 4;
    static LdsAttr attr;
    if (const int ae = attr.ensure((const void*)mlp_sk_kernel<CH, NGRP, NQ1, NL, SPU, NRB>, LDS)) return ae;
    const int row_tiles = (d->M + 32 * NRB - 1) / (32 * NRB);
    hipLaunchKernelGGL((mlp_sk_kernel<CH, NGRP, NQ1, NL, SPU, NRB>), dim3(row_tiles * d->S2), dim3(64 * (NRB * CH / 32 + NL)), LDS, st, *d);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int lvae_mlp_sk(const lvae_mlp_sk_desc* d, void* stream) {
    if (!d || !d->y || !d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->gamma || !d->res || !d->out || !d->ws || d->M <= 0) return -22;
    const int C = d->C, hid = d->hid, S1 = d->S1 > 1 ? d->S1 : 1, S2 = d->S2;
    if (S2 < 2 || (C & 31) || (hid & 31) || hid % S2 || (C / 32) % S1 || (long)d->M * C * 4 > 0x7fffffffL || (long)C * hid * 4 > 0x7fffffffL) return -22;
    const int CH = hid / S2, ngrp = (C + CH - 1) / CH;
    hipStream_t st = (hipStream_t)stream;
    int rc = -22;
    if (C != 512) return -22;
    // 64-row workgroups where 32-row ones would be more than one round of the chip's CUs (tuning hook of tools/r6_mlp_sk_bench.py: LVAE_SK_NRB)
#ifdef LVAE_EXPERIMENTAL_BUILD
    static const int force_nrb = getenv("LVAE_SK_NRB") ? atoi(getenv("LVAE_SK_NRB")) : 0;
#else
    constexpr int force_nrb = 0;                                    // (the product library's launch paths read no environment)
#endif
    const bool two = force_nrb ? force_nrb == 2 : ((d->M + 31) / 32) * S2 > lvae_cu_count();
    if (CH == 128 && ngrp == 4) rc = two ? launch_sk<128, 4, 16, 8, 1, 2>(d, st) : launch_sk<128, 4, 16, 8, 2, 1>(d, st);
    else if (CH == 192 && ngrp == 3) rc = launch_sk<192, 3, 16, 8, 2, 1>(d, st);      // (two row blocks would be 20 waves)
    else if (CH == 256 && ngrp == 2) rc = launch_sk<256, 2, 16, 4, 1, 1>(d, st);
    if (rc) return rc;
    // the second pass of the parallel split-K form, unchanged: out = res + gamma * (sum over the S2 planes in slice order + bias)
    lvae_gemm_desc g = {};
    g.M = d->M; g.N = C; g.K = hid;
    g.bias = d->b2; g.gamma = d->gamma; g.res = d->res; g.ldres = C; g.out = d->out; g.ldo = C;
    g.epi = LVAE_EPI_GAMMA_RES; g.store = LVAE_ST_ROWMAJOR; g.ws = d->ws; g.ksplit = S2;
    return lvae_splitk_reduce_launch(&g, S2, st);
}

// 1 when lvae_mlp_sk takes this shape (the host's rule, lvae/engine.py::mlp_sk_ok, asks before it records the launch)
extern "C" int lvae_mlp_sk_supported(int C, int hid, int S1, int S2) {
    if (C != 512 || S2 < 2 || S1 < 1 || (hid & 31) || hid % S2 || (C / 32) % S1) return 0;
    const int CH = hid / S2, ngrp = (C + CH - 1) / CH;
    return (CH == 128 && ngrp == 4) || (CH == 192 && ngrp == 3) || (CH == 256 && ngrp == 2);
}
