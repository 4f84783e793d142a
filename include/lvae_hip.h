/*
 * include/lvae_hip.h -- C ABI of liblvae_hip.so, the MI355X-native (gfx950) drop-in for the native code the
 * reference's QARV / QRes-VAE encode+decode hot path reaches.
 *
 * The reference (duanzhiihao/lossy-vae) is 100% Python; its hot path crosses into native code only through
 * third-party packages:
 *   - CompressAI pybind11 entry points  RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes /
 *     pmf_to_quantized_cdf, reached at  lvae/models/qarv/model.py:107,113,124  and
 *     lvae/models/qresvae/model.py:325,339,356  -> replaced by the `lvae_rans_*` / `lvae_pmf_*` host functions;
 *   - ATen/cuDNN/cuBLAS kernels behind  lvae/models/common.py:142-161 (ConvNeXtBlockAdaLN.forward),
 *     common.py:29-38 (patch_downsample / patch_upsample), qarv/model.py:36-39,44-75 (prior / posterior /
 *     z_proj convs, softplus/exp, build_indexes, quantize)  -> replaced by the `lvae_*_f32` device launchers.
 *
 * Conventions
 *   - Plain C: raw pointers + sizes; no torch types.  Device pointers are owned by the caller (in the Python
 *     host they are torch tensors' data_ptr()); all device launchers are asynchronous on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream) and return a hipError_t as int (0 = success),
 *     or a negative value for argument errors detected on the host.
 *   - Activations are NHWC fp32 ([B][H][W][C], C contiguous); a "row" m is one pixel (b,h,w).
 *   - Host coder functions are thread-safe and stateless (no globals); caller owns all buffers.
 *   - Entropy-coder streams are byte-compatible with the oracle restatement of CompressAI's rANS
 *     (oracle/rans_oracle.c): 64-bit rANS (ryg rans64), 16-bit precision, 4-bit bypass escapes.
 */
#ifndef LVAE_HIP_H
#define LVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------ meta */
int lvae_abi_version(void);          /* bumps on any signature change */
const char* lvae_build_info(void);   /* "gfx950 hipcc <ver> ..." */

/* ------------------------------------------------------------------------------------------------ host coder
 * Replaces compressai._CXX.pmf_to_quantized_cdf(list[float], int) -> list[int]
 * (called from GaussianConditional.update(): qarv/model.py:123-124, qresvae/model.py:317-325).
 * cdf_out has n+1 entries.  Returns 0, or -1 (negative / non-finite pmf), -2 (zero total), -3 (cannot fix up). */
int lvae_pmf_to_quantized_cdf(const float* pmf, int n, int precision, uint32_t* cdf_out);

/* Whole-table builder (GaussianConditional.update() semantics): for each of n_scales scales
 * c=ceil(scale*m), len=2c+1, pmf[k]=Phi((.5-|k-c|)/s)-Phi((-.5-|k-c|)/s) in fp32, tail=2*lower[0],
 * row=pmf_to_quantized_cdf([pmf, tail]).  cdf_form: 0 = erf form 0.5*(1+erf(x/sqrt2)) (DiscretizedGaussian,
 * lvae/models/entropy_coding.py:81-82), 1 = erfc form 0.5*erfc(-x/sqrt2) (stock GaussianConditional, qres34m).
 * `multiplier` = -ppf(tail_mass/2) (6.10941... for 1e-9).  qcdf is [n_scales][row_stride] int32, zero-filled
 * beyond each row's length; cdf_len[i]=len_i+2; offset[i]=-c_i.  Returns max row length (+2) or <0.
 * NOTE: uses the C library's erff/erfcf; the Python host's update() uses torch ops exactly like the reference
 * so that tables are bit-identical to it -- this entry point is for non-Python integrators. */
int lvae_build_gaussian_tables(const float* scale_table, int n_scales, double multiplier, int cdf_form,
                               int32_t* qcdf, int row_stride, int32_t* cdf_len, int32_t* offset);

/* Replaces RansEncoder().encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets) -> bytes
 * (qarv/model.py:107 via GaussianConditional.compress).  One call = one self-contained stream.
 * Returns bytes written (multiple of 4, >= 8) or <0: -2 = out_cap too small, -4 = bad index. */
long lvae_rans_encode_with_indexes(const int32_t* sym, const uint8_t* idx, size_t n,
                                   const int32_t* qcdf, int row_stride, const int32_t* cdf_len,
                                   const int32_t* offset, uint8_t* out, size_t out_cap);

/* Test hook of the encoder's division-free step: lvae_rans_encode_with_indexes computes x' = ((x / freq) << 16) + (x % freq) + start with
 * a per-symbol reciprocal (Alverson; the form of ryg's rans64.h -- the published coder divides).  This applies ONE encode step to the
 * state x both ways (division / reciprocal) and returns both next states; 0 if the renormalisation decisions agree too.  1 <= freq <= 65535. */
int lvae_rans_enc_step_selftest(uint64_t x, uint32_t start, uint32_t freq, uint64_t* by_division, uint64_t* by_reciprocal);

/* Replaces RansDecoder().decode_with_indexes(bytes, indexes, cdfs, cdf_sizes, offsets) -> list[int]
 * (qarv/model.py:113 via GaussianConditional.decompress).  Returns 0 or <0 (-1 malformed, -3 overrun). */
int lvae_rans_decode_with_indexes(const uint8_t* in, size_t in_len, const uint8_t* idx, size_t n,
                                  const int32_t* qcdf, int row_stride, const int32_t* cdf_len,
                                  const int32_t* offset, int32_t* sym_out);

/* Batched variants: n_streams independent streams coded by up to n_threads host threads (0 = hardware
 * concurrency).  Stream s uses sym[s]/idx[s]/n[s]; outputs land in out[s] (capacity out_cap[s]) and
 * out_len[s] receives the byte count (or the negative error).  Returns 0 if all streams succeeded. */
int lvae_rans_encode_batch(int n_streams, const int32_t* const* sym, const uint8_t* const* idx, const size_t* n,
                           const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                           uint8_t* const* out, const size_t* out_cap, long* out_len, int n_threads);
int lvae_rans_decode_batch(int n_streams, const uint8_t* const* in, const size_t* in_len,
                           const uint8_t* const* idx, const size_t* n,
                           const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                           int32_t* const* sym_out, int* status, int n_threads);

/* ------------------------------------------------------------------------------------------------ status word
 * One device int per launch plan, zeroed by the caller, OR-ed into by the kernels that are handed its address, copied to the host at
 * synchronisation points the codec already has (lvae_encode_blocks / lvae_decode_blocks do it per latent block).
 *   RANGE            an input pixel outside [0, 1] or NaN -- the reference's `assert 0 <= im.min() <= im.max() <= 1`
 *                    (qarv/model.py:219-220, qresvae/model.py:492), raised by the stem / range kernels;
 *   NONFINITE_PRIOR  a prior parameter (mean or log-scale, qarv/model.py:51-53) is NaN / inf        (lvae_prior_index_f32,
 *                    lvae_lossless_params_f32);
 *   NONFINITE_LATENT a posterior mean / quantised latent is NaN / inf or does not fit an int32      (lvae_quantize_f32);
 *   NONFINITE_IMAGE  a reconstruction value is NaN / inf before the final clamp                     (ST_IMAGE store, lvae_lossless_output_f32).
 * Why: the reference computes its 1x1 convs / MLPs in fp32 (common.py:154, qarv/model.py:36-39) and cannot overflow at 65504; the default
 * arithmetic here (prec 4, "f16x2") splits every operand into fp16 terms, so an activation >= 65520 becomes inf in its hi term.  An
 * inf / NaN never turns back into a finite value on the way (MFMA sums, GELU, residual adds, LayerNorm all propagate it), and every
 * tensor of the codec ends in one of the three sinks above -- the encoder's features in a posterior mean, the top-down state in the next
 * prior or in the image -- so checking the sinks catches every overflow before a byte string or an image is returned.  The Python host
 * raises lvae.NonFiniteError naming `model.set_gemm_precision('bf16x3')` (bf16 terms have fp32's exponent range). */
enum {
    LVAE_STATUS_RANGE = 1, LVAE_STATUS_NONFINITE_PRIOR = 2, LVAE_STATUS_NONFINITE_LATENT = 4, LVAE_STATUS_NONFINITE_IMAGE = 8
};

/* ------------------------------------------------------------------------------------------------ device kernels
 * GEMM family: out[m][n] = epilogue( sum_k A[m][k] * Wt[n][k] + bias[n] ), fp32 in / fp32 accumulate on
 * v_mfma_f32_32x32x2_f32.  Replaces timm Mlp fc1/fc2 (common.py:131-132,154), conv 1x1 (qarv/model.py:36,38,39;
 * common.py:33-38), conv k=s patch_downsample (common.py:29-30) and the 3x3 posterior head (qarv/model.py:37). */
enum {
    LVAE_A_PLAIN  = 0,  /* A row m = [A0[m*lda0 .. +K0) , A1[m*lda1 .. +K1)]   (A1 optional: fused torch.cat) */
    LVAE_A_PATCH2 = 1,  /* 2x2/stride-2 patches of an NHWC [B][2H][2W][Cin] map; K = 4*Cin, order (i,j,ci)    */
    LVAE_A_CONV3  = 2   /* 3x3/pad-1 taps of an NHWC [B][H][W][Cin] map;       K = 9*Cin, order (i,j,ci)      */
};
enum {
    LVAE_EPI_BIAS       = 0,  /* acc + bias                                                                   */
    LVAE_EPI_BIAS_GELU  = 1,  /* gelu_erf(acc + bias)                 (fc1, common.py:132 nn.GELU exact form) */
    LVAE_EPI_GAMMA_RES  = 2,  /* res + gamma[n]*(acc + bias)          (fc2 + layer-scale + shortcut, :157-160) */
    LVAE_EPI_RES        = 3   /* res + acc + bias                     (fuse_feature_and_z, qarv/model.py:72-75) */
};
enum {
    LVAE_ST_ROWMAJOR    = 0,  /* out[m*ldo + n]                                                               */
    LVAE_ST_SHUFFLE     = 2,  /* PixelShuffle(r) into NHWC [B][H*r][W*r][N/r^2]; columns pre-permuted to
                                 n' = (i*r+j)*Cout + c  (common.py:33-38)                                     */
    LVAE_ST_IMAGE       = 3   /* final layer: PixelShuffle(r) + clamp(-1,1)*0.5+0.5 into NCHW [B][N/r^2][H*r][W*r];
                                 columns in the reference order n = c*r^2 + i*r + j (qarv/model.py:224-232)  */
};
typedef struct {
    const float* A0; const float* A1;     /* A sources (A1 may be NULL) */
    long lda0, lda1;                      /* row strides in floats (PLAIN) */
    int  K0, K1;                          /* PLAIN: lengths of the two sources; PATCH2/CONV3: K0 = Cin */
    int  H, W;                            /* PATCH2/CONV3/SHUFFLE/IMAGE: spatial size of the row grid (rows = B*H*W) */
    const float* Wt; long ldw;            /* weights [N][K], K contiguous */
    const float* bias;                    /* [N] */
    const float* gamma;                   /* [N]  (EPI_GAMMA_RES) */
    const float* res; long ldres;         /* residual rows (EPI_GAMMA_RES / EPI_RES); may alias out */
    float* out; long ldo;
    int  M, N, K;                         /* K = total reduction length */
    int  a_mode, epi, store, r;           /* r = pixel-shuffle factor */
    int  a_gelu;                          /* 1: apply gelu_erf to every A element on load (VDBlock's c1(gelu(x)),
                                             lvae/models/qresvae/model.py:143-144) */
    int  prec;                            /* 0: fp32 MFMA (v_mfma_f32_32x32x2_f32, an exact fmaf chain);
                                             1: operands rounded to bf16 (RNE), fp32 accumulate on v_mfma_f32_32x32x16_bf16
                                                (BASELINE config 5, not the parity path);
                                             2: "bf16x3": each fp32 operand split exactly into hi+mid+lo bf16 terms, six
                                                cross-term bf16 MFMAs per step, fp32 accumulate -- fp32-class accuracy
                                                (relative error of a product <= 2^-24).  prec 1/2 need Wt16, K % 8 == 0 */
                                          /* 3: REDUCED precision (BASELINE config 5): activations stored as bf16, operands quantised to
                                                OCP MX-fp8 (e4m3 + one E8M0 scale per 32 k) on v_mfma_scale_f32_32x32x64_f8f6f4, fp32
                                                accumulate; Wt16 = [N][ldw] e4m3 bytes then [N][ldw/32] scale bytes, ldw = K rounded
                                                up to 64 (zero padded; lvae.models.base.pack_mxfp8).  Not a parity path */
                                          /* 4: "f16x2": each fp32 operand split into hi + lo' * 2^-11 fp16 terms (23 of fp32's 24
                                                significant bits), three cross-term fp16 MFMAs per step (hi*hi | hi*lo' + lo'*hi in a
                                                second fp32 accumulator) -- fp32-class accuracy at half the matrix-pipe time of prec 2;
                                                operands must stay below 65504 in magnitude.  PLAIN (incl. [A0 | A1]), CONV3 and PATCH2 A
                                                modes, K % 32 == 0 -- or K % 16 == 0 with N <= 96, one PLAIN / CONV3 source, no a_gelu
                                                (csrc/gemm_h2n.hip) -- ldw == K; Wt16 = [N][K/16][2][16] fp16 (lvae.models.base.pack_f16x2).
                                                Selected by the host per GEMM (csrc/gemm_h2.hip; cfg: 0 = the library chooses kernel and
                                                tile, 1 / 2 = gemm_h2_kernel with 64 / 128-wide tiles, 3 = gemm_h2n_kernel wherever it
                                                applies -- every choice gives the same bits) */
    const unsigned short* Wt16;           /* prec 1: weights as bf16 bit patterns, [N][K], row stride ldw;
                                             prec 2: three such planes hi | mid | lo, plane stride N*ldw elements,
                                             followed -- when K % 32 == 0 and ldw == K -- by the same values in
                                             k16-interleaved order [N][K/16][3][16] (lvae.models.base.pack_bf16x3) */
    int  cfg;                             /* tile configuration: 0 = library heuristic, k>0 = candidate k-1 of
                                             lvae_gemm_num_configs() (results are bit-identical for every choice;
                                             the Python host autotunes this per shape at plan-build time) */
    int  ksplit;                          /* split-K: S > 1 cuts K into S equal slices (K % (32*S) == 0) computed by S x tiles
                                             workgroups into `ws`, then reduced IN SLICE ORDER and passed through the epilogue by a
                                             second kernel -- deterministic; for the few-tile, long-K layers (stride 32/64 MLPs, 3x3
                                             heads).  Row-major store only, N % 4 == 0.  The host must choose S independently of
                                             the batch size (per-image rows), so that batched and single-image calls agree */
    float* ws;                            /* split-K workspace, S*M*N floats (unused when ksplit <= 1) */
    int  a_bf16, out_bf16;                /* prec 3 only: A (both sources) / out and res are bf16 bit patterns (2-byte elements;
                                             lda, ldo, ldres stay in ELEMENTS); 0 = fp32.  The final NCHW image is always fp32 */
    int* cnt;                             /* split-K arrival counters, one int per output tile (>= ceil(M/64)*ceil(N/32) entries covers
                                             every tile shape), ZERO before the first launch; each launch leaves them zero again.
                                             Non-NULL: the last slice workgroup of a tile to arrive reduces the S slabs in slice order
                                             inside the GEMM launch (no second kernel; same bits).  NULL: separate reduce launch */
    int* status;                          /* optional device status word (LVAE_STATUS_* bits, above): the final-image store (ST_IMAGE) ORs
                                             LVAE_STATUS_NONFINITE_IMAGE into it when a value is NaN / inf before the clamp (the clamp would
                                             hide it).  NULL = no check */
    int  a_h2, out_h2;                    /* prec 4 only, "pre-split" operands in the f16x2 plane format H2K32 = [rows][K/32][2][32] fp16
                                             (per 32 k: 32 hi terms, then 32 lo' terms; a row is K*4 bytes like its fp32 form;
                                             lvae.models.base.pack_f16x2_k32): a_h2 = 1: A0 is such a buffer (PLAIN, K1 = 0, lda0 = K,
                                             K % 32 == 0), written by a producer with out_h2 (lvae_dwconv_ln_h2, a GEMM) -- the GEMM then
                                             streams both operands global -> LDS by DMA with no conversion in its main loop
                                             (csrc/gemm_h2p.hip; Wt16 must be H2K32 as well).  out_h2 = 1: the result (ROWMAJOR,
                                             EPI_BIAS / EPI_BIAS_GELU, N % 32 == 0, ldo = N) is stored split in that format instead
                                             of fp32.  Same bits as splitting inside the consumer: the split is exact and unique.
                                             prec 3 (reduced-precision mode): the same two flags with the MX-fp8 operand format Q8 of an
                                             [R][K] matrix (K % 64 == 0): R*K e4m3 bytes row-major, then the E8M0 block scales as
                                             [K/64][R][2] bytes (lvae.models.base.pack_mxfp8_q8): a_h2 = 1: A0 (and Wt16) are Q8 buffers,
                                             written by lvae_dwconv_ln_q8 / a GEMM with out_h2 (csrc/gemm_q8.hip: no quantiser in the main
                                             loop); out_h2 = 1: the result (EPI_BIAS / EPI_BIAS_GELU, N % 64 == 0, ldo = N) is stored as Q8 */
    int  defer_reduce;                    /* ksplit > 1, parallel form only: 1 = the launch writes the S partial-sum planes to `ws` and does NOT
                                             run the reduce pass (no bias / epilogue is applied): a consumer that reads the planes itself
                                             finishes the job -- lvae_prior_index_sk_f32 sums them in slice order, adds the bias and goes on to
                                             the scale indexes: one launch less per latent block, the bits of the two-launch form (round 6) */
} lvae_gemm_desc;
int lvae_gemm_f32(const lvae_gemm_desc* d, void* stream);

/* The MLP of a ConvNeXt block as ONE launch (f16x2 arithmetic, pre-split operands; csrc/mlp_h2c.hip; common.py:131-132,154-158):
 *     out[m][c] = res[m][c] + gamma[c] * ( fc2( gelu_erf( fc1(y)[m] + b1 ) )[c] + b2[c] )
 * One persistent 512-thread workgroup per CU walks 128-row (C = 384: 64-row) tiles; per tile the hidden dimension is walked in chunks
 * that stay in the CU's LDS, both weight matrices are STREAMED by LDS-DMA (they do not fit a CU).  Instances: (C, hid) = (128, 192) -- the
 * decoder's stride-4 blocks (qarv/zoo.py:86-87; since round 5 the tile's A rows are resident in LDS: fetched once per tile, one tile ahead), (192, 384) -- the encoder's stride-4 blocks (qarv/zoo.py:38-40) and qres34m's,
 * (384, 768) -- the stride-8 blocks, taken by the host from M = 49152 rows on (lvae.engine.Plan.FUSED_MLP_MIN_ROWS).  -22 for any other
 * (C, hid) and for M * C * 4 >= 2^31 (32-bit row offsets: the host cuts larger maps into row ranges).
 * y: H2K32 planes [M][C] (lvae_dwconv_ln_h2); w1: H2K32 [hid][C]; w2: H2K32 [C][hid] (lvae.models.base.pack_f16x2_k32); res / out: fp32
 * [M][C] (may alias).  Every output bit equals the two lvae_gemm_f32 launches (prec 4, a_h2 / out_h2, ksplit = 1) it replaces. */
typedef struct {
    const void* y; const void* w1; const float* b1; const void* w2; const float* b2; const float* gamma;
    const float* res; float* out;
    int M, C, hid;
} lvae_mlp_desc;
int lvae_mlp_h2f(const lvae_mlp_desc* d, void* stream);

/* The MLP of a ConvNeXt block on the SMALL maps (stride 32 / 64: both GEMMs run split-K there), csrc/mlp_sk.hip (round 6):
 * out = res + gamma * (fc2(gelu(fc1(y) + b1)) + b2) with fc1's K = C in S1 slices and fc2's K = hid in S2 >= 2 slices, as TWO launches
 * instead of three or four -- workgroup (32 rows, slice c) computes the hid / S2 hidden columns of fc2's slice c itself (fc1's S1
 * slices folded in order), keeps them in LDS and writes fc2's partial sums to plane c of `ws` (S2 planes of M x C floats); the
 * split-K reduce launch then finishes.  Every output bit equals the lvae_gemm_f32 launches it replaces (prec 4, a_h2 / out_h2,
 * ksplit = S1 for fc1 and S2 for fc2: reference lvae/models/common.py:154-158 computes the same MLP in fp32).  y / w1 / w2 in H2K32
 * ([M][C], [hid][C], [C][hid]); res and out may alias.  Shapes: lvae_mlp_sk_supported(C, hid, S1, S2) != 0. */
typedef struct {
    const void* y; const void* w1; const float* b1; const void* w2; const float* b2; const float* gamma;
    const float* res; float* out; float* ws;
    int M, C, hid, S1, S2;
} lvae_mlp_sk_desc;
int lvae_mlp_sk(const lvae_mlp_sk_desc* d, void* stream);
int lvae_mlp_sk_supported(int C, int hid, int S1, int S2);
int lvae_gemm_num_configs(void);      /* number of selectable tile configurations */

/* Native replay of a recorded launch-plan segment (csrc/plan_runtime.cpp; lvae/engine.py: Plan.run): ONE foreign call instead of one
 * per launch.  Every entry names an entry point of this header (`kind`) and carries its arguments by class in call order: pointers in
 * p[], integers (int / long) in i[], floats in f[]; the stream argument comes from the call (`side` != 0: the side stream).
 * LVAE_OP_ORDER: p[0] = event, i[0] != 0: side stream waits for main (fork), else main waits for side (join).
 * Returns 0, or the first failing launch's code with its index in *failed_index (may be NULL). */
enum {
    LVAE_OP_GEMM = 1, LVAE_OP_DWCONV_LN_F32, LVAE_OP_DWCONV_LN_H2, LVAE_OP_DWCONV_LN_BF16, LVAE_OP_DWCONV_LN_Q8, LVAE_OP_STEM_F32, LVAE_OP_STEM_BF16,
    LVAE_OP_BIAS_EXPAND_F32, LVAE_OP_BIAS_EXPAND_BF16, LVAE_OP_PRIOR_INDEX, LVAE_OP_QUANTIZE, LVAE_OP_DEQUANTIZE, LVAE_OP_GAUSSIAN_NLL,
    LVAE_OP_LOSSLESS_PARAMS, LVAE_OP_LOSSLESS_OUTPUT, LVAE_OP_MLP_H2F, LVAE_OP_MLP_SK, LVAE_OP_PRIOR_INDEX_SK, LVAE_OP_QUANTIZE_SK, LVAE_OP_ORDER
};
typedef struct { int kind; int side; void* p[8]; long i[6]; double f[2]; } lvae_op;
#define LVAE_TRACE_MAGIC 1985229328.0      /* lvae_decode_blocks: seconds[1] of a timeline request */
int lvae_run_ops(const lvae_op* ops, int n, void* stream, void* side_stream, int* failed_index);

/* One pipeline group's DECODE as a single foreign call (replaces the per-latent-block Python loop around the reference's
 * `block.decompress` calls, qarv/model.py:531-557; qresvae/model.py:446-454): for every latent block, in order --
 *   launch its plan segment (up to its prior / index kernel) -> copy its scale indexes to pinned host memory -> wait for the stream ->
 *   rANS-decode the block's n_images streams (lvae_rans_decode_batch) into pinned host memory -> copy the symbols to the device --
 * then launch the tail segment (no wait: the caller synchronises).  `strings` / `string_len` are block-major: entry b * n_images + i is
 * image i's stream of block b.  Index / symbol buffers hold n_images * per_image entries per block, image after image.
 * `status_dev` / `status_host` (optional; status_host = one pinned int): the plan's status word (LVAE_STATUS_*) is copied ONCE, behind the
 * tail (the per-block chain is latency-bound and carries no extra copy: scale indexes are valid table rows whatever the prior parameters
 * were, so the coder cannot be hurt by them).  The CALLER reads *status_host after its own synchronisation and before it hands the
 * reconstruction on: non-zero = non-finite prior parameters / reconstruction -- an fp16 overflow of the default arithmetic, or a stream
 * written under another arithmetic.  A stream that fails to decode (-74) is reported as -75 (EOVERFLOW) when the word is set at that
 * point (garbage indexes, not a corrupt stream).  The caller zeroes the device word again.
 * Returns 0, a launch error (failed_block = block, failed_op = index in its segment; block n_blocks = the tail), -75 (above), or -74
 * (EBADMSG) when a stream is corrupt / truncated (failed_block = its block).  seconds[0] / [1] (optional: an array of TWO doubles or more,
 * output) receive the time spent waiting for the GPU segments and inside the coder.  Timeline (measurement; IN/out): with BOTH
 * seconds[0] = -(capacity of the array in doubles, 8 ... 4096) and seconds[1] = LVAE_TRACE_MAGIC on entry -- an uninitialised output array
 * never is -- absolute steady-clock stamps (s) follow the two totals, clamped to that capacity: per block b, seconds[2 + 4 b ..] = segment launch begins / segment + index copy issued /
 * indexes on the host / block decoded and its symbols on their way to the device; seconds[2 + 4 n_blocks] = tail issued. */
typedef struct {
    const lvae_op* ops; int n_ops;
    const uint8_t* idx_dev; uint8_t* idx_host;        /* n_images * per_image bytes.  idx_dev = NULL: the segment's prior-index launch was recorded
                                                         with idx_host (pinned, device-mapped host memory) as its output -- no copy is issued */
    int32_t* sym_host; int32_t* sym_dev;              /* n_images * per_image int32.  sym_dev = NULL: the NEXT segment's dequantize launch reads
                                                         sym_host itself (written by the coder before that segment is issued) -- no copy */
    size_t per_image;
} lvae_dec_block;
int lvae_decode_blocks(const lvae_dec_block* blocks, int n_blocks, int n_images, const uint8_t* const* strings, const size_t* string_len,
                       const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                       const lvae_op* tail_ops, int n_tail, const int* status_dev, int* status_host, void* stream, void* side_stream,
                       int n_threads, int* failed_block, int* failed_op, double* seconds);

/* One pipeline group's ENCODE as a single foreign call (the loop around `block.compress`, qarv/model.py:516-529): launch every block's
 * segment (through its quantize kernel), each followed by the copies of its symbols / scale indexes to pinned host memory and an event;
 * then, block by block, wait for its event and rANS-encode its n_images streams (lvae_rans_encode_batch) into out[b * n_images + i]
 * (capacity out_cap[b * n_images + i]), while the GPU computes the later blocks.  out_len[b * n_images + i] receives the byte count.
 * `status_dev` / `status_host` (optional; status_host = one pinned int): the plan's status word (LVAE_STATUS_*), copied ONCE behind the last
 * block's segment and checked before that block is coded: LVAE_STATUS_RANGE set returns -34 (ERANGE: the reference's input assert), any
 * other bit -75 (EOVERFLOW: non-finite prior parameters / posterior means -- an fp16 overflow of the default arithmetic), with
 * failed_block = n_blocks - 1; the strings of the earlier blocks are to be discarded (the coder accepts any int32 symbol, and every scale
 * index is a valid table row, so coding them was harmless).  The caller zeroes the device word again.  Events are created and destroyed
 * inside the call. */
typedef struct {
    const lvae_op* ops; int n_ops;
    const int32_t* sym_dev; int32_t* sym_host;        /* sym_dev / idx_dev = NULL: the segment's quantize / prior-index launches wrote the pinned */
    const uint8_t* idx_dev; uint8_t* idx_host;        /* host arrays themselves (as lvae_dec_block) -- no copies, the event follows the segment  */
    size_t per_image;
} lvae_enc_block;
int lvae_encode_blocks(const lvae_enc_block* blocks, int n_blocks, int n_images, uint8_t* const* out, const size_t* out_cap, long* out_len,
                       const int32_t* qcdf, int row_stride, const int32_t* cdf_len, const int32_t* offset,
                       const int* status_dev, int* status_host, void* stream, void* side_stream, int n_threads,
                       int* failed_block, int* failed_op, double* seconds);

/* Depthwise kxk conv (+bias) -> LayerNorm over C (eps 1e-6, biased variance, no affine) -> AdaLN
 * y*(1+scale)+shift, one pass over an NHWC map (common.py:145-152).  wt is [k*k][C] (tap-major), `ln_w`/`ln_b`
 * (optional, may be NULL) are the LayerNorm affine of qres34m's MyConvNeXtBlock (qresvae/model.py:168-182);
 * `shift`/`scale1p` (optional) are the per-lambda AdaLN vectors with scale1p = 1+scale.
 * Supported: k in {1,3,5,7}; C in {128,144,192,256,288,384,512}.  Returns -22 for unsupported shapes.
 * The bits of an output pixel depend on (C, k, which affines are given) only -- not on B, H, W or the launch geometry:
 * C in {128,192,256,384,512} with at most one affine runs csrc/dwconv_cl.hip, everything else the sliding-window kernel. */
int lvae_dwconv_ln_f32(const float* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                       const float* shift, const float* scale1p, float* y,
                       int B, int H, int W, int C, int k, void* stream);

/* The same operator with the result stored PRE-SPLIT for the f16x2 GEMM that consumes it (lvae_gemm_desc.a_h2): y is an H2K32 buffer
 * [B*H*W][C/32][2][32] fp16 (B*H*W*C*4 bytes, like the fp32 map); value = split of exactly the fp32 result lvae_dwconv_ln_f32 gives.
 * C in {128,192,256,384,512}, k in {1,3,5,7}, at most one affine (the csrc/dwconv_cl.hip instances); -22 otherwise. */
int lvae_dwconv_ln_h2(const float* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                      const float* shift, const float* scale1p, void* y,
                      int B, int H, int W, int C, int k, void* stream);

/* Reduced-precision mode: bf16 map in, result quantised to MX-fp8 (format Q8 of lvae_gemm_desc: [B*H*W][C] e4m3 bytes, then
 * [C/64][B*H*W][2] E8M0 scales) for the GEMM that consumes it (prec 3, a_h2 = 1).  Same shape rule as lvae_dwconv_ln_h2. */
int lvae_dwconv_ln_q8(const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                      const float* shift, const float* scale1p, void* y,
                      int B, int H, int W, int C, int k, void* stream);

/* bf16-storage forms of the reduced-precision mode (BASELINE config 5; prec 3 of lvae_gemm_f32): x / y / out are bf16 bit patterns
 * (NHWC, 2-byte elements), arithmetic in fp32 registers, parameters fp32.  Same semantics as the _f32 entry points otherwise. */
int lvae_dwconv_ln_bf16(const void* x, const float* wt, const float* bias, const float* ln_w, const float* ln_b,
                        const float* shift, const float* scale1p, void* y,
                        int B, int H, int W, int C, int k, void* stream);
int lvae_stem_bf16(const float* im, const float* wt, const float* bias, void* out,
                   int B, int H, int W, int Cout, float im_shift, float im_scale, int* range_flag, void* stream);
int lvae_bias_expand_bf16(const float* bias, void* out, long M, int C, void* stream);

/* Stem: NCHW image [B][3][H][W] in [0,1] -> (im+shift)*scale (qarv/model.py:221) -> conv 4x4/stride 4
 * (zoo.py:37) -> NHWC [B][H/4][W/4][Cout].  wt is [48][Cout] with k = (ci*4+i)*4+j. Cout <= 256, multiple of 64. */
int lvae_stem_f32(const float* im, const float* wt, const float* bias, float* out,
                  int B, int H, int W, int Cout, float im_shift, float im_scale, int* range_flag, void* stream);
/* range_flag (device int, may be NULL): bit 0 is OR-ed in when any input value lies outside [0, 1] or is NaN -- the reference's
 * `assert 0 <= im.min() <= im.max() <= 1` (qarv/model.py:219-220, qresvae/model.py:492) without its device sync: the caller zeroes
 * the flag once and reads it at a synchronisation point it already has.  lvae_range_flag_f32 is the same check as a stand-alone
 * pass over n floats (n % 4 == 0) for encoders whose first layer is not this stem (qres17m). */
int lvae_range_flag_f32(const float* x, long n, float lo, float hi, int* flag, void* stream);

/* y[n] = (gelu_out? gelu : id)( sum_k Wt[n][k] * (gelu_in? gelu(x[k]) : x[k]) + b[n] ) -- the lambda-embedding
 * MLP (qarv/model.py:206-210) and all AdaLN `embedding_layer`s (common.py:123-127) as one concatenated GEMV. */
int lvae_gemv_f32(const float* Wt, const float* b, const float* x, float* y, int N, int K,
                  int gelu_in, int gelu_out, void* stream);

/* Prior epilogue (qarv/model.py:51-53 + GaussianConditional.build_indexes, :106,112): prm is the `prior` conv
 * output [M][2z] (NHWC rows; first z = mean, last z = log-scale).  Writes pm [M][z] and, per image b, the scale
 * index of every latent element in the coder's NCHW raster order idx[b][c][h][w] (uint8 0..n_scales-1):
 *   pv = exp(softplus(x+2.3)-2.3); s = max(pv, bound); idx = #{i < n_scales-1 : table[i] < s}. */
int lvae_prior_index_f32(const float* prm, float* pm, uint8_t* idx, const float* scale_table, int n_scales,
                         float scale_bound, int B, int HW, int z, int* status, void* stream);
/* status (optional): LVAE_STATUS_NONFINITE_PRIOR is OR-ed in when a mean or log-scale parameter is NaN / inf (see "status word"). */
/* The same behind a split-K `prior` GEMM whose reduce pass was deferred (lvae_gemm_desc.defer_reduce): prm[m][c] = ((ws[0] + ws[1]) + ...
 * + ws[S-1])[m][c] + bias[c] -- planes of [B*HW][2z] floats, summed in slice order like splitk_reduce does -- is written (other consumers
 * read it: lvae_gaussian_nll_f32, lvae_prior_sample_f32) and indexed in ONE launch. */
int lvae_prior_index_sk_f32(const float* ws, int S, const float* bias, float* prm, float* pm, uint8_t* idx, const float* scale_table,
                            int n_scales, float scale_bound, int B, int HW, int z, int* status, void* stream);

/* GaussianConditional.quantize (qarv/model.py:107-108): sym = int32(rint_half_even(qm - pm)) in NCHW raster order
 * per image, zhat = float(sym) + pm in NHWC (row stride ldz, see below). */
int lvae_quantize_f32(const float* qm, const float* pm, int32_t* sym, float* zhat, int B, int HW, int z, int ldz,
                      int* status, void* stream);
/* The same behind a split-K `posterior` GEMM whose reduce pass was deferred: qm[m][c] = ((ws[0] + ws[1]) + ... + ws[S-1])[m][c] + bias[c]
 * (planes of [B*HW][z] floats, slice order) is written and quantised in ONE launch. */
int lvae_quantize_sk_f32(const float* ws, int S, const float* bias, float* qm, const float* pm, int32_t* sym, float* zhat, int B, int HW,
                         int z, int ldz, int* status, void* stream);
/* status (optional): LVAE_STATUS_NONFINITE_LATENT is OR-ed in when qm - pm is NaN / inf or |rint(qm - pm)| >= 2^31. */
/* GaussianConditional.dequantize (qarv/model.py:113): zhat = float(sym) + pm; sym in NCHW raster order.
 * In both, zhat rows have stride ldz >= z floats; columns [z, ldz) are written as zeros (lets a following 3x3 conv
 * consume a channel count rounded up to a multiple of 4: qres34m z = 14, 10). */
int lvae_dequantize_f32(const int32_t* sym, const float* pm, float* zhat, int B, int HW, int z, int ldz, void* stream);

/* Sampling branch of a latent block (qarv/model.py:98-100, mode='sampling' with latent=None; conditional_sample /
 * unconditional_sample :365-404): z = pm + pv * N(0,1) * t + U(-0.5, 0.5) * t, with pm / pv derived from the prior conv output
 * `prm` exactly as in lvae_prior_index_f32 (pv = exp(softplus(lv + 2.3) - 2.3), NOT lower-bounded).  Device RNG: Philox4x32-10
 * keyed by `seed`, one counter per element (`offset` + element index), Box-Muller for the normal -- reproducible for a given
 * (seed, offset) whatever the launch geometry; t = 0 gives z = pm exactly.  z rows have stride ldz >= zdim (pad written as 0). */
int lvae_prior_sample_f32(const float* prm, float* z, long M, int zdim, int ldz, float t, unsigned long long seed,
                          unsigned long long offset, void* stream);

/* GaussianNLLOutputNet coding parameters of qres34m_lossless (qresvae/model.py:69-94).  raw = the fused conv_mean | conv_scale
 * output after PixelShuffle, NHWC [B*H*W][6] (mean c0..2, log-scale c0..2), H x W = image size.  For every (b, c, y, x) in the
 * coder's NCHW raster order, with bin = 1/127.5 and the reference's fp32 operation order:
 *   pm  = (rint(m*127.5 + 127.5)/127.5 - 1) / bin          ("workaround to make sure lossless", :72)
 *   s   = exp(ls - ln(bin));  idx = #{i < n_scales-1 : table[i] < max(s, bound)}          (build_indexes)
 *   sym = rint(((im - 0.5)*2)/bin - pm)        only when im != NULL (encoder); im is (B,3,H,W) in [0,1]. */
int lvae_lossless_params_f32(const float* raw, const float* im, float* pm, uint8_t* idx, int32_t* sym, const float* table,
                             int n_scales, float bound, int B, int H, int W, int* status, void* stream);
/* status (optional): LVAE_STATUS_NONFINITE_PRIOR when a mean / log-scale parameter is NaN / inf. */

/* ... and its decoder side (:86-94 + process_output :496-504): out = clamp((sym + pm)*bin, -1, 1)*0.5 + 0.5, NCHW. */
int lvae_lossless_output_f32(const int32_t* sym, const float* pm, float* out, long n, int* status, void* stream);
/* status (optional): LVAE_STATUS_NONFINITE_IMAGE when sym + pm is NaN / inf (the clamp would hide it). */

/* Eval-mode rate estimate of one latent block (qarv/model.py:95-96 = CompressAI GaussianConditional.forward in eval mode):
 * out_nats[b] += sum over the block's elements of -ln max(P, 1e-9), P = Phi((.5-|sym|)/s) - Phi((-.5-|sym|)/s) in fp32 with
 * s from the prior conv output `prm` as in lvae_prior_index_f32; cdf_form 0 = erf (QARV), 1 = erfc (QRes).  sym is in the
 * coder's NCHW raster order; out_nats (double[B]) must be zeroed by the caller. */
int lvae_gaussian_nll_f32(const float* prm, const int32_t* sym, double* out_nats, float scale_bound, int B, int HW, int z,
                          int cdf_form, void* stream);

/* y = gelu_erf(x) elementwise: the exact-erf GELU used by every fused epilogue, exposed for numerics tests. */
int lvae_gelu_f32(const float* x, float* y, long n, void* stream);

/* Broadcast a [C] vector to all M rows (get_bias, qarv/model.py:289-292). */
int lvae_bias_expand_f32(const float* bias, float* out, long M, int C, void* stream);

/* sum over all elements of (a-b)^2 per image into out[b] (double), for PSNR (lvae/evaluation.py:47-49);
 * out must be zeroed by the caller. */
int lvae_sqerr_sum_f32(const float* a, const float* b, double* out, int B, long n_per_image, void* stream);
/* Deterministic form for ONE image pair of n floats: block i of n_partials writes partials[i] = its share of sum((a-b)^2) in fp64
 * (fixed element -> thread map, no atomics); the caller adds the partials in index order.  Run-to-run and process-to-process
 * identical, which the sharded evaluation needs to reproduce the single-process means bit for bit. */
int lvae_sqerr_partials_f32(const float* a, const float* b, double* partials, int n_partials, long n, void* stream);

/* Stream ordering for launch plans with independent branches (lvae/engine.py: Plan.fork / Plan.join): an event without timing, and
 * "work enqueued on to_stream from now on runs after the work enqueued on from_stream so far" (hipEventRecord + hipStreamWaitEvent). */
void* lvae_event_create(void);
int   lvae_event_destroy(void* event);
int   lvae_stream_order(void* from_stream, void* to_stream, void* event);

#ifdef __cplusplus
}
#endif
#endif /* LVAE_HIP_H */
