"""Launch plans: a model's encode/decode for one (batch, height, width) is recorded ONCE as a flat list of native
launches (function pointer + fully resolved arguments: device addresses of pre-allocated NHWC buffers and packed
weights), then replayed with a tight loop -- or as a HIP graph -- on the current HIP stream.

This is the MI355X-side replacement for the reference's eager `nn.Module.forward` dispatch (~450 leaf modules and
~10^3 elementwise kernels per encode: SURVEY.md 8(a) A15).  PyTorch is used only as the allocator / stream owner.
"""
import ctypes
import os

import torch

from . import _native
from ._native import GemmDesc


# tile configurations of lvae_gemm_f32 (index = cfg-1): (BM, BN) -- used only to prune autotune candidates
_GEMM_TILES = [(128, 128), (128, 64), (64, 64), (256, 256), (256, 192), (256, 224), (256, 128), (128, 256), (128, 192), (128, 32),
               (64, 64), (128, 64)]
_TUNE_CACHE = {}


def autotune_gemm(lib, d, stream_ptr, device):
    """Pick the fastest tile configuration for this GEMM shape by timing the candidates on the GPU (2 timed runs each
    after a warm-up).  Every configuration produces bit-identical results (fixed k-order), so this only affects speed."""
    key = (d.M, d.N, d.K, d.K0, d.K1, d.a_mode, d.epi, d.store)
    best = _TUNE_CACHE.get(key)
    if best is not None:
        return best
    cands = []
    for i, (bm, bn) in enumerate(_GEMM_TILES):
        if bn >= 2 * d.N and bn > 32:          # more than half of the tile's columns would be padding
            continue
        if d.N > 4 * bn and bn <= 64:          # narrow tiles on a wide problem
            continue
        if bm >= 4 * d.M and bm > 64:
            continue
        cands.append(i + 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream(device)
    best, best_t = 0, float('inf')
    for c in cands:
        d.cfg = c
        if lib.lvae_gemm_f32(ctypes.byref(d), stream_ptr) != 0:
            continue
        e0.record(st)
        for _ in range(2):
            lib.lvae_gemm_f32(ctypes.byref(d), stream_ptr)
        e1.record(st)
        e1.synchronize()
        t = e0.elapsed_time(e1)
        if t < best_t:
            best, best_t = c, t
    _TUNE_CACHE[key] = best
    return best


# Split-K policy constants.  They are part of the bitstream contract (the slice count fixes a GEMM's summation order, and encoder and
# decoder priors must agree bit for bit), so they are compile-time constants of the package, not environment knobs.
KSPLIT_MAX_TILES_PER_IMAGE = 96       # only the few-tile layers (stride-16..64 MLPs, 3x3 heads) are split
KSPLIT_TARGET_WORKGROUPS = 256        # slices x tiles-per-image stays within one workgroup per CU
INKERNEL_REDUCE_MAX_BYTES = 48 * 1024  # slabs of one tile (S x tile bytes) up to which the last-arriver reduction is used


def auto_ksplit(m1, N, K, store, ldo, ldres, prec):
    """Number of K slices for a GEMM whose PER-IMAGE row count is m1 (so the choice does not depend on the batch size: batched
    and single-image calls, and the encoder and decoder of one image size, use the same summation order).  Split-K is for the
    few-tile, long-K layers only (stride-16..64 MLPs, the 3x3 posterior heads; <= 96 tiles per image): there a workgroup's K loop, not the MFMA rate, sets
    the launch time (~1500 cycles per 16-deep stage whatever the tile)."""
    if store != _native.ST_ROWMAJOR or (N & 3) or (ldo & 3) or (ldres & 3) or K % 32:
        return 1
    nk = K // 32
    if prec == 1:
        if K % 64:
            return 1
        nk = K // 64
    tiles1 = ((m1 + 127) // 128) * ((N + 63) // 64)
    if tiles1 > KSPLIT_MAX_TILES_PER_IMAGE:
        return 1
    limit = min(nk // 4, KSPLIT_TARGET_WORKGROUPS // tiles1)   # >= 4 k-tiles (128 deep) per slice
    best = 1
    for s in range(2, max(2, limit) + 1):
        if s <= limit and nk % s == 0:
            best = s
    return best


def h2_eligible(d):
    """Does csrc/gemm_h2.hip / gemm_h2n.hip (prec 4, f16x2) take this GEMM?  Mirrors lvae_gemm_h2_try / lvae_gemm_h2n_try: a rule in the
    GEMM's shape only (never in M), so batched and single-image calls, encoder and decoder agree."""
    if d.ldw != d.K:
        return False
    if d.K % 32:
        # K = 16 (mod 32) -- the 3x3 convs over 48 channels of the qres bottleneck blocks: gemm_h2_kernel walks the k16 steps in pairs;
        # the narrow-output kernel (csrc/gemm_h2n.hip: N <= 96, plain rows or the 3x3 gather) takes any whole number of them
        if d.K % 16 or d.N > 96 or d.a_gelu or d.out_h2 or d.ksplit > 1:
            return False
        if d.a_mode == _native.A_PLAIN:
            return d.K1 == 0 and d.K0 == d.K and d.lda0 % 4 == 0 and d.M * d.lda0 * 4 <= 0x7ffffff0
        if d.a_mode == _native.A_CONV3:
            return d.K0 % 16 == 0 and d.K == 9 * d.K0 and d.K1 == 0 and d.H > 0 and d.W > 0 and d.M * d.K0 * 4 <= 0x7ffffff0
        return False
    if d.a_mode == _native.A_PLAIN:
        if d.lda0 % 4 or d.K0 + d.K1 != d.K:
            return False
        if d.K1 and (not d.A1 or d.K0 % 16 or d.lda1 % 4):
            return False
        return True
    if d.a_mode == _native.A_CONV3:
        return d.K0 % 16 == 0 and d.K == 9 * d.K0 and d.K1 == 0 and d.H > 0 and d.W > 0 and d.M * d.K0 * 4 <= 0x7ffffff0
    if d.a_mode == _native.A_PATCH2:
        return d.K0 % 8 == 0 and d.K == 4 * d.K0 and d.K1 == 0 and d.H > 0 and d.W > 0 and not d.a_gelu and d.M * 16 * d.K0 <= 0x7ffffff0
    return False


_ORDER = object()        # marker of a stream-ordering entry in Plan.ops


class NonFiniteError(ArithmeticError):
    """A NaN / inf reached one of the codec's sinks (prior parameters, posterior means / symbols, the reconstruction): nothing is
    returned.  The reference computes its 1x1 convs and MLPs in fp32 (common.py:154, qarv/model.py:36-39); under the default 'f16x2'
    arithmetic of this package an activation of 65520 or more overflows fp16 (csrc/gemm_h2.hip), which is what this error usually
    means -- `model.set_gemm_precision('bf16x3')` (same accuracy class, fp32's exponent range) on BOTH the encoding and the decoding
    side avoids it.  On decode it also fires for a stream written under another arithmetic or by other weights."""

    def __init__(self, word, prec=None, where=''):
        names = [n for bit, n in ((_native.STATUS_NONFINITE_PRIOR, 'prior parameters'), (_native.STATUS_NONFINITE_LATENT, 'posterior mean / symbols'),
                                  (_native.STATUS_NONFINITE_IMAGE, 'reconstruction')) if word & bit]
        self.word = word
        hint = ("under the default 'f16x2' GEMM arithmetic this is an activation beyond fp16's range (>= 65520): call "
                "model.set_gemm_precision('bf16x3') on the encoding AND the decoding side" if prec in (None, 'f16x2') else
                f"GEMM arithmetic {prec!r}: the weights / inputs produce values outside the arithmetic's range")
        super().__init__(f"non-finite values (NaN / inf) in {', '.join(names) or 'the codec'}{' ' + where if where else ''}; nothing was "
                         f"returned.  {hint} (a stream decoded under another arithmetic or with other weights than it was written with "
                         f"fails the same way)")


class _EventHolder:
    def __init__(self, lib, ev):
        self.lib, self.ev = lib, ev

    def __del__(self):
        try:
            self.lib.lvae_event_destroy(self.ev)
        except Exception:
            pass


class Plan:
    autotune = os.environ.get('LVAE_AUTOTUNE', '0') == '1'    # opt-in: in-situ gains were within noise (docs/MEASUREMENT_HISTORY.md 5)

    def __init__(self, device):
        self.lib = _native.lib()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('lvae hot path runs on an AMD GPU (torch device "cuda:N" under ROCm); there is no CPU '
                               f'fallback (got device {self.device})')
        self.ops = []          # [(fn, args, label)]
        self.keep = []         # tensors / descs that must outlive the plan
        self.bufs = {}
        self.flops = 0
        self.prec = 0          # lvae_gemm_desc.prec for this plan's GEMMs (0 fp32, 1 bf16, 2 bf16x3, 3 MX-fp8 + bf16 storage)
        self.adt = torch.float32   # storage type of the feature maps (bfloat16 in the reduced-precision mode)
        self.w16 = None        # reduced-precision mode: {fp32 weight address: bf16 copy address} (set by the model's plan)
        self.w16_x3 = None     # f16x2 plans: the bf16x3 map for the GEMMs the f16x2 kernel does not take
        self.w16_k32 = None    # f16x2 plans: weights in the H2K32 plane format for the pre-split-operand GEMMs (csrc/gemm_h2p.hip)
        self.w16_q8 = None     # fp8 plans: weights in the Q8 format of the pre-quantised-operand GEMMs (csrc/gemm_q8.hip)
        self.graphs = {}       # (lo, hi) -> torch.cuda.CUDAGraph (a hipGraph of that launch range), captured on 2nd use
        self.segments = {}     # (lo, hi, n_ops) -> (lvae_op array, n): the native form of that launch range
        self.seen = set()
        # Independent branches on a side stream (small maps only: there the GPU is far from full and the launches of a branch are
        # pure latency): ops recorded between side_begin() / side_end() go to `side_stream`, ordered against the main stream by
        # fork (side waits for main) / join (main waits for side) events.  Results do not change -- same kernels, same inputs.
        self.side_stream = None
        self.on_side = False

    # ---- memory
    def buf(self, name, numel, dtype=torch.float32):
        t = self.bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            if t is not None:
                self.keep.append(t)     # launches already recorded keep pointing at the old (smaller) scratch
            t = torch.empty(int(numel), dtype=dtype, device=self.device)
            self.bufs[name] = t
        return t

    def new(self, numel, dtype=torch.float32):
        t = torch.empty(int(numel), dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    # ---- the plan's status word (include/lvae_hip.h "status word"): one device int the stem kernel (input range, the reference's
    # `assert 0 <= im.min() <= im.max() <= 1`, qarv/model.py:219-220, qresvae/model.py:492) and the codec's sinks -- prior parameters,
    # posterior means / symbols, the final image store -- OR their LVAE_STATUS_* bits into.  It is zero unless something went wrong, so
    # it is zeroed when allocated and after a raise only, never per call.
    def status_ptr(self):
        """Address of the plan's device status word (allocated on first use, with its pinned host mirror)."""
        if getattr(self, 'status', None) is None:
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        return self.status.data_ptr()

    def fetch_status(self):
        """Queue the 4-byte D2H copy of the status word on the current stream (before a sync / event the caller has anyway)."""
        if getattr(self, 'status', None) is not None:
            self.status_host.copy_(self.status, non_blocking=True)

    def raise_if_flagged(self, word=None, where=''):
        """After a synchronisation that covers fetch_status() (or with the word a native group loop returned): raise like the
        reference's preprocess_input assert for an out-of-range input, NonFiniteError for a NaN / inf that reached a sink."""
        if getattr(self, 'status', None) is None:
            return
        w = int(self.status_host[0]) if word is None else int(word)
        if w == 0:
            return
        self.status.zero_()
        self.status_host.zero_()
        if w & _native.STATUS_RANGE:
            raise AssertionError('input image values must lie in [0, 1] (reference: preprocess_input, '
                                 '`0 <= im.min() <= im.max() <= 1`)')
        raise NonFiniteError(w, getattr(self, 'prec_name', None), where)

    # ---- recording
    def add(self, fn, args, label=''):
        self.ops.append((fn, tuple(args), label, self.on_side))

    def enable_side_stream(self):
        if self.side_stream is None and not self.use_graphs:
            self.side_stream = torch.cuda.Stream(device=self.device)
        return self.side_stream is not None

    def _order(self, side_waits_for_main, label):
        ev = self.lib.lvae_event_create()
        if not ev:
            raise RuntimeError('hipEventCreate failed')
        self.keep.append(_EventHolder(self.lib, ev))
        self.ops.append((_ORDER, (bool(side_waits_for_main), ev), label, False))

    def fork(self, label='fork'):
        """Side-stream work recorded after this point starts after the main-stream work recorded before it."""
        if self.side_stream is not None:
            self._order(True, label)

    def join(self, label='join'):
        """Main-stream work recorded after this point starts after the side-stream work recorded before it."""
        if self.side_stream is not None:
            self._order(False, label)

    def side_begin(self):
        self.on_side = self.side_stream is not None

    def side_end(self):
        self.on_side = False

    def sname(self, name):
        """Scratch buffers of side-stream ops are separate from the main stream's (they run concurrently)."""
        return name + '_side' if self.on_side else name

    DEFER_MAX_PLANES = 8               # ... up to this many planes (in situ, B = 1: 4 planes -5 us per block, 9 planes +7 us, 36 planes +30 us against the reduce launch)
    DEFER_HEAD_REDUCE = True           # prior / posterior heads leave their split-K planes to the index / quantize launch (False: reduce launch; A/B tool, same bits)
    use_q8_pipeline = True             # reduced-precision plans: producer-side quantisation (False: the in-GEMM quantiser everywhere; A/B tool)
    H2P_MIN_ROWS_PER_IMAGE = 1536      # stride-4 / 8 / 16 maps of a 512x768 image; below, the few-tile layers want split-K (gemm_h2.hip)

    def mlp_h2p_ok(self, C, hid, k, n_affine=1, rows_per_image=None):
        """f16x2 plans: can the MLP of a ConvNeXt block (dwconv+LN -> fc1 -> GELU -> fc2) run with pre-split operands -- the depthwise
        kernel and fc1's epilogue storing hi / lo' planes (H2K32) that fc1 / fc2 stream by LDS-DMA (csrc/gemm_h2p.hip)?  A rule in the
        block's shape and the rows of ONE image (never the batch size): the arithmetic -- hence every bit -- is the same either way
        (tests/test_gpu_f16x2.py), the rule only picks the faster pipeline."""
        return (self.prec == 4 and self.w16_k32 is not None and C in (128, 192, 256, 384, 512) and k in (1, 3, 5, 7) and n_affine <= 1
                and C % 32 == 0 and hid % 32 == 0 and (rows_per_image is None or rows_per_image >= self.H2P_MIN_ROWS_PER_IMAGE))

    # Serial split-K (gemm_h2p FOLD) against parallel split-K (gemm_h2 + reduce launch) on the few-tile layers: launch-time models fitted
    # to tools/microbench.py gemmsk on MI355X (profiles/r03_gemm_serial_splitk.txt).  Both forms give the same bits, so -- unlike the slice
    # count -- this choice may depend on the batch.
    SERIAL_US_BASE, SERIAL_US_PER_STAGE = 7.0, 0.36            # + per 32-deep stage, times the rounds of 256 tiles of 128 x 64
    PARALLEL_US_BASE, PARALLEL_US_PER_MELEM = 13.3, 1.83       # + per 1e6 workspace elements (S x M x N, written and re-read)

    def serial_splitk_pays(self, M, N, K, S):
        tiles = ((M + 127) // 128) * ((N + 63) // 64)
        serial = self.SERIAL_US_BASE + self.SERIAL_US_PER_STAGE * (K // 32) * max(1.0, tiles / 256.0)
        parallel = self.PARALLEL_US_BASE + self.PARALLEL_US_PER_MELEM * (S * M * N / 1e6) if S > 1 else serial + 1.0
        return serial < parallel

    def mlp_pipeline(self, C, hid, k, rows_per_image, n_affine=1):
        """f16x2 plans: how the MLP of a ConvNeXt block runs.  -> (pre1, pre2, S1, S2): fc1 / fc2 take their A operand pre-split (H2K32
        planes written by the depthwise kernel / fc1's epilogue) and run as gemm_h2p with S1 / S2 K slices (1 = no split-K; > 1 = the
        serial form).  Maps of >= 1536 rows per image: always (mlp_h2p_ok).  Smaller maps are the split-K layers: the slice counts are the
        per-image rule's (auto_ksplit -- the summation order, i.e. the bits), and the pre-split serial form is taken where the BATCH
        makes it the faster one (serial_splitk_pays); it produces the bits of the parallel form (tests/test_gpu_f16x2.py::
        test_gemm_h2p_serial_split_k_equals_parallel_split_k), so batched and single-image plans still agree."""
        if self.mlp_h2p_ok(C, hid, k, n_affine, rows_per_image):
            return True, True, 1, 1
        if not self.mlp_h2p_ok(C, hid, k, n_affine, None):
            return False, False, None, None
        M = self.B * rows_per_image
        S1 = auto_ksplit(rows_per_image, hid, C, _native.ST_ROWMAJOR, hid, 0, 4)
        S2 = auto_ksplit(rows_per_image, C, hid, _native.ST_ROWMAJOR, C, C, 4)
        pre1 = self.serial_splitk_pays(M, hid, C, S1)
        pre2 = pre1 and self.serial_splitk_pays(M, C, hid, S2)
        return pre1, pre2, (S1 if pre1 else None), (S2 if pre2 else None)

    def mlp_q8_ok(self, C, hid, k, n_affine=1):
        """Reduced-precision plans (prec 3): can the MLP of a ConvNeXt block run with its operands quantised by their PRODUCERS (the
        depthwise kernel and fc1's epilogue store MX-fp8 + block scales, csrc/gemm_q8.hip streams them by LDS-DMA)?  A rule in the
        block's shape only: in this mode the two pipelines differ in arithmetic (the producer quantises the fp32 value, the in-GEMM
        quantiser its bf16 rounding), so encoder and decoder must make the same choice for the same block."""
        return (self.prec == 3 and self.use_q8_pipeline and self.w16_q8 is not None and C in (128, 192, 256, 384, 512) and k in (1, 3, 5, 7) and n_affine <= 1
                and C % 64 == 0 and hid % 64 == 0)

    def gemm(self, *, A0, K0, M, N, Wt, bias, out, lda0=None, A1=None, K1=0, lda1=0, ldw=None, ldo=None,
             gamma=None, res=None, ldres=0, a_mode=_native.A_PLAIN, epi=_native.EPI_BIAS, store=_native.ST_ROWMAJOR,
             r=0, H=0, W=0, K=None, a_gelu=0, Wt16=None, exact=False, ksplit=None, a_bf16=None, out_bf16=None, a_h2=False,
             out_h2=False, defer_reduce=False, label='gemm'):
        """Record one lvae_gemm_f32 launch.  defer_reduce=True (a GEMM with a plain bias epilogue whose consumer can sum split-K planes
        itself: the prior head in front of lvae_prior_index_sk_f32): when the launch runs parallel split-K, its reduce pass is left to the
        consumer and (workspace address, S) is returned; otherwise -- no split-K, or a form that has no planes -- None, and `out` is final."""
        if K is None:
            K = K0 + K1
        if a_h2:               # both operands pre-converted (f16x2: H2K32 planes; fp8 mode: Q8): mlp_h2p_ok() / mlp_q8_ok() said so
            assert self.prec in (3, 4) and Wt16 is None
            Wt16 = (self.w16_k32 if self.prec == 4 else self.w16_q8).get(Wt)
            assert Wt16, f'{label}: weights do not fit the pre-converted operand format'
        if Wt16 is None and self.w16 is not None:
            Wt16 = self.w16.get(Wt)
        d = GemmDesc()
        d.A0, d.A1 = A0, A1
        d.lda0, d.lda1 = (lda0 if lda0 is not None else K0), lda1
        d.K0, d.K1, d.H, d.W = K0, K1, H, W
        d.Wt, d.ldw = Wt, (ldw if ldw is not None else K)
        d.bias, d.gamma, d.res, d.ldres = bias, gamma, res, ldres
        d.out, d.ldo = out, (ldo if ldo is not None else N)
        d.M, d.N, d.K = M, N, K
        d.a_mode, d.epi, d.store, d.r = a_mode, epi, store, r
        d.status = self.status_ptr() if store == _native.ST_IMAGE else None      # a NaN / inf would pass the final clamp unseen
        d.cfg = 0
        d.a_gelu = a_gelu
        # bf16 / bf16x3 only when the plan provides the bf16 planes; exact=True forces the fp32 MFMA (pure data-movement GEMMs with
        # 0/1 weights: nearest upsampling, space-to-depth -- x*1 + 0*... must reproduce x bit for bit)
        d.prec = (self.prec or 1) if (Wt16 and K % 8 == 0 and not exact) else 0
        d.a_h2, d.out_h2 = int(bool(a_h2)), int(bool(out_h2))
        if a_h2:
            assert a_mode == _native.A_PLAIN and K1 == 0 and K % (32 if self.prec == 4 else 64) == 0 and d.lda0 == K and d.ldw == K, label
            ksplit = (ksplit or 1) if self.prec == 4 else 1        # f16x2: > 1 = serial split-K inside gemm_h2p (mlp_pipeline)
            assert (K // 32) % ksplit == 0, label
        if out_h2:
            assert self.prec in (3, 4) and store == _native.ST_ROWMAJOR and N % (32 if self.prec == 4 else 64) == 0 and d.ldo == N, label
            if not a_h2:
                ksplit = 1
        if self.prec == 4 and not exact and not a_h2 and not (d.prec == 4 and h2_eligible(d)):
            # f16x2 plans: what csrc/gemm_h2.hip / gemm_h2n.hip do not take (K % 16 != 0; K = 16 mod 32 with a wide output; a weight beyond fp16's range) runs on
            # the bf16x3 arithmetic -- decided by the GEMM's shape and weights only, so encoder and decoder, batched and single-image
            # calls agree
            Wt16 = self.w16_x3.get(Wt) if self.w16_x3 is not None else None
            d.prec = 2 if (Wt16 and K % 8 == 0) else 0
        d.Wt16 = Wt16 if d.prec else None
        if self.prec == 3:
            # reduced-precision plans (BASELINE config 5): bf16 maps in HBM, MX-fp8 operands; weight rows are padded to 64 k
            assert d.prec == 3, f'{label}: no MX-fp8 form of this GEMM (K={K})'
            d.ldw = K if a_h2 else (K + 63) // 64 * 64
            d.a_bf16 = 1 if a_bf16 is None else int(a_bf16)
            d.out_bf16 = (0 if store == _native.ST_IMAGE else 1) if out_bf16 is None else int(out_bf16)
            ksplit = 1
        if ksplit is None:
            ksplit = auto_ksplit(M // max(1, getattr(self, 'B', 1)), N, K, store, d.ldo, ldres, d.prec)
        deferred = None
        if ksplit > 1 and a_h2:
            d.ksplit = ksplit                                      # serial form: no workspace, no reduce launch
        elif (1 < ksplit <= self.DEFER_MAX_PLANES and defer_reduce and epi == _native.EPI_BIAS and store == _native.ST_ROWMAJOR
              and d.ldo == N and bias):
            # (a workspace of its own name: the consumer launch follows at once, on the same stream)
            d.ksplit, d.ws, d.defer_reduce = ksplit, self.buf(self.sname('deferred_ws'), ksplit * M * N).data_ptr(), 1
            deferred = (d.ws, ksplit)
        elif ksplit > 1:
            d.ksplit, d.ws = ksplit, self.buf(self.sname('splitk_ws'), ksplit * M * N).data_ptr()
            # In-kernel slice reduction (lvae_gemm_desc.cnt: a tile's last-arriving slice workgroup sums the S slabs in place of the
            # second launch) is taken only when the slabs of one tile are small: the last arriver reads S x tile bytes ALONE at the
            # cross-XCD rate (~65 GB/s per workgroup), so with the 128 x 128..192 tiles of the MLP layers (64-98 KB x S) it costs more
            # than the reduce launch it removes (measured on MI355X: 105 -> 100 Mpixels/s at B = 8, 12.1 -> 13.3 ms at B = 1).
            if not deferred and ksplit * 128 * min(N, 192) * 4 <= INKERNEL_REDUCE_MAX_BYTES:
                n_cnt = ((M + 63) // 64) * ((N + 31) // 32)
                cname = self.sname('splitk_cnt')
                cnt = self.bufs.get(cname)
                if cnt is None or cnt.numel() < n_cnt:
                    if cnt is not None:
                        self.keep.append(cnt)
                    cnt = self.bufs[cname] = torch.zeros(max(4096, n_cnt), dtype=torch.int32, device=self.device)
                d.cnt = cnt.data_ptr()
        if self.autotune and M * N >= 64 * 64:
            sp = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            d.cfg = autotune_gemm(self.lib, d, sp, self.device)
        self.keep.append(d)
        self.flops += 2 * M * N * K
        self.add(self.lib.lvae_gemm_f32, (ctypes.byref(d),), label)
        return deferred

    # (C, hidden) block shapes whose MLP runs as ONE launch (csrc/mlp_h2c.hip: hidden dimension walked in chunks, weights streamed): the
    # decoder's and the encoder's stride-4 blocks
    FUSED_MLP_SHAPES = ((128, 192), (192, 384), (384, 768))
    # (384, 768) -- 64-row tiles: profiles/r04_mlp_h2c_384x768.txt -- pays from three tiles per CU on (-9 ... -13 %) and loses at 1.5 tiles
    # per CU (M = 24576: one pipeline group of four 512x768 images), where it is not taken
    FUSED_MLP_MIN_ROWS = {(384, 768): 49152}

    def mlp_fused_ok(self, C, hid, k, n_affine=1, M=None, rows_per_image=None):
        """f16x2 plans: does the block's MLP run as ONE launch (csrc/mlp_h2c.hip)?  Its bits are those of the two-launch path WITHOUT
        split-K (tests/test_gpu_f16x2.py::test_mlp_h2f_equals_two_gemms, test_mlp_h2c_equals_two_gemms).  Shapes without a row threshold
        are fused at every size -- a rule in the block's shape alone.  A shape WITH a threshold (FUSED_MLP_MIN_ROWS) looks at the
        launch's row count M = B * rows_per_image, so batched and single-image plans of one image size may differ in their choice:
        that is only sound where the two-launch alternative is the S = 1 pipeline too, i.e. on maps of >= H2P_MIN_ROWS_PER_IMAGE rows per
        image (mlp_h2p_ok with rows_per_image) -- on smaller maps the alternative is split-K with another summation order, and the
        fused form is not taken whatever the batch (qres34m's width-384 blocks on 256x256 images in groups of >= 48).
        (Launches beyond the kernel's 32-bit row offsets, M * C * 4 >= 2^31, are cut into row ranges by mlp_fused -- rows are independent.)"""
        if (C, hid) not in self.FUSED_MLP_SHAPES or not self.mlp_h2p_ok(C, hid, k, n_affine, None):
            return False
        min_rows = self.FUSED_MLP_MIN_ROWS.get((C, hid), 0)
        if min_rows == 0:
            return True
        if rows_per_image is None or not self.mlp_h2p_ok(C, hid, k, n_affine, rows_per_image):
            return False
        return (M or 0) >= min_rows

    def mlp_fused(self, *, y, M, C, hid, w1, b1, w2, b2, gamma, res, out, label='mlp'):
        """out = res + gamma * (fc2(gelu(fc1(y) + b1)) + b2) with y pre-split (lvae_dwconv_ln_h2): lvae_mlp_h2f.  The kernel addresses a
        launch's rows with 32-bit byte offsets: a map of 2 GiB or more goes out as several launches over row ranges (the MLP's rows are
        independent, so the bits do not change -- ADVICE r04)."""
        assert self.prec == 4 and (C, hid) in self.FUSED_MLP_SHAPES
        w1h, w2h = self.w16_k32.get(w1), self.w16_k32.get(w2)
        assert w1h and w2h, f'{label}: weights do not fit the pre-split operand format'
        max_rows = ((2 ** 31 - 1) // (C * 4)) // 128 * 128
        for r0 in range(0, M, max_rows):
            rows = min(max_rows, M - r0)
            d = _native.MlpDesc()
            o = r0 * C * 4                                   # y (H2K32 planes), res and out all have C * 4 bytes per row
            d.y, d.w1, d.b1, d.w2, d.b2, d.gamma, d.res, d.out = y + o, w1h, b1, w2h, b2, gamma, res + o, out + o
            d.M, d.C, d.hid = rows, C, hid
            self.keep.append(d)
            self.add(self.lib.lvae_mlp_h2f, (ctypes.byref(d),), label if M <= max_rows else f'{label}[{r0}:]')
        self.flops += 4 * M * C * hid

    # The MLP of a small-map block (both GEMMs split-K) as fused launch + reduce (csrc/mlp_sk.hip, round 6): same bits as the split-K
    # launches whichever form runs them, so the rule may look at the batch.  Taken up to this many rows per launch (beyond, the
    # pre-split serial split-K launches fill the chip and stream each weight once per 128 rows instead of once per 32).
    MLP_SK_MAX_ROWS = int(os.environ.get('LVAE_MLP_SK_MAX_ROWS', '1024'))

    def mlp_sk_ok(self, C, hid, k, rows_per_image, M, n_affine=1):
        """f16x2 plans: (S1, S2) when the block's MLP runs as lvae_mlp_sk, else None.  Only where the two-launch alternative is split-K
        with S2 >= 2 (maps below H2P_MIN_ROWS_PER_IMAGE rows per image); the slice counts are the per-image rule's (auto_ksplit)."""
        if self.MLP_SK_MAX_ROWS <= 0 or M > self.MLP_SK_MAX_ROWS:
            return None
        if not self.mlp_h2p_ok(C, hid, k, n_affine, None) or self.mlp_h2p_ok(C, hid, k, n_affine, rows_per_image):
            return None
        S1 = auto_ksplit(rows_per_image, hid, C, _native.ST_ROWMAJOR, hid, 0, 4)
        S2 = auto_ksplit(rows_per_image, C, hid, _native.ST_ROWMAJOR, C, C, 4)
        if S2 < 2 or not self.lib.lvae_mlp_sk_supported(C, hid, S1, S2):
            return None
        return S1, S2

    def mlp_sk(self, *, y, M, C, hid, S1, S2, w1, b1, w2, b2, gamma, res, out, label='mlp'):
        """out = res + gamma * (fc2(gelu(fc1(y) + b1)) + b2) with y pre-split (lvae_dwconv_ln_h2), fc1 in S1 and fc2 in S2 K slices."""
        assert self.prec == 4
        w1h, w2h = self.w16_k32.get(w1), self.w16_k32.get(w2)
        assert w1h and w2h, f'{label}: weights do not fit the pre-split operand format'
        d = _native.MlpSkDesc()
        d.y, d.w1, d.b1, d.w2, d.b2, d.gamma, d.res, d.out = y, w1h, b1, w2h, b2, gamma, res, out
        d.ws = self.buf(self.sname('splitk_ws'), S2 * M * C).data_ptr()
        d.M, d.C, d.hid, d.S1, d.S2 = M, C, hid, S1, S2
        self.keep.append(d)
        self.add(self.lib.lvae_mlp_sk, (ctypes.byref(d),), label)
        self.flops += 4 * M * C * hid

    # ---- execution
    # opt-in: measured +-0.5% at B=1 (the path is GPU-latency-bound, not launch-bound) and HIP's global capture mode
    # conflicts with the two pipeline-group threads launching concurrently (hipErrorStreamCaptureInvalidated).
    use_graphs = os.environ.get('LVAE_GRAPHS', '0') == '1'

    # Replay: a launch range is compiled once into a native segment (an array of lvae_op: entry point id + arguments by class) and run
    # by ONE foreign call that needs no interpreter state (csrc/plan_runtime.cpp) -- with several pipeline groups launching from their
    # own threads, one ctypes call per launch made the interpreter lock the schedule.  LVAE_PY_REPLAY=1 keeps the per-launch Python
    # loop (debugging: the failing launch's label comes with the error either way).
    py_replay = os.environ.get('LVAE_PY_REPLAY', '0') == '1'

    def _segment(self, lo, hi):
        key = (lo, hi, len(self.ops))
        seg = self.segments.get(key)
        if seg is None:
            ops = self.ops[lo:hi]
            arr = (_native.Op * max(1, len(ops)))()
            for o, (fn, args, label, side) in zip(arr, ops):
                o.side = int(bool(side))
                if fn is _ORDER:
                    o.kind, o.p[0], o.i[0] = _native.OP_ORDER, args[1], int(bool(args[0]))
                    continue
                o.kind = _native.OP_KINDS[fn.__name__]
                np_, ni, nf = 0, 0, 0
                for a, t in zip(args, fn.argtypes[:-1]):
                    if t in (ctypes.c_float, ctypes.c_double):
                        o.f[nf] = float(a); nf += 1
                    elif t in (ctypes.c_int, ctypes.c_long):
                        o.i[ni] = int(a); ni += 1
                    else:                                   # pointer: None, an address, or a byref() of a kept descriptor
                        o.p[np_] = a if (a is None or isinstance(a, int)) else ctypes.cast(a, ctypes.c_void_p).value
                        np_ += 1
            seg = self.segments[key] = (arr, len(ops))
        return seg

    def _run_native(self, lo, hi, s, seg=None):
        """seg = (array, count): a prepared segment of ops lo .. hi (models/base.py: the host-aliased copy of a decode's first segment)."""
        arr, n = seg if seg is not None else self._segment(lo, len(self.ops) if hi is None else hi)
        bad = ctypes.c_int(-1)
        ss = self.side_stream.cuda_stream if self.side_stream is not None else None
        rc = self.lib.lvae_run_ops(arr, n, ctypes.c_void_p(s), ctypes.c_void_p(ss) if ss is not None else None, ctypes.byref(bad))
        if rc != 0:
            label = self.ops[lo + bad.value][2] if bad.value >= 0 else '?'
            raise RuntimeError(f'native launch "{label}" failed: rc={rc}')

    def _run_eager(self, lo, hi, s):
        if not self.py_replay:
            return self._run_native(lo, hi, s)
        sp = ctypes.c_void_p(s)
        ss = ctypes.c_void_p(self.side_stream.cuda_stream) if self.side_stream is not None else None
        for fn, args, label, side in self.ops[lo:hi]:
            if fn is _ORDER:
                rc = self.lib.lvae_stream_order(sp, ss, args[1]) if args[0] else self.lib.lvae_stream_order(ss, sp, args[1])
            else:
                rc = fn(*args, ss if side else sp)
            if rc != 0:
                raise RuntimeError(f'native launch "{label}" failed: rc={rc}')

    def run(self, lo=0, hi=None, stream=None):
        """Replay ops[lo:hi] on `stream` (raw hipStream_t; default: torch's current stream).  The first use of a range
        runs eagerly (also warms one-time kernel attributes); the second use captures it into a hipGraph (all buffers are
        pre-allocated, so the capture contains kernel nodes only); later uses replay the graph with ONE host call instead
        of one ctypes call per launch."""
        if torch.cuda.current_device() != self.device.index:      # launches go to the plan's GPU whatever the caller's current device
            with torch.cuda.device(self.device):
                return self.run(lo, hi, stream)
        cur = torch.cuda.current_stream(self.device)
        s = stream if stream is not None else cur.cuda_stream
        key = (lo, hi)
        if self.use_graphs and s == cur.cuda_stream:
            g = self.graphs.get(key)
            if g:
                g.replay()
                return
            if g is None and key in self.seen and cur.cuda_stream != 0:
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=cur):
                        self._run_eager(lo, hi, torch.cuda.current_stream(self.device).cuda_stream)
                    self.graphs[key] = g
                    g.replay()
                    return
                except Exception as e:      # capture unsupported in this context: stay eager for this range
                    self.graphs[key] = False
                    import warnings
                    warnings.warn(f'hipGraph capture failed for launch range {key}: {e}; running eagerly')
            self.seen.add(key)
        self._run_eager(lo, hi, s)


def ptr(t, offset_elems=0):
    """Device address of a torch tensor (+ element offset)."""
    return t.data_ptr() + offset_elems * t.element_size()
