"""Shared host-side machinery of the codecs: batch -> pipeline groups (one HIP stream + one host thread each, so that a
group's host rANS coding overlaps the other group's GPU work), coder thread budget."""
import ctypes
import functools
import logging
import os

import numpy as np
import torch
import torch.nn as nn

from ..engine import NonFiniteError


import threading
_W16_LOCK = threading.Lock()


def split_bf16x3(t):
    """Exact 3-term bf16 split of an fp32 tensor: returns a (3, *t.shape) bf16 tensor [hi, mid, lo] with hi+mid+lo == t up to
    2^-25 |t| (same arithmetic as the kernel applies to activations: RNE conversions, exact fp32 residuals)."""
    hi = t.to(torch.bfloat16)
    r1 = t - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo], 0).contiguous()


def pack_bf16x3(t):
    """The prec-2 weight buffer of lvae_gemm_f32 for an [N][K] fp32 weight: the three bf16 planes hi | mid | lo ([3][N][K]),
    followed -- when K % 32 == 0 -- by the same values in k16-interleaved order [N][K/16][3][16] (what the double-buffered
    gemm_x3k16_kernel streams: one 16-deep stage of one row is 96 contiguous bytes)."""
    planes = split_bf16x3(t)
    n, k = t.shape
    if k % 32:
        return planes.reshape(-1)
    inter = planes.view(3, n, k // 16, 16).permute(1, 2, 0, 3).contiguous()
    return torch.cat([planes.reshape(-1), inter.reshape(-1)])


def split_f16x2(t):
    """2-term fp16 split of an fp32 tensor: (2, *t.shape) fp16 [hi, lo'] with hi + lo' / 2048 == t up to 2^-24 |t| -- the
    arithmetic csrc/gemm_h2.hip applies to activations: hi = f16(t) (RNE), lo' = f16((t - hi) * 2048) (the residual is exact)."""
    hi = t.to(torch.float16)
    lo = ((t - hi.float()) * 2048.0).to(torch.float16)
    return torch.stack([hi, lo], 0).contiguous()


def pack_f16x2(t):
    """The prec-4 weight buffer of lvae_gemm_f32 for an [N][K] fp32 weight (K % 16 == 0): fp16 planes hi | lo' in k16-interleaved
    order [N][K/16][2][16] -- one 16-deep stage of one row is 64 contiguous bytes.  None when a weight does not fit fp16's range
    (the caller keeps that GEMM on the bf16x3 arithmetic)."""
    n, k = t.shape
    if k % 16 or not bool(torch.isfinite(t).all()) or float(t.abs().max()) >= 65504.0:
        return None
    planes = split_f16x2(t.float())
    return planes.view(2, n, k // 16, 16).permute(1, 2, 0, 3).contiguous().reshape(-1)


def pack_f16x2_k32(t):
    """An [N][K] fp32 matrix (K % 32 == 0) in the plane format H2K32 = [N][K/32][2][32] fp16 (include/lvae_hip.h: lvae_gemm_desc.a_h2):
    per row and 32 k, the 32 hi terms then the 32 lo' terms -- what csrc/gemm_h2p.hip streams for both operands.  None when a value
    does not fit fp16's range."""
    n, k = t.shape
    if k % 32 or not bool(torch.isfinite(t).all()) or float(t.abs().max()) >= 65504.0:
        return None
    planes = split_f16x2(t.float())
    return planes.view(2, n, k // 32, 32).permute(1, 2, 0, 3).contiguous().reshape(-1)


def unpack_f16x2_k32(buf, n, k):
    """Inverse of the H2K32 layout (tests): (hi, lo') fp16 tensors [n][k] of a buffer holding n rows of k elements."""
    v = buf.view(torch.float16).reshape(n, k // 32, 2, 32)
    return v[:, :, 0, :].reshape(n, k), v[:, :, 1, :].reshape(n, k)


def pack_mxfp8(t):
    """The prec-3 weight buffer of lvae_gemm_f32 for an [N][K] fp32 weight: OCP MX-fp8 -- e4m3 elements with one E8M0
    (power-of-two) scale per 32 consecutive k of a row -- as a uint8 tensor: [N][Kp] element bytes followed by [N][Kp/32] scale
    bytes, Kp = K rounded up to a multiple of 64 (zero padded).  Block scale: 2^e with amax / 2^e in (224, 448] (e4m3's largest
    finite value is 448); elements rounded to nearest even by torch's float8_e4m3fn conversion -- the same rule the kernel applies
    to activations with v_cvt_pk_fp8_f32."""
    n, k = t.shape
    kp = (k + 63) // 64 * 64
    w = torch.zeros(n, kp, dtype=torch.float32, device=t.device)
    w[:, :k] = t.float()
    blk = w.view(n, kp // 32, 32)
    amax = blk.abs().amax(dim=2)
    bits = amax.view(torch.int32)
    eb = ((bits >> 23) & 0xff) - 8
    eb = eb + ((bits & 0x7fffff) > 0x600000).to(torch.int32)
    eb = eb.clamp(1, 254)
    inv = ((254 - eb) << 23).view(torch.float32)                      # 2^(127 - eb), exact
    q = (blk * inv.unsqueeze(2)).to(torch.float8_e4m3fn).view(torch.uint8).reshape(n, kp)
    return torch.cat([q.reshape(-1), eb.to(torch.uint8).reshape(-1)]).contiguous()


def pack_mxfp8_q8(t):
    """An [N][K] fp32 matrix (K % 64 == 0) in the MX-fp8 operand format Q8 of csrc/gemm_q8.hip (include/lvae_hip.h: lvae_gemm_desc.a_h2
    with prec 3): N*K e4m3 bytes row-major, then the E8M0 block scales as [K/64][N][2].  Same element / scale rule as pack_mxfp8."""
    n, k = t.shape
    assert k % 64 == 0
    buf = pack_mxfp8(t)
    data, sc = buf[:n * k], buf[n * k:].view(n, k // 64, 2)
    return torch.cat([data, sc.permute(1, 0, 2).contiguous().reshape(-1)]).contiguous()


def unpack_mxfp8_q8(buf, n, k):
    """Inverse of the Q8 layout (tests): the fp32 values an [n][k] Q8 buffer stands for."""
    buf = buf.view(torch.uint8).reshape(-1)
    data, sc = buf[:n * k], buf[n * k:n * k + n * k // 32].view(k // 64, n, 2).permute(1, 0, 2).contiguous().reshape(-1)
    return unpack_mxfp8(torch.cat([data, sc]), n, k)


def unpack_mxfp8(buf, n, k):
    """Inverse of pack_mxfp8 (tests): the fp32 values the MX-fp8 weights stand for, [N][K]."""
    kp = (k + 63) // 64 * 64
    q = buf[:n * kp].view(torch.float8_e4m3fn).float().view(n, kp // 32, 32)
    eb = buf[n * kp:].to(torch.int32).view(n, kp // 32)
    scale = torch.pow(2.0, (eb - 127).double()).float()
    return (q * scale.unsqueeze(2)).view(n, kp)[:, :k].contiguous()


class LazyW16:
    """{address of an fp32 GEMM weight: address of its reduced-precision copy}, filled ON FIRST USE by Plan.gemm: only tensors
    that are actually passed as `Wt` get a bf16 / bf16x3 copy (the packed dict also holds the AdaLN matrix, depthwise tables,
    biases ... which never are).  Thread-safe: plans are recorded concurrently by the pipeline-group threads."""

    def __init__(self, tensors, mode):
        self.by_ptr = {t.data_ptr(): t for t in tensors.values()
                       if t.dim() == 2 and t.dtype == torch.float32 and t.shape[1] % 8 == 0}
        self.mode, self.map, self.keep = mode, {}, []

    def get(self, ptr):
        h = self.map.get(ptr)
        if h is not None:
            return h
        t = self.by_ptr.get(ptr)
        if t is None:
            return None
        with _W16_LOCK:
            h = self.map.get(ptr)
            if h is None:
                c = (pack_bf16x3(t) if self.mode == 'bf16x3' else pack_mxfp8(t) if self.mode == 'mxfp8'
                     else pack_f16x2(t) if self.mode == 'f16x2' else pack_f16x2_k32(t) if self.mode == 'f16x2k32'
                     else (pack_mxfp8_q8(t) if t.shape[1] % 64 == 0 else None) if self.mode == 'mxfp8q8'
                     else t.to(torch.bfloat16).contiguous())
                if c is None:                       # f16x2: a weight outside fp16's range
                    self.map[ptr] = 0
                    return None
                self.keep.append(c)
                h = self.map[ptr] = c.data_ptr()
        return h or None


def bf16x3_weight_map(tensors):
    m = LazyW16(tensors, 'bf16x3')
    return m, m.keep


def bf16_weight_map(tensors):
    m = LazyW16(tensors, 'bf16')
    return m, m.keep


def f16x2_weight_map(tensors):
    m = LazyW16(tensors, 'f16x2')
    return m, m.keep


def f16x2k32_weight_map(tensors):
    m = LazyW16(tensors, 'f16x2k32')
    return m, m.keep


def mxfp8q8_weight_map(tensors):
    m = LazyW16(tensors, 'mxfp8q8')
    return m, m.keep


def mxfp8_weight_map(tensors):
    m = LazyW16(tensors, 'mxfp8')
    return m, m.keep


# GEMM arithmetic of newly built models.  A bitstream decodes only under the arithmetic that produced it (the priors must match bit
# for bit) and the container -- the reference's, byte for byte -- does not record it: the default is FIXED here (no environment
# override), another mode is an explicit `model.set_gemm_precision(...)` call that must be made identically on both sides
# (DESIGN.md 4).  compress_mode() logs the active mode once per model.
DEFAULT_PRECISION = 'f16x2'
PRECISIONS = ('fp32', 'bf16', 'bf16x3', 'f16x2', 'fp8')
PREC_CODE = {'fp32': 0, 'bf16': 1, 'bf16x3': 2, 'fp8': 3, 'f16x2': 4}        # lvae_gemm_desc.prec
assert DEFAULT_PRECISION in PRECISIONS, DEFAULT_PRECISION
_log = logging.getLogger('lvae')


def on_model_device(fn):
    """Entry points that launch raw HIP kernels run with the MODEL's GPU as the current device (the launches go to streams of
    that device; the caller's current device may be another GPU of the node)."""
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        dev = self._dummy.device
        if dev.type == 'cuda' and torch.cuda.current_device() != dev.index:
            with torch.cuda.device(dev):
                return fn(self, *a, **k)
        return fn(self, *a, **k)
    return wrapper


class CodecBase(nn.Module):
    def _init_codec_base(self):
        self.coder_threads = 0          # 0 = all hardware threads
        self.pipeline_groups = int(os.environ.get('LVAE_GROUPS', '2'))
        self.enc_groups = int(os.environ.get('LVAE_ENC_GROUPS', '0'))      # 0 = use pipeline_groups
        self.dec_groups = int(os.environ.get('LVAE_DEC_GROUPS', '0'))
        self._streams = []
        self._pool = None
        # measurement hook (bench.py's roofline pass): the groups of a call run ONE AFTER THE OTHER, each on its own stream, instead of
        # concurrently from their threads -- the same plans and launches as the product configuration, but a launch bracketed by HIP
        # events on its stream is then alone on the GPU
        self.serial_groups = False
        # default: fp32-class accuracy on the bf16 matrix cores (same parity as the exact fp32 MFMA path, 1.2-1.5x faster)
        self._prec = DEFAULT_PRECISION

    def set_gemm_precision(self, mode):
        """'fp32': exact fp32 MFMA (fmaf chains); 'bf16x3': fp32-class accuracy from three-term bf16 splits on the bf16 MFMA
        (2.7x less matrix-pipe time); 'f16x2': fp32-class accuracy from two-term fp16 splits, three fp16 MFMAs per product step
        (half of bf16x3's again; GEMMs it does not cover -- 2x2 patch gathers, K % 32 != 0 -- run as bf16x3); 'bf16': operands rounded to bf16, fp32 activations in HBM; 'fp8': BASELINE config 5 --
        activations STORED as bf16 and every channel-mixing GEMM on the block-scaled MX-fp8 MFMA (visibly different numerics,
        half the HBM traffic).  Bitstreams are only decodable in the mode that produced them (the priors must match bit for bit)."""
        assert mode in PRECISIONS
        self._prec = mode
        self._prec_logged = None

    def _log_precision(self):
        """Called by compress_mode(): one log line per model and mode naming the arithmetic its bitstreams are tied to."""
        if getattr(self, '_prec_logged', None) != self._prec:
            self._prec_logged = self._prec
            # WARNING level: the container (the reference's, byte for byte) does not record the arithmetic, and a stream decoded under
            # another one fails -- with lvae.NonFiniteError at best, as a wrong picture at worst.  Builds before round 3 defaulted to 'bf16x3'.
            _log.warning("lvae: %s codes with GEMM arithmetic %r%s; a bitstream decodes only under the arithmetic that wrote it "
                         "(streams of builds that defaulted to 'bf16x3' need model.set_gemm_precision('bf16x3') on this side too)",
                         type(self).__name__, self._prec, ' (package default)' if self._prec == DEFAULT_PRECISION else ' (set explicitly)')

    # ---- one pipeline group's decode / encode as ONE foreign call (csrc/plan_runtime.cpp: lvae_decode_blocks / lvae_encode_blocks)
    native_group_loops = os.environ.get('LVAE_PY_GROUP_LOOP') != '1'        # debugging / A-B switch: '1' = the per-block Python loops
    status_checks = os.environ.get('LVAE_NO_STATUS_CHECK') != '1'           # A-B switch of tools/ab_status.sh ONLY: '1' = the group loops run
                                                                            # without the status word (what the non-finite guard costs)

    # The coder's arrays travel without copies (round 5): the index / quantize / dequantize launches of a group's native loops are recorded
    # with the plan's PINNED HOST arrays as their raster operands (device-mapped host memory; the kernels touch the raster in runs of 64
    # consecutive entries, csrc/pointwise.hip), so a decode has no blit launch between a segment and the coder or between the coder and the
    # next segment, and an encode none behind its segments.  'both' | 'dec' | 'enc' | 'none' (A-B switch LVAE_ZERO_COPY: tools/r5_zero_copy.sh); the per-block Python loops and the test hooks keep the device arrays.
    zero_copy_coder_io = os.environ.get('LVAE_ZERO_COPY', 'both')

    @staticmethod
    def _alias_host(pl, seg, n_ops):
        """A copy of a native segment whose launches address pl.sym_host / pl.idx_host wherever the recorded ones address pl.sym_all / pl.idx_all."""
        from .. import _native
        raster_ops = {_native.OP_KINDS[k] for k in ('lvae_prior_index_f32', 'lvae_prior_index_sk_f32', 'lvae_quantize_f32', 'lvae_quantize_sk_f32', 'lvae_dequantize_f32')}
        spans = [(pl.sym_all.data_ptr(), pl.sym_all.numel() * 4, pl.sym_host.data_ptr()), (pl.idx_all.data_ptr(), pl.idx_all.numel(), pl.idx_host.data_ptr())]
        out = (_native.Op * max(1, n_ops))()
        ctypes.memmove(out, seg, ctypes.sizeof(_native.Op) * n_ops)
        for o in out[:n_ops]:
            if o.kind not in raster_ops:
                continue
            for j in range(len(o.p)):
                a = o.p[j]
                if a:
                    for lo_, size, host in spans:
                        if lo_ <= a < lo_ + size:
                            o.p[j] = host + (a - lo_)
        return out

    @classmethod
    def _group_blocks(cls_, pl, kind, cuts, offs, n):
        """The plan's latent blocks as a native array, cached on the plan: `cuts` = op index after each block's segment, `offs` = its
        element offset into sym_all / idx_all; per_image from pl.lat_shapes."""
        from .. import _native
        key = '_native_blocks_' + kind
        cached = getattr(pl, key, None)
        if cached is None:
            cls = _native.DecBlock if kind == 'dec' else _native.EncBlock
            zero_copy = cls_.zero_copy_coder_io in (kind, 'both')
            arr = (cls * len(cuts))()
            segs, lo = [], 0
            for li, cut in enumerate(cuts):
                seg, n_ops = pl._segment(lo, cut)
                segs.append(seg)
                z, hw = pl.lat_shapes[li]
                o = offs[li]
                b = arr[li]
                b.ops, b.n_ops, b.per_image = ctypes.cast(seg, ctypes.c_void_p).value, n_ops, z * hw
                b.idx_dev, b.idx_host = pl.idx_all.data_ptr() + o, pl.idx_host.data_ptr() + o
                b.sym_dev, b.sym_host = pl.sym_all.data_ptr() + 4 * o, pl.sym_host.data_ptr() + 4 * o
                if zero_copy:
                    seg = cls_._alias_host(pl, seg, n_ops)
                    segs[-1] = seg
                    b.ops, b.idx_dev, b.sym_dev = ctypes.cast(seg, ctypes.c_void_p).value, None, None
                lo = cut
            tail, n_tail = pl._segment(lo, len(pl.ops)) if kind == 'dec' else (None, 0)
            if zero_copy and n_tail:                       # (the last block's symbols are read by the tail's first launch)
                tail = cls_._alias_host(pl, tail, n_tail)
            early = None
            if kind == 'dec' and len(cuts):                # the same blocks with block 0's launches taken out: _decode_group_native issues
                early = (cls * len(cuts))()                # that segment itself, before it looks at the strings
                ctypes.memmove(early, arr, ctypes.sizeof(arr))
                early[0].n_ops = 0
            cached = (arr, segs, tail, n_tail, early)
            setattr(pl, key, cached)
        return cached

    def _decode_group_native(self, pl, cuts, offs, n, strings, tables, nthreads, stream, T=None):
        """strings[b][li]: image b's stream of latent block li.  Runs the group's whole decode; the caller copies pl.out afterwards."""
        from .. import _native
        arr, segs, tail, n_tail, early = self._group_blocks(pl, 'dec', cuts, offs, n)
        nb = len(cuts)
        # The first segment (bias -> ... -> prior of the top latent block) does not depend on the bitstream: it is on its way to the GPU
        # before this thread turns to the strings (header parsing, container offsets, pointer tables: ~0.1-0.2 ms of interpreter time
        # per group that used to precede the first launch -- tools/dec_timeline.py).  The foreign call releases the interpreter lock,
        # so the other group's thread prepares meanwhile.
        if early is not None:
            with torch.cuda.device(pl.device):
                pl._run_native(0, cuts[0], stream.cuda_stream, seg=(segs[0], arr[0].n_ops))
            arr = early
        if callable(strings):
            strings = strings()                            # [image][block] -> bytes, or (bytes-like container, offset, length): no copies
        qcdf, cdf_len, offset = tables
        addr, size = [], []
        for li in range(nb):                               # block-major
            for b in range(n):
                s_ = strings[b][li]
                if isinstance(s_, tuple):
                    base, off, ln = s_
                    assert isinstance(base, bytes)
                    addr.append(ctypes.cast(ctypes.c_char_p(base), ctypes.c_void_p).value + off); size.append(ln)
                else:
                    addr.append(ctypes.cast(ctypes.c_char_p(s_), ctypes.c_void_p).value if len(s_) else None); size.append(len(s_))
        sp = (ctypes.c_void_p * len(addr))(*addr)
        sl = (ctypes.c_size_t * len(size))(*size)
        fb, fo = ctypes.c_int(-1), ctypes.c_int(-1)
        trace = getattr(self, 'dec_trace', None)             # measurement hook (tools/dec_timeline.py): a list collects per-block stamps
        secs = (ctypes.c_double * (64 if trace is not None else 2))()
        if trace is not None:
            secs[0], secs[1] = -64.0, _native.TRACE_MAGIC          # timeline request: capacity AND the magic word (include/lvae_hip.h)
        ss = pl.side_stream.cuda_stream if pl.side_stream is not None else None
        st_dev = pl.status_ptr() if self.status_checks else None
        with torch.cuda.device(pl.device):
            rc = _native.lib().lvae_decode_blocks(arr, nb, n, sp, sl, qcdf.ctypes.data, qcdf.shape[1], cdf_len.ctypes.data, offset.ctypes.data,
                                                  ctypes.cast(tail, ctypes.c_void_p) if n_tail else None, n_tail, st_dev, pl.status_host.data_ptr(),
                                                  ctypes.c_void_p(stream.cuda_stream),
                                                  ctypes.c_void_p(ss) if ss is not None else None, int(nthreads), ctypes.byref(fb), ctypes.byref(fo), secs)
        if rc != 0:
            # every error exit leaves the plan's status word CLEAN: a bit that stayed set on the device (a corrupt stream decoded against
            # garbage, a failed launch) would otherwise fail the next, healthy call on this cached plan
            stream.synchronize()
            word = int(pl.status_host[0]) if rc == -75 else 0
            self._reset_status(pl)
            if rc == -75:       # a stream that did not decode because its scale indexes came from NaN / inf prior parameters
                raise NonFiniteError(word, getattr(pl, 'prec_name', None), f'while decoding (latent block {fb.value} of {nb})')
            if rc == -74:
                raise ValueError(f'rANS decode failed in latent block {fb.value} (corrupt or truncated bitstream)')
            raise RuntimeError(f'native decode failed: rc={rc} at latent block {fb.value}, launch {fo.value}')
        if trace is not None:
            trace.append((n, nb, list(secs)))
        if T is not None:
            T['dec_gpu_seg'] = T.get('dec_gpu_seg', 0) + secs[0]
            T['dec_rans'] = T.get('dec_rans', 0) + secs[1]

    def _encode_group_native(self, pl, cuts, offs, n, tables, nthreads, stream, T=None):
        """Runs the group's whole encode (launches, progressive hand-over, rANS).  -> strings[li][b] (bytes)."""
        from .. import _native
        arr, _segs, _tail, _nt, _early = self._group_blocks(pl, 'enc', cuts, offs, n)
        nb = len(cuts)
        qcdf, cdf_len, offset = tables
        caps = [8 * pl.lat_shapes[li][0] * pl.lat_shapes[li][1] + 64 for li in range(nb)]
        outs = getattr(pl, '_native_enc_out', None)
        if outs is None:                                   # output buffers live on the plan (one per block and image)
            outs = pl._native_enc_out = [np.empty(caps[li], dtype=np.uint8) for li in range(nb) for _ in range(n)]
        op = (ctypes.c_void_p * len(outs))(*[x.ctypes.data for x in outs])
        oc = (ctypes.c_size_t * len(outs))(*[caps[li] for li in range(nb) for _ in range(n)])
        out_len = (ctypes.c_long * len(outs))()
        fb, fo = ctypes.c_int(-1), ctypes.c_int(-1)
        secs = (ctypes.c_double * 3)()
        ss = pl.side_stream.cuda_stream if pl.side_stream is not None else None
        st_dev = pl.status_ptr() if self.status_checks else None
        with torch.cuda.device(pl.device):
            rc = _native.lib().lvae_encode_blocks(arr, nb, n, op, oc, out_len, qcdf.ctypes.data, qcdf.shape[1], cdf_len.ctypes.data, offset.ctypes.data,
                                                  st_dev, pl.status_host.data_ptr(), ctypes.c_void_p(stream.cuda_stream),
                                                  ctypes.c_void_p(ss) if ss is not None else None, int(nthreads), ctypes.byref(fb), ctypes.byref(fo), secs)
        if rc in (-34, -75):    # out-of-range input (the reference's assert) / NaN or inf in a prior parameter or posterior mean
            word = int(pl.status_host[0])
            self._reset_status(pl)
            pl.raise_if_flagged(word, where='while encoding')
            raise RuntimeError(f'native encode reported rc={rc} without a status word')
        if rc != 0:
            self._reset_status(pl)
            raise RuntimeError(f'native encode failed: rc={rc} at latent block {fb.value}, launch {fo.value}')
        if T is not None:
            T['enc_launch'] = T.get('enc_launch', 0) + secs[0]
            T['enc_gpu_wait'] = T.get('enc_gpu_wait', 0) + secs[1]
            T['enc_rans'] = T.get('enc_rans', 0) + secs[2]
        return [[outs[li * n + b][:out_len[li * n + b]].tobytes() for b in range(n)] for li in range(nb)]

    @staticmethod
    def _reset_status(pl):
        """Error exits of the native group loops: wait for the plan's device, then zero its status word and the pinned mirror."""
        if getattr(pl, 'status', None) is not None:
            torch.cuda.synchronize(pl.device)
            pl.status.zero_()
            pl.status_host.zero_()
            torch.cuda.synchronize(pl.device)

    def _check_decoded(self, groups, plan_of):
        """decompress_batch's last step: every group's decode has been queued (its status word travels to pinned memory behind its
        tail segment); wait for the caller's stream -- which waits for the groups' -- and raise if a NaN / inf reached a prior
        parameter or the reconstruction.  The reference's protocol synchronises right after decompress() anyway
        (scripts/speedtest-lvae.py:34-36), so this costs nothing there; the reconstruction is never handed on unchecked."""
        if not self.status_checks:
            return
        torch.cuda.current_stream(self._dummy.device).synchronize()
        err = None
        for g, (_start, n) in enumerate(groups):
            try:
                plan_of(g, n).raise_if_flagged(where='while decoding')
            except ArithmeticError as e:                   # clear every group's word before raising
                err = err or e
        if err is not None:
            raise err

    # ---- test access (not on the hot path)
    @torch.no_grad()
    def _trace_blocks(self, pl, B, force_z=None):
        """Run an encode plan latent block by latent block (`pl.qcuts`: the op index right after each block's quantize launch)
        and copy out, per block, what the launches left in their buffers: symbols, indexes, pm, lv (the raw log-variance
        parameter, before softplus), qm -- numpy arrays shaped (B, z, hw) in the coder's NCHW raster order.  force_z[li], a
        (B, z, h, w) tensor or a pair (batch rows, (len(rows), z, h, w) tensor), overwrites the block's quantised latent before the
        blocks below it run (teacher forcing)."""
        out, lo = [], 0
        for li, cut in enumerate(pl.qcuts):
            pl.run(lo, cut)
            lo = cut
            pl.fetch_status()
            torch.cuda.synchronize(pl.device)
            pl.raise_if_flagged(where=f'(encode trace, latent block {li})')
            z, hw = pl.lat_shapes[li]
            M, o = B * hw, pl.sym_off[li]
            prm = pl.prm_bufs[li][:M * 2 * z].view(B, hw, 2 * z)
            nchw = lambda t: t.permute(0, 2, 1).contiguous().cpu().numpy()
            out.append(dict(symbols=pl.sym_all[o:o + M * z].view(B, z, hw).cpu().numpy(),
                            indexes=pl.idx_all[o:o + M * z].view(B, z, hw).cpu().numpy(),
                            pm=nchw(prm[:, :, :z]), lv=nchw(prm[:, :, z:]),
                            qm=nchw(pl.qm_bufs[li][:M * z].view(B, hw, z))))
            if force_z is not None and force_z[li] is not None:
                ld = pl.zhat_ld[li]
                rows, zt = force_z[li] if isinstance(force_z[li], tuple) else (list(range(B)), force_z[li])
                zt = zt.to(pl.device, torch.float32).reshape(len(rows), z, hw).permute(0, 2, 1)
                pl.zhat_bufs[li][:M * ld].view(B, hw, ld)[rows, :, :z] = zt
        torch.cuda.synchronize(pl.device)
        return out

    def _coder_threads_per_group(self, n_groups):
        if n_groups == 1:
            return self.coder_threads
        return max(1, (self.coder_threads or len(os.sched_getaffinity(0))) // n_groups)

    def _groups(self, B, kind=None):
        """Split a batch of B into contiguous groups [(start, size)] for the stream/thread pipeline."""
        want = {'enc': self.enc_groups, 'dec': self.dec_groups}.get(kind, 0) or self.pipeline_groups
        G = max(1, min(int(want), B))
        if B < 4:
            G = 1
        base, rem = divmod(B, G)
        out, o = [], 0
        for g in range(G):
            n = base + (1 if g < rem else 0)
            out.append((o, n))
            o += n
        return out

    def _run_groups(self, fn, groups):
        """Run fn(g, start, size, stream) for every group: inline for one group, else one host thread + HIP stream each."""
        dev = self._dummy.device
        while len(self._streams) < len(groups):
            self._streams.append(torch.cuda.Stream(device=dev))
        if len(groups) == 1:        # inline, but still on a side stream (the legacy default stream cannot be graph-captured)
            st, cur = self._streams[0], torch.cuda.current_stream(dev)
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                res = [fn(0, groups[0][0], groups[0][1], st)]
            cur.wait_stream(st)
            return res
        if self.serial_groups:
            cur, res = torch.cuda.current_stream(dev), []
            for g in range(len(groups)):
                st = self._streams[g]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    res.append(fn(g, groups[g][0], groups[g][1], st))
                st.synchronize()
                cur.wait_stream(st)
            return res
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=8)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))        # lambda tables / inputs produced on the caller's stream

        stagger = float(getattr(self, 'group_stagger_s', 0.0) or 0.0)       # study knob (tools/r6_stagger_groups.py): group g starts g x this later

        def work(g):
            st = self._streams[g]
            st.wait_event(ev)
            if stagger > 0.0 and g:
                import time as _t
                _t.sleep(g * stagger)              # (sleep, not a spin: a spinning Python thread keeps the GIL from the other groups' threads)
            with torch.cuda.stream(st):
                return fn(g, groups[g][0], groups[g][1], st)
        # the last group runs on the calling thread (no hand-over latency for it; the pool threads have theirs first)
        futs = [self._pool.submit(work, g) for g in range(len(groups) - 1)]
        res, err, last = [], None, None
        try:
            last = work(len(groups) - 1)
        except BaseException as e:                        # noqa: BLE001
            err = e
        for f in futs:                                    # every group finishes before an error is raised: the plans and streams
            try:                                          # of a failed call must be idle when the caller tries again
                res.append(f.result())
            except BaseException as e:                    # noqa: BLE001
                err = err or e
        res.append(last)
        if err is not None:
            raise err
        cur = torch.cuda.current_stream(dev)
        for g in range(len(groups)):                      # caller's stream sees the groups' results
            cur.wait_stream(self._streams[g])
        return res

