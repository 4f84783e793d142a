"""oracle/compressai_semantics.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (PyTorch CPU ops + the plain-C coder in rans_oracle.c) of the pieces of the
third-party package CompressAI that the reference's hot path calls.  CompressAI is NOT vendored in
/root/reference and NOT installed in this image (PyPI `compressai`, version unpinned:
/root/reference/README.md:67); this file restates its published behaviour as recalled in
SURVEY.md Appendix A.1, anchored on the reference's own call sites:

  lvae/models/entropy_coding.py:6-7,52-82   (subclassing GaussianConditional, ctor bypass at :60)
  lvae/models/qarv/model.py:95,106-108,112-113,124
  lvae/models/qresvae/model.py:63-67,84-92,240,275,324-325,338-340,355-356

PARITY UNPINNED against real CompressAI (no golden bitstreams exist in the reference; see
rans_oracle.c header).  The classes keep CompressAI's names/attributes so that they double as the
stand-in module when the *reference's own* Python is imported to generate golden vectors
(tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.stats
import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_oracle_lib(force=False):
    """Compile rans_oracle.c with gcc into oracle/librans_oracle.so (idempotent)."""
    src = os.path.join(_HERE, 'rans_oracle.c')
    out = os.path.join(_HERE, 'librans_oracle.so')
    if force or (not os.path.exists(out)) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-std=c99', '-shared', '-fPIC', '-o', out, src, '-lm'])
    return out


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_oracle_lib())
        c = ctypes
        lib.oracle_pmf_to_quantized_cdf.restype = c.c_int
        lib.oracle_pmf_to_quantized_cdf.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_void_p]
        lib.oracle_rans_encode_with_indexes.restype = c.c_long
        lib.oracle_rans_encode_with_indexes.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t, c.c_void_p, c.c_int,
                                                        c.c_void_p, c.c_void_p, c.c_void_p, c.c_size_t]
        lib.oracle_rans_decode_with_indexes.restype = c.c_int
        lib.oracle_rans_decode_with_indexes.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.c_void_p,
                                                        c.c_int, c.c_void_p, c.c_void_p, c.c_void_p]
        _LIB = lib
    return _LIB


# ------------------------------------------------------------------ compressai._CXX / compressai.ans
def pmf_to_quantized_cdf(pmf, precision=16):
    """compressai._CXX.pmf_to_quantized_cdf(list[float], int) -> list[int]  (SURVEY Appendix A.2)."""
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.zeros(p.size + 1, dtype=np.uint32)
    rc = _lib().oracle_pmf_to_quantized_cdf(p.ctypes.data, int(p.size), int(precision), out.ctypes.data)
    if rc != 0:
        raise ValueError(f'pmf_to_quantized_cdf failed rc={rc}')
    return out.astype(np.int64).tolist()


def log_spaced_table(lo, hi, steps):
    """The reference's `torch.exp(torch.linspace(log lo, log hi, steps))` scale tables (entropy_coding.py:72-75,
    qresvae/model.py:60-67,317-325) with every rounding pinned (fp32 linspace with one rounding per element emulated in float64,
    float64 exp rounded once): the vectorised torch kernels differ in the last ulp between CPUs, and the oracle must build the same
    tables on the GPU box's host as here.  Bit-identical to the reference run behind tests/golden/*tables.npz."""
    import math
    start, end = np.float32(math.log(lo)), np.float32(math.log(hi))
    step = np.float32((end - start) / np.float32(steps - 1))
    s64, e64, d64 = float(start), float(end), float(step)
    t = np.array([np.float32(s64 + d64 * i) if i < steps // 2 else np.float32(e64 - d64 * (steps - 1 - i)) for i in range(steps)],
                 dtype=np.float32)
    return torch.from_numpy(np.exp(t.astype(np.float64)).astype(np.float32))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _cdf_matrix(cdfs):
    m = _i32(cdfs)
    assert m.ndim == 2
    return m


class RansEncoder:
    """compressai.ans.RansEncoder."""
    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets):
        sym, idx, cdf = _i32(symbols), _i32(indexes), _cdf_matrix(cdfs)
        sizes, offs = _i32(cdfs_sizes), _i32(offsets)
        assert sym.shape == idx.shape
        cap = 16 + 8 * sym.size * 4
        out = np.empty(cap, dtype=np.uint8)
        n = _lib().oracle_rans_encode_with_indexes(sym.ctypes.data, idx.ctypes.data, sym.size, cdf.ctypes.data,
                                                   cdf.shape[1], sizes.ctypes.data, offs.ctypes.data,
                                                   out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError(f'oracle encode failed rc={n}')
        return out[:n].tobytes()


class RansDecoder:
    """compressai.ans.RansDecoder."""
    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes, offsets, as_array=False):
        idx, cdf = _i32(indexes), _cdf_matrix(cdfs)
        sizes, offs = _i32(cdfs_sizes), _i32(offsets)
        buf = np.frombuffer(encoded, dtype=np.uint8)
        out = np.empty(idx.size, dtype=np.int32)
        rc = _lib().oracle_rans_decode_with_indexes(buf.ctypes.data, buf.size, idx.ctypes.data, idx.size,
                                                    cdf.ctypes.data, cdf.shape[1], sizes.ctypes.data,
                                                    offs.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f'oracle decode failed rc={rc}')
        return out if as_array else out.tolist()


# ------------------------------------------------------------------ compressai.ops.LowerBound
class LowerBound(nn.Module):
    """compressai.ops.LowerBound: forward = max(x, bound); registers buffer `bound` (1,)."""
    def __init__(self, bound):
        super().__init__()
        self.register_buffer('bound', torch.Tensor([float(bound)]))

    def forward(self, x):
        return torch.max(x, self.bound)


# ------------------------------------------------------------------ compressai.entropy_models
class EntropyModel(nn.Module):
    def __init__(self, likelihood_bound=1e-9, entropy_coder=None, entropy_coder_precision=16):
        super().__init__()
        self._encoder, self._decoder = RansEncoder(), RansDecoder()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer('_offset', torch.IntTensor())
        self.register_buffer('_quantized_cdf', torch.IntTensor())
        self.register_buffer('_cdf_length', torch.IntTensor())

    def quantize(self, inputs, mode, means=None):
        if mode not in ('noise', 'dequantize', 'symbols'):
            raise ValueError(f'Invalid quantization mode: "{mode}"')
        if mode == 'noise':
            half = float(0.5)
            noise = torch.empty_like(inputs).uniform_(-half, half)
            return inputs + noise
        outputs = inputs.clone()
        if means is not None:
            outputs -= means
        outputs = torch.round(outputs)  # half-to-even
        if mode == 'dequantize':
            if means is not None:
                outputs += means
            return outputs
        return outputs.int()

    @staticmethod
    def dequantize(inputs, means=None, dtype=torch.float):
        if means is not None:
            outputs = inputs.type_as(means)
            outputs += means
        else:
            outputs = inputs.type(dtype)
        return outputs

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32, device=pmf.device)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = torch.IntTensor(pmf_to_quantized_cdf(prob.tolist(), self.entropy_coder_precision))
            cdf[i, : _cdf.size(0)] = _cdf
        return cdf

    # array_io = True: tensors go to the C coder as numpy arrays instead of through CompressAI's Python-list interface (what the
    # reference really pays per call: .tolist() of every symbol).  Same bytes; used by bench.py's cpu_baseline so that the timed
    # host baseline is not handicapped by list conversions.
    array_io = False

    def compress(self, inputs, indexes, means=None):
        symbols = self.quantize(inputs, 'symbols', means)
        assert inputs.dim() >= 2 and inputs.size() == indexes.size()
        assert self._quantized_cdf.numel() > 0, 'Uninitialized CDFs. Run update() first'
        if self.array_io:
            cdf, sizes, offs = self._quantized_cdf.numpy(), self._cdf_length.reshape(-1).int().numpy(), self._offset.reshape(-1).int().numpy()
            return [self._encoder.encode_with_indexes(symbols[i].reshape(-1).int().numpy(), indexes[i].reshape(-1).int().numpy(),
                                                      cdf, sizes, offs) for i in range(symbols.size(0))]
        cdf, sizes, offs = self._quantized_cdf.tolist(), self._cdf_length.reshape(-1).int().tolist(), \
            self._offset.reshape(-1).int().tolist()
        strings = []
        for i in range(symbols.size(0)):
            strings.append(self._encoder.encode_with_indexes(
                symbols[i].reshape(-1).int().tolist(), indexes[i].reshape(-1).int().tolist(), cdf, sizes, offs))
        return strings

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        assert isinstance(strings, (tuple, list)) and len(strings) == indexes.size(0)
        assert self._quantized_cdf.numel() > 0, 'Uninitialized CDFs. Run update() first'
        cdf = self._quantized_cdf
        outputs = cdf.new_empty(indexes.size())
        if self.array_io:
            cdfn, sizes, offs = cdf.numpy(), self._cdf_length.reshape(-1).int().numpy(), self._offset.reshape(-1).int().numpy()
            for i, s in enumerate(strings):
                values = self._decoder.decode_with_indexes(s, indexes[i].reshape(-1).int().numpy(), cdfn, sizes, offs, as_array=True)
                outputs[i] = torch.from_numpy(values).reshape(outputs[i].size())
            return self.dequantize(outputs, means, dtype)
        cdfl, sizes, offs = cdf.tolist(), self._cdf_length.reshape(-1).int().tolist(), \
            self._offset.reshape(-1).int().tolist()
        for i, s in enumerate(strings):
            values = self._decoder.decode_with_indexes(s, indexes[i].reshape(-1).int().tolist(), cdfl, sizes, offs)
            outputs[i] = torch.tensor(values, device=outputs.device, dtype=outputs.dtype).reshape(outputs[i].size())
        return self.dequantize(outputs, means, dtype)


class GaussianConditional(EntropyModel):
    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        if not isinstance(scale_table, (type(None), list, tuple)):
            raise ValueError(f'Invalid type for scale_table "{type(scale_table)}"')
        if scale_table and (scale_table != sorted(scale_table) or any(s <= 0 for s in scale_table)):
            raise ValueError(f'Invalid scale_table "({scale_table})"')
        self.tail_mass = float(tail_mass)
        if scale_bound is None and scale_table:
            scale_bound = self.scale_table[0]
        if scale_bound <= 0:
            raise ValueError('Invalid parameters')
        self.lower_bound_scale = LowerBound(scale_bound)
        self.register_buffer('scale_table', self._prepare_scale_table(scale_table) if scale_table else torch.Tensor())
        self.register_buffer('scale_bound', torch.Tensor([float(scale_bound)]) if scale_bound is not None else None)

    @staticmethod
    def _prepare_scale_table(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    def _standardized_cumulative(self, inputs):
        half = float(0.5)
        const = float(-(2 ** -0.5))
        return half * torch.erfc(const * inputs)  # erfc form (stock CompressAI)

    @staticmethod
    def _standardized_quantile(quantile):
        return scipy.stats.norm.ppf(quantile)

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        device = self.scale_table.device
        self.scale_table = self._prepare_scale_table(scale_table).to(device)
        self.update()
        return True

    def update(self):
        multiplier = -self._standardized_quantile(self.tail_mass / 2)
        pmf_center = torch.ceil(self.scale_table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        device = pmf_center.device
        samples = torch.abs(torch.arange(max_length, device=device).int() - pmf_center[:, None])
        samples_scale = self.scale_table.unsqueeze(1)
        samples = samples.float()
        samples_scale = samples_scale.float()
        upper = self._standardized_cumulative((0.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self._quantized_cdf = quantized_cdf
        self._offset = -pmf_center
        self._cdf_length = pmf_length + 2

    def _likelihood(self, inputs, scales, means=None):
        half = float(0.5)
        values = inputs - means if means is not None else inputs
        scales = self.lower_bound_scale(scales)
        values = torch.abs(values)
        upper = self._standardized_cumulative((half - values) / scales)
        lower = self._standardized_cumulative((-half - values) / scales)
        return upper - lower

    def forward(self, inputs, scales, means=None, training=None):
        if training is None:
            training = self.training
        outputs = self.quantize(inputs, 'noise' if training else 'dequantize', means)
        likelihood = self._likelihood(outputs, scales, means)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        return outputs, likelihood

    def build_indexes(self, scales):
        scales = self.lower_bound_scale(scales)
        indexes = scales.new_full(scales.size(), len(self.scale_table) - 1).int()
        for s in self.scale_table[:-1]:
            indexes -= (scales <= s).int()
        return indexes
