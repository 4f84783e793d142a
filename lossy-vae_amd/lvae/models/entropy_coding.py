"""Host side of the per-latent Gaussian conditional entropy model.

Mirrors the reference's `DiscretizedGaussian` (lvae/models/entropy_coding.py:52-82, a CompressAI
`GaussianConditional` subclass) and the stock `GaussianConditional(None)` used by QRes-VAE
(lvae/models/qresvae/model.py:240,317-325): same buffers (`_offset`, `_quantized_cdf`, `_cdf_length`,
`lower_bound_scale.bound`, `likelihood_lower_bound.bound`), same `update()` arithmetic -- but
  * the CDF rows are quantised by the native `lvae_pmf_to_quantized_cdf` (C ABI) instead of CompressAI's pybind11,
  * `build_indexes` / `quantize` / `dequantize` run inside HIP kernels (lvae_prior_index_f32, lvae_quantize_f32,
    lvae_dequantize_f32) on uint8 indexes / int32 symbols, never as 63 compare passes or Python lists,
  * streams are produced by the native multi-threaded rANS coder (`lvae_rans_*_batch`).
The pmf follows the reference's fp32 expression (`td.Normal.cdf` = 0.5*(1+erf(x/sqrt2)) for QARV; 0.5*erfc(-x/sqrt2) for
QRes) but is ALWAYS evaluated on the host by `lvae_build_gaussian_tables` with a correctly rounded erf -- never by the GPU's
torch.erf -- so the tables are bit-identical to the ones the reference builds on the CPU whatever device the module lives on.
"""
import ctypes
import math

import numpy as np
import scipy.stats
import torch
import torch.nn as nn

from .. import _native


def log_spaced_table(lo, hi, steps):
    """exp(linspace(log lo, log hi, steps)) as float32 -- the scale tables of the reference (entropy_coding.py:72-75: 64 scales
    0.11..20; qresvae/model.py:317-325: 0.1..20; :60-67: 128 scales) -- evaluated so that every host gives the SAME bits: the
    reference's `torch.exp(torch.linspace(...))` runs through vectorised float kernels whose last ulp depends on the CPU (AVX2 vs
    AVX-512 builds of the same torch differ in a few entries), and a table that differs by one ulp moves CDF rows by a count, i.e.
    makes a stream undecodable on another machine.  Here: the fp32 linspace torch computes (start + step*i for the first half,
    end - step*(n-1-i) for the second, ONE rounding per element) emulated in float64, then exp in float64 rounded once to float32.
    Bit-identical to the tables of the reference run that produced tests/golden/*tables.npz (all three tables)."""
    start, end = np.float32(math.log(lo)), np.float32(math.log(hi))
    step = np.float32((end - start) / np.float32(steps - 1))
    s64, e64, d64 = float(start), float(end), float(step)
    t = np.array([np.float32(s64 + d64 * i) if i < steps // 2 else np.float32(e64 - d64 * (steps - 1 - i)) for i in range(steps)],
                 dtype=np.float32)
    return torch.from_numpy(np.exp(t.astype(np.float64)).astype(np.float32))


class _Bound(nn.Module):
    """State-dict-compatible stand-in for compressai.ops.LowerBound (one buffer `bound`, shape (1,))."""
    def __init__(self, bound):
        super().__init__()
        self.register_buffer('bound', torch.Tensor([float(bound)]))


class DiscretizedGaussian(nn.Module):
    def __init__(self, scale_table=None, cdf_form='erf', scale_bound=None, tail_mass=1e-9, persistent_table=False):
        super().__init__()
        assert cdf_form in ('erf', 'erfc')
        self.cdf_form = cdf_form
        self.entropy_coder_precision = 16
        self.likelihood_lower_bound = _Bound(1e-9)
        self.register_buffer('_offset', torch.IntTensor())
        self.register_buffer('_quantized_cdf', torch.IntTensor())
        self.register_buffer('_cdf_length', torch.IntTensor())
        if scale_table is None and cdf_form == 'erf':
            scale_table = self._get_default_scale_table()                      # entropy_coding.py:72-75
        if scale_table is None:
            scale_table = torch.Tensor()
        self.register_buffer('scale_table', torch.as_tensor(scale_table, dtype=torch.float32), persistent=persistent_table)
        self.tail_mass = float(tail_mass)
        if scale_bound is None:
            scale_bound = float(self.scale_table[0]) if self.scale_table.numel() else 0.11
        self.lower_bound_scale = _Bound(scale_bound)
        self._host = None      # numpy int32 copies for the coder

    @staticmethod
    def _get_default_scale_table():
        return log_spaced_table(0.11, 20.0, 64)

    def _standardized_cumulative(self, inputs):
        if self.cdf_form == 'erf':      # torch.distributions.Normal(0,1).cdf, entropy_coding.py:70,81-82
            return 0.5 * (1 + torch.erf(inputs / math.sqrt(2)))
        return 0.5 * torch.erfc(-(2 ** -0.5) * inputs)   # stock CompressAI

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        self.scale_table = torch.as_tensor(scale_table, dtype=torch.float32).to(self.scale_table.device)
        self.update()
        return True

    @torch.no_grad()
    def update(self):
        """GaussianConditional.update(): builds int32[n, max_len+2] CDF rows.

        The rows come from the native `lvae_build_gaussian_tables` (host C++: pmf in fp32 exactly as the reference's torch
        expression, erf/erfc correctly rounded from double, CompressAI's pmf_to_quantized_cdf) -- NOT from torch.erf on the
        module's device: a bitstream must decode on any box, so its tables may not depend on a GPU's (or a torch build's)
        last-ulp erf.  Bit-identical to the tables the reference builds on the CPU (tests/test_host_coder.py, both CDF forms)."""
        lib = _native.lib()
        multiplier = float(-scipy.stats.norm.ppf(self.tail_mass / 2))
        table = np.ascontiguousarray(self.scale_table.detach().cpu().numpy().astype(np.float32))
        n = int(table.size)
        if n == 0:
            raise ValueError('empty scale table')
        centers = np.ceil(table * np.float32(multiplier)).astype(np.int64)
        stride = int(2 * centers.max() + 1) + 2
        cdf = np.zeros((n, stride), dtype=np.int32)
        cdf_len = np.zeros(n, dtype=np.int32)
        offset = np.zeros(n, dtype=np.int32)
        rc = lib.lvae_build_gaussian_tables(table.ctypes.data, n, multiplier, 0 if self.cdf_form == 'erf' else 1,
                                            cdf.ctypes.data, stride, cdf_len.ctypes.data, offset.ctypes.data)
        if rc != stride:
            raise ValueError(f'lvae_build_gaussian_tables failed: rc={rc} (expected row length {stride})')
        dev = self.scale_table.device
        self._quantized_cdf = torch.from_numpy(cdf).to(dev)
        self._offset = torch.from_numpy(offset).to(dev)
        self._cdf_length = torch.from_numpy(cdf_len).to(dev)
        self._host = (cdf, cdf_len, offset)

    def host_tables(self):
        """(qcdf int32 [n][stride], cdf_len int32 [n], offset int32 [n]) as contiguous numpy arrays."""
        if self._quantized_cdf.numel() == 0:
            raise RuntimeError('Uninitialized CDFs. Run compress_mode()/update() first')
        if self._host is None:
            self._host = (np.ascontiguousarray(self._quantized_cdf.cpu().numpy().astype(np.int32)),
                          np.ascontiguousarray(self._cdf_length.cpu().numpy().astype(np.int32)),
                          np.ascontiguousarray(self._offset.cpu().numpy().astype(np.int32)))
        return self._host

    def _apply(self, fn, *a, **k):      # .to()/.cuda() moves buffers: drop the host cache
        self._host = None
        return super()._apply(fn, *a, **k)


# ----------------------------------------------------------------------------------------------- batched host coding
def _ptr_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def rans_encode_streams(tables, sym_views, idx_views, n_threads=0):
    """Encode len(sym_views) independent streams (numpy int32 / uint8 views, e.g. slices of pinned buffers) with the
    native threaded coder.  Returns a list of bytes."""
    lib = _native.lib()
    qcdf, cdf_len, offset = tables
    ns = len(sym_views)
    if ns == 0:
        return []
    sizes = [int(s.size) for s in sym_views]
    caps = [8 * n + 64 for n in sizes]
    outs = [np.empty(c, dtype=np.uint8) for c in caps]
    out_len = (ctypes.c_long * ns)()
    rc = lib.lvae_rans_encode_batch(
        ns, _ptr_array([s.ctypes.data for s in sym_views]), _ptr_array([i.ctypes.data for i in idx_views]),
        (ctypes.c_size_t * ns)(*sizes), qcdf.ctypes.data, qcdf.shape[1], cdf_len.ctypes.data, offset.ctypes.data,
        _ptr_array([o.ctypes.data for o in outs]), (ctypes.c_size_t * ns)(*caps), out_len, int(n_threads))
    if rc != 0:
        raise RuntimeError(f'lvae_rans_encode_batch failed rc={rc}')
    return [outs[i][:out_len[i]].tobytes() for i in range(ns)]


def rans_decode_streams(tables, strings, idx_views, sym_out_views, n_threads=0):
    """Decode streams into the given int32 output views (in place)."""
    lib = _native.lib()
    qcdf, cdf_len, offset = tables
    ns = len(strings)
    if ns == 0:
        return
    bufs = [np.frombuffer(s, dtype=np.uint8) for s in strings]
    sizes = [int(i.size) for i in idx_views]
    status = (ctypes.c_int * ns)()
    rc = lib.lvae_rans_decode_batch(
        ns, _ptr_array([b.ctypes.data for b in bufs]), (ctypes.c_size_t * ns)(*[b.size for b in bufs]),
        _ptr_array([i.ctypes.data for i in idx_views]), (ctypes.c_size_t * ns)(*sizes),
        qcdf.ctypes.data, qcdf.shape[1], cdf_len.ctypes.data, offset.ctypes.data,
        _ptr_array([o.ctypes.data for o in sym_out_views]), status, int(n_threads))
    if rc != 0:
        raise ValueError(f'lvae_rans_decode_batch failed rc={rc} (corrupt or truncated bitstream)')
