"""Host side of the per-latent Gaussian conditional entropy model.

Mirrors the reference's `DiscretizedGaussian` (lvae/models/entropy_coding.py:52-82, a CompressAI
`GaussianConditional` subclass) and the stock `GaussianConditional(None)` used by QRes-VAE
(lvae/models/qresvae/model.py:240,317-325): same buffers (`_offset`, `_quantized_cdf`, `_cdf_length`,
`lower_bound_scale.bound`, `likelihood_lower_bound.bound`), same `update()` arithmetic -- but
  * the CDF rows are quantised by the native `lvae_pmf_to_quantized_cdf` (C ABI) instead of CompressAI's pybind11,
  * `build_indexes` / `quantize` / `dequantize` run inside HIP kernels (lvae_prior_index_f32, lvae_quantize_f32,
    lvae_dequantize_f32) on uint8 indexes / int32 symbols, never as 63 compare passes or Python lists,
  * streams are produced by the native multi-threaded rANS coder (`lvae_rans_*_batch`).
The pmf itself is evaluated with the same torch ops, on the module's device, as the reference does
(`td.Normal.cdf` = 0.5*(1+erf(x/sqrt2)) in fp32 for QARV; 0.5*erfc(-x/sqrt2) for QRes), so the tables are
bit-identical to the reference's on the same device.
"""
import ctypes
import math

import numpy as np
import scipy.stats
import torch
import torch.nn as nn

from .. import _native


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
        return torch.exp(torch.linspace(math.log(0.11), math.log(20.0), steps=64))

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
        """GaussianConditional.update(): builds int32[n, max_len+2] CDF rows."""
        lib = _native.lib()
        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        table = self.scale_table
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length).item())
        samples = torch.abs(torch.arange(max_length, device=table.device).int() - pmf_center[:, None]).float()
        scale = table.unsqueeze(1).float()
        upper = self._standardized_cumulative((0.5 - samples) / scale)
        lower = self._standardized_cumulative((-0.5 - samples) / scale)
        pmf = (upper - lower).cpu().numpy()
        tail = (2 * lower[:, :1]).cpu().numpy()
        lengths = pmf_length.cpu().numpy()
        n = len(lengths)
        cdf = np.zeros((n, max_length + 2), dtype=np.int32)
        for i in range(n):
            L = int(lengths[i])
            prob = np.ascontiguousarray(np.concatenate([pmf[i, :L], tail[i]]).astype(np.float32))
            row = np.zeros(L + 2, dtype=np.uint32)
            rc = lib.lvae_pmf_to_quantized_cdf(prob.ctypes.data, L + 1, self.entropy_coder_precision, row.ctypes.data)
            if rc != 0:
                raise ValueError(f'pmf_to_quantized_cdf failed for scale {i}: rc={rc}')
            cdf[i, :L + 2] = row.astype(np.int32)
        dev = table.device
        self._quantized_cdf = torch.from_numpy(cdf).to(dev)
        self._offset = (-pmf_center).to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)
        self._host = None

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
