"""not-gpu: host-side logic of the drop-in package: registry, state-dict compatibility, containers, padding."""
import os
import struct

import numpy as np
import pytest
import torch

import lvae
from lvae.utils import coding
from oracle import qarv_oracle


def test_registry_contract():
    assert 'qarv_base' in lvae.models.registry._all_models
    with pytest.raises(KeyError):
        lvae.get_model('no_such_model')
    assert lvae.get_model is lvae.models.registry.get_model


@pytest.fixture(scope='module')
def model():
    return lvae.get_model('qarv_base', lmb_range=(16, 2048))


def test_state_dict_keys_and_shapes_match_reference_inventory(model):
    """952 entries (907 parameters + 45 entropy-model buffers), 93.433 M parameters, reference key names."""
    sd = model.state_dict()
    assert len(sd) == 952
    want = dict(qarv_oracle.qarv_param_shapes(qarv_oracle.qarv_base_arch()))
    have = {k: tuple(v.shape) for k, v in model.named_parameters()}
    assert have == want
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 93433400
    bufs = [k for k in sd if k not in have]
    assert len(bufs) == 45 and all('.discrete_gaussian.' in k for k in bufs)
    assert {k.rsplit('.discrete_gaussian.', 1)[1] for k in bufs} == {
        '_offset', '_quantized_cdf', '_cdf_length', 'likelihood_lower_bound.bound', 'lower_bound_scale.bound'}


def test_attributes_used_by_the_harness_scripts(model):
    assert model.max_stride == 64 and model.lmb_range == (16.0, 2048.0) and model.default_lmb == 2048.0
    assert model.num_latents == 9 and hasattr(model, 'compress_file') and hasattr(model, 'decompress_file')
    model.default_lmb = 100.0          # eval-var-rate.py:41-42 sets it as a plain attribute
    assert model.default_lmb == 100.0
    model.default_lmb = 2048.0
    assert next(model.parameters()).device.type == 'cpu'


def test_compress_mode_before_to_device_builds_tables(model):
    model.compress_mode()              # eval-fix-rate.py:30 calls it before .to(device)
    for b in model.dec_blocks:
        if getattr(b, 'is_latent_block', False):
            assert tuple(b.discrete_gaussian._quantized_cdf.shape) == (64, 249)
    sd = model.state_dict()
    m2 = lvae.get_model('qarv_base')
    with pytest.raises(RuntimeError):
        m2.load_state_dict(sd)         # like the reference: empty buffers vs built tables is a size mismatch
    m2.compress_mode()
    m2.load_state_dict(sd)


def test_cpu_is_refused_loudly(model):
    model.compress_mode()
    with pytest.raises(RuntimeError, match='GPU'):
        model.compress(torch.rand(1, 3, 64, 64))


def test_pack_unpack_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pack_byte_strings.npz'))
    for i in range(3):
        joined, parts, o = g[f'case{i}.joined'].tobytes(), [], 0
        for n in g[f'case{i}.lengths'].tolist():
            parts.append(joined[o:o + n]); o += n
        packed = coding.pack_byte_strings(parts)
        assert packed == g[f'case{i}.packed'].tobytes()
        assert coding.unpack_byte_string(packed) == parts
    with pytest.raises(AssertionError):
        coding.unpack_byte_string(coding.pack_byte_strings([b'abc', b'de'])[:-1])
    assert coding.unpack_byte_string(coding.pack_byte_strings([])) == []


def test_pad_divisible_by_and_to_tensor():
    from PIL import Image
    a = (np.arange(70 * 100 * 3) % 251).astype(np.uint8).reshape(70, 100, 3)
    p = coding.pad_divisible_by(Image.fromarray(a), 64)
    assert (p.height, p.width) == (128, 128)
    pa = np.asarray(p)
    assert np.array_equal(pa[:70, :100], a) and np.array_equal(pa[100, :100], a[69]) and np.array_equal(pa[:70, 120], a[:, 99])
    assert np.array_equal(pa, qarv_oracle.pad_divisible_by_u8(a, 64))
    same = coding.pad_divisible_by(Image.fromarray(a[:64, :64]), 64)
    assert (same.height, same.width) == (64, 64)
    t = coding.pil_to_tensor01(Image.fromarray(a))
    assert t.shape == (3, 70, 100) and t.dtype == torch.float32 and float(t.max()) <= 1.0
    assert torch.equal(t, torch.from_numpy(a).permute(2, 0, 1).float().div(255))


def test_header_layout():
    # qarv/model.py:525-528,567: '2H'(h,w) | 'f'(lambda) | '3H'(nB,H/64,W/64) | 'B'(9) | '9I' | payload -> 51 bytes overhead
    body = struct.pack('f', 2048.0) + struct.pack('3H', 1, 8, 12) + coding.pack_byte_strings([b''] * 9)
    assert len(struct.pack('2H', 512, 768) + body) == 51


def test_auto_ksplit_is_batch_independent_and_valid():
    """lvae.engine.auto_ksplit takes the PER-IMAGE row count, so the number of K slices (and with it the summation order) cannot
    depend on the batch size; every returned S divides the k-tile count and keeps >= 4 k-tiles per slice."""
    from lvae import _native
    from lvae.engine import auto_ksplit
    RM = _native.ST_ROWMAJOR
    for m1, N, K in [(96, 512, 1024), (96, 512, 2048), (384, 512, 1536), (384, 1024, 512), (1536, 384, 768), (1536, 96, 3456),
                     (6144, 256, 448), (24576, 192, 384), (96, 32, 4608)]:
        s = auto_ksplit(m1, N, K, RM, N, N, 2)
        assert s >= 1 and (K // 32) % s == 0 and (s == 1 or K // 32 // s >= 4), (m1, N, K, s)
    assert auto_ksplit(96, 512, 1024, RM, 512, 512, 2) > 1                 # stride-64 MLP: few tiles, long K
    assert auto_ksplit(24576, 192, 384, RM, 192, 192, 2) == 1              # stride-4 layer: plenty of tiles
    assert auto_ksplit(96, 512, 1000, RM, 512, 512, 2) == 1                # K not a multiple of 32
    assert auto_ksplit(96, 510, 1024, RM, 510, 512, 2) == 1                # N not a multiple of 4
    assert auto_ksplit(96, 512, 1024, _native.ST_SHUFFLE, 512, 512, 2) == 1


def test_mlp_pipeline_keeps_the_slice_count_per_image_and_picks_the_form_by_batch():
    """engine.Plan.mlp_pipeline: for the split-K layers (maps below 1536 rows per image) the slice counts are auto_ksplit's of the
    per-image shape at EVERY batch size -- only the form (serial pre-split / parallel) follows the batch; large maps always run
    pre-split without split-K; fc2 is never pre-split unless fc1 is."""
    from lvae import _native
    from lvae.engine import Plan, auto_ksplit
    RM = _native.ST_ROWMAJOR

    def plan(B):
        pl = Plan.__new__(Plan)                      # host logic only: no device, no library
        pl.prec, pl.w16_k32, pl.B = 4, {}, B
        return pl
    for rows, C, hid in [(384, 512, 1024), (384, 512, 1536), (96, 512, 1024), (96, 512, 2048)]:
        s1, s2 = auto_ksplit(rows, hid, C, RM, hid, 0, 4), auto_ksplit(rows, C, hid, RM, C, C, 4)
        assert s1 > 1 and s2 > 1
        forms = []
        for B in (1, 2, 4, 8, 16):
            pre1, pre2, S1, S2 = plan(B).mlp_pipeline(C, hid, 3, rows)
            assert (S1 == s1 if pre1 else S1 is None) and (S2 == s2 if pre2 else S2 is None) and (pre1 or not pre2)
            forms.append((pre1, pre2))
        assert forms[-1][0]                          # a large batch takes the serial form for fc1
        assert all(a <= b for a, b in zip(forms, forms[1:]))       # ... monotonically: once serial, serial for larger batches
    assert plan(1).mlp_pipeline(384, 768, 7, 6144) == (True, True, 1, 1)          # stride-8 map: pre-split, no split-K
    assert plan(8).mlp_pipeline(384, 768, 7, 6144) == (True, True, 1, 1)
    assert plan(8).mlp_pipeline(144, 288, 7, 96) == (False, False, None, None)     # a width the pre-split producers do not have
    pl = plan(8); pl.prec = 2
    assert pl.mlp_pipeline(512, 1024, 3, 384) == (False, False, None, None)        # other arithmetics: untouched


def test_fused_mlp_rule_never_trades_split_k_bits_for_a_batch_threshold():
    """engine.Plan.mlp_fused_ok: a fused-MLP shape with a ROW threshold may follow the batch only where the
    two-launch alternative has the fused kernel's summation order (maps of >= 1536 rows per image: S = 1); on smaller maps the
    alternative is split-K, so the fused form is refused at EVERY batch size there -- qres34m's width-384 blocks on 256x256 images
    decode the same in a group of 48 as alone.  Shapes without a threshold are fused whatever the size; an oversized launch is cut
    into row ranges instead of changing pipeline."""
    from lvae import _native
    from lvae.engine import Plan

    class _Lib:
        lvae_mlp_h2f = object()

    def plan(B):
        pl = Plan.__new__(Plan)
        pl.prec, pl.w16_k32, pl.B = 4, {1: 11, 2: 22}, B
        pl.ops, pl.keep, pl.flops, pl.on_side, pl.lib = [], [], 0, False, _Lib()
        return pl
    for B in (1, 4, 8, 48, 96, 512):
        rows = 32 * 32                                   # a 256x256 image at stride 8
        assert not plan(B).mlp_fused_ok(384, 768, 5, M=B * rows, rows_per_image=rows), B
        assert plan(B).mlp_pipeline(384, 768, 5, rows)[2:] == plan(1).mlp_pipeline(384, 768, 5, rows)[2:]
        assert not plan(B).mlp_fused_ok(384, 768, 5, M=B * rows)              # no per-image row count given: never by M alone
    assert not plan(4).mlp_fused_ok(384, 768, 7, M=4 * 6144, rows_per_image=6144)      # below the threshold: two launches, S = 1 ...
    assert plan(8).mlp_fused_ok(384, 768, 7, M=8 * 6144, rows_per_image=6144)          # ... the same bits as the fused launch
    assert plan(4).mlp_pipeline(384, 768, 7, 6144) == (True, True, 1, 1)
    for B, rows in ((1, 64), (1, 24576), (8, 24576), (4000, 1024)):
        assert plan(B).mlp_fused_ok(192, 384, 7, M=B * rows, rows_per_image=rows)
        assert plan(B).mlp_fused_ok(128, 192, 7, M=B * rows, rows_per_image=rows)
    pl = plan(4000)                                       # 4000 x 1024 rows x 192 channels x 4 B = 3.1 GB: two row ranges, offsets in bytes
    M = 4000 * 1024
    pl.mlp_fused(y=1 << 40, M=M, C=192, hid=384, w1=1, b1=5, w2=2, b2=6, gamma=7, res=2 << 40, out=3 << 40)
    descs = [d for d in pl.keep if isinstance(d, _native.MlpDesc)]
    assert len(descs) == 2 and sum(d.M for d in descs) == M and all(d.M * 192 * 4 < 2 ** 31 for d in descs)
    assert descs[0].M % 128 == 0
    assert descs[1].y - descs[0].y == descs[0].M * 192 * 4 == descs[1].out - descs[0].out == descs[1].res - descs[0].res
    assert pl.flops == 4 * M * 192 * 384


def test_small_map_fused_mlp_rule_keeps_the_split_k_contract():
    """engine.Plan.mlp_sk_ok (round 6, csrc/mlp_sk.hip): the fused small-map MLP is taken only where the two-launch alternative is
    split-K (maps below 1536 rows per image), with the PER-IMAGE slice counts of auto_ksplit -- the summation order, hence the bits, do
    not depend on the batch -- for shapes the library supports, up to MLP_SK_MAX_ROWS rows per launch (a speed rule: same bits either
    way); never on other arithmetics, never on the large maps (S = 1 there)."""
    from lvae import _native
    from lvae.engine import Plan, auto_ksplit

    def plan(B, prec=4):
        pl = Plan.__new__(Plan)
        pl.prec, pl.w16_k32, pl.B = prec, {1: 11, 2: 22}, B
        pl.ops, pl.keep, pl.flops, pl.on_side, pl.lib, pl.bufs = [], [], 0, False, _native.lib(), {}
        return pl
    # qarv_base at 512x768: stride 64 (96 rows per image) and stride 32 (384 rows)
    assert plan(1).mlp_sk_ok(512, 2048, 1, 96, 96) == (4, 16) == plan(4).mlp_sk_ok(512, 2048, 1, 96, 384) == plan(8).mlp_sk_ok(512, 2048, 1, 96, 768)
    assert plan(1).mlp_sk_ok(512, 1536, 3, 384, 384) == (2, 8)
    assert plan(1).mlp_sk_ok(512, 1024, 3, 384, 384) == (4, 8) == (auto_ksplit(384, 1024, 512, 0, 1024, 0, 4), auto_ksplit(384, 512, 1024, 0, 512, 512, 4))
    assert plan(4).mlp_sk_ok(512, 1536, 3, 384, 4 * 384) is None                      # 1536 rows per launch: the serial split-K launches are faster there
    assert plan(1).mlp_sk_ok(384, 768, 5, 1536, 1536) is None                         # stride-16 map: no split-K, nothing to fuse this way
    assert plan(1).mlp_sk_ok(256, 448, 7, 96, 96) is None                             # shape the kernel has no instance for
    assert plan(1, prec=2).mlp_sk_ok(512, 2048, 1, 96, 96) is None                    # other arithmetics: untouched
    # whichever form runs, the slice counts are mlp_pipeline's
    for rows, C, hid, k in ((96, 512, 2048, 1), (96, 512, 1024, 1), (384, 512, 1536, 3), (384, 512, 1024, 3)):
        S1 = auto_ksplit(rows, hid, C, 0, hid, 0, 4); S2 = auto_ksplit(rows, C, hid, 0, C, C, 4)
        assert plan(1).mlp_sk_ok(C, hid, k, rows, rows) == (S1, S2)
        assert _native.lib().lvae_mlp_sk_supported(C, hid, S1, S2) == 1
    # argument errors come back as -22 before anything is launched (no GPU needed): a shape without an instance, no split-K, a NULL operand
    import ctypes
    d = _native.MlpSkDesc()
    for f in ('y', 'w1', 'b1', 'w2', 'b2', 'gamma', 'res', 'out', 'ws'):
        setattr(d, f, 4096)
    d.M, d.C, d.hid, d.S1, d.S2 = 96, 256, 448, 2, 2
    assert _native.lib().lvae_mlp_sk(ctypes.byref(d), None) == -22 and _native.lib().lvae_mlp_sk_supported(256, 448, 2, 2) == 0
    d.C, d.hid, d.S1, d.S2 = 512, 2048, 4, 1
    assert _native.lib().lvae_mlp_sk(ctypes.byref(d), None) == -22
    d.S2, d.ws = 16, None
    assert _native.lib().lvae_mlp_sk(ctypes.byref(d), None) == -22


def test_numa_pinning_groups_ranks_by_host(monkeypatch):
    """lvae/utils/numa.pin_ranks_collectively on a faked 2-node x 4-GPU job: ranks are grouped by HOSTNAME (identical cpulists on two
    machines must not be pooled), split their node's cores in global-rank order, and refuse a topology that covers < 90 % of the CPUs."""
    import os
    import socket
    from lvae.utils import numa
    node = {0: list(range(0, 8)), 1: list(range(8, 16))}                 # two NUMA nodes of 8 cores per machine, 2 GPUs each
    ranks = [('hostA', node[0]), ('hostA', node[0]), ('hostA', node[1]), ('hostA', node[1]),
             ('hostB', node[0]), ('hostB', node[0]), ('hostB', node[1]), ('hostB', node[1])]
    allowed = set(range(16))
    pinned = {}

    class FakeDist:
        def __init__(self, rank):
            self.rank = rank

        def get_rank(self):
            return self.rank

        def get_world_size(self):
            return len(ranks)

        def all_gather_object(self, out, mine):
            for r, (h, cp) in enumerate(ranks):
                out[r] = (h, cp, sorted(allowed))

    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(allowed))
    for r, (h, cp) in enumerate(ranks):
        monkeypatch.setattr(socket, 'gethostname', lambda h=h: h)
        monkeypatch.setattr(numa, 'gpu_numa_cpus', lambda idx, cp=cp: cp)
        monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, cpus, r=r: pinned.__setitem__(r, list(cpus)))
        assert numa.pin_ranks_collectively(r % 4, FakeDist(r)) == 4
    assert pinned[0] == [0, 1, 2, 3] and pinned[1] == [4, 5, 6, 7] and pinned[2] == [8, 9, 10, 11] and pinned[3] == [12, 13, 14, 15]
    assert [pinned[r] for r in range(4, 8)] == [pinned[r] for r in range(4)]          # the other machine: the same layout, not 8-way slices
    # a container that reports ONE node for every GPU: nothing is pinned ... unless forced (the single-GPU rehearsal)
    ranks[:] = [('hostA', node[0])] * 8
    pinned.clear()
    monkeypatch.setattr(socket, 'gethostname', lambda: 'hostA')
    monkeypatch.setattr(numa, 'gpu_numa_cpus', lambda idx: node[0])
    monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, cpus: pinned.__setitem__('x', list(cpus)))
    assert numa.pin_ranks_collectively(0, FakeDist(3)) is None and not pinned
    assert numa.pin_ranks_collectively(0, FakeDist(3), force=True) == 1 and pinned['x'] == [3]


def test_published_numbers_fixture_and_gate_skips_cleanly(tmp_path):
    """tests/golden/published/published_rd.json (the reference's published RD points, numbers only) is well-formed, and
    scripts/accept-published.py -- the acceptance gate for north_star's "identical bpp/PSNR on Kodak" -- exits 0 with a SKIP line when
    the trained checkpoints / image folders are not there (this offline build), before touching any GPU."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cases = json.load(open(os.path.join(repo, 'tests', 'golden', 'published', 'published_rd.json')))['cases']
    assert set(cases) == {'qarv_base', 'qres34m'}
    for model, per in cases.items():
        assert set(per) == {'kodak', 'clic2022-test', 'tecnick-rgb-1200'}
        for ds, c in per.items():
            n = len(c['lambdas'])
            assert n == (16 if model == 'qarv_base' else 8) and len(c['bpp']) == len(c['psnr']) == n
            assert all(a < b for a, b in zip(c['bpp'], c['bpp'][1:])) and all(a < b for a, b in zip(c['psnr'], c['psnr'][1:]))
    k = cases['qarv_base']['kodak']
    assert abs(k['bpp'][-1] - 2.2102898) < 1e-6 and abs(k['psnr'][-1] - 44.371040) < 1e-5 and k['lambdas'][-1] == 2048.0
    env = dict(os.environ, TORCH_HOME=str(tmp_path / 'th'), LVAE_DATASETS=str(tmp_path / 'ds'))
    for args in (['-m', 'qarv_base'], ['-m', 'qres34m', '-n', 'clic2022-test']):
        r = subprocess.run([sys.executable, os.path.join(repo, 'scripts', 'accept-published.py')] + args, env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith('SKIP: not available offline:'), (r.stdout, r.stderr)
