"""not-gpu: the N>1 path (images sharded rank::world + one all_gather of per-image stats) on CPU with the
gloo backend, world_size 2: the sharded evaluation must reproduce the single-process means EXACTLY (fp64, image order)."""
import os
import struct
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import seeded_init


class _StubCodec(torch.nn.Module):
    """Deterministic stand-in with the model file API (the real model needs a GPU): 'compresses' by 4-bit quantisation."""
    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))

    def compress_file(self, img_path, output_path, lmb=None):
        from PIL import Image
        a = np.asarray(Image.open(img_path))
        with open(output_path, 'wb') as f:
            f.write(struct.pack('2H', a.shape[0], a.shape[1]) + (a >> 4).astype(np.uint8).tobytes()[::2])

    def decompress_file(self, bits_path):
        with open(bits_path, 'rb') as f:
            h, w = struct.unpack('2H', f.read(4))
        g = np.random.default_rng(h * 1000 + w)
        return torch.from_numpy(g.random((1, 3, h, w)).astype(np.float32))


def _make_images(d, n=5):
    from PIL import Image
    for i in range(n):
        Image.fromarray(seeded_init.synthetic_image_u8(32 + 8 * i, 40 + 4 * i, seed=i)).save(os.path.join(d, f'im{i:02d}.png'))


def _worker(rank, world, d, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lvae.evaluation import imcoding_evaluate_sharded
    res = imcoding_evaluate_sharded(_StubCodec(), d)
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_eval_equals_single(tmp_path):
    from lvae.evaluation import imcoding_evaluate, shard_paths
    d = str(tmp_path)
    _make_images(d)
    single = imcoding_evaluate(_StubCodec(), d)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, d, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == single, (res, single)
    paths = list(range(7))
    assert shard_paths(paths, 0, 2) == [0, 2, 4, 6] and shard_paths(paths, 1, 2) == [1, 3, 5]


def test_lpt_partition_and_same_size_batches():
    """SURVEY.md 8(e) host side: LPT by padded pixel count (every image exactly once, makespan <= 4/3 OPT + deterministic), batches
    of equal padded size bounded by count and pixels."""
    from lvae.evaluation import batch_same_size, lpt_partition
    clic = [1408 * 2048, 2048 * 1408, 1152 * 2048, 2048 * 1536] * 7 + [1408 * 2048, 2048 * 1408]       # 30 images, CLIC-like
    parts = lpt_partition(clic, 8)
    assert sorted(i for p in parts for i in p) == list(range(30)) and all(p == sorted(p) for p in parts)
    loads = [sum(clic[i] for i in p) for p in parts]
    stride = [sum(clic[i] for i in range(r, 30, 8)) for r in range(8)]
    assert max(loads) <= max(stride) and max(loads) <= 4 / 3 * (sum(clic) / 8) + max(clic) / 3
    assert lpt_partition(clic, 8) == parts and lpt_partition([5, 5, 5], 1) == [[0, 1, 2]] and lpt_partition([], 2) == [[], []]
    shapes = [(1408, 2048), (2048, 1408), (1408, 2048), (64, 64), (1408, 2048), (64, 64)] + [(1408, 2048)] * 5
    b = batch_same_size(list(range(11)), shapes, max_batch=8, max_pixels=4 * 1408 * 2048)
    assert sorted(i for g in b for i in g) == list(range(11))
    assert all(len({shapes[i] for i in g}) == 1 and len(g) <= 8 and len(g) * shapes[g[0]][0] * shapes[g[0]][1] <= 4 * 1408 * 2048 for g in b)
    assert [0, 2, 4, 6] in b and [3, 5] in b and [1] in b


def test_numa_cpulist_parser_and_noop_without_topology():
    from lvae.utils import numa
    assert numa._parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11] and numa._parse_cpulist('') == []
    before = os.sched_getaffinity(0)
    assert numa.pin_rank(0) is None or isinstance(numa.pin_rank(0), int)       # no GPU here: unreadable topology -> no-op
    if not torch.cuda.is_available():
        assert os.sched_getaffinity(0) == before
