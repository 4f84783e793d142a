"""Evaluation harness: drop-in for the reference's `imcoding_evaluate` (lvae/evaluation.py:15-67).

Same contract: sorted rglob('*.*') over the dataset folder; per image compress_file -> file size*8 ->
decompress_file; PSNR on the un-rounded float reconstruction; bpp over the ORIGINAL pixel count; mean of per-image
values.  `imcoding_evaluate_sharded` is the multi-GPU form (one process per GPU, images rank::world, one tiny
all_gather of per-image stats over RCCL/xGMI) -- the only collective on this path.
"""
import math
from collections import defaultdict
from pathlib import Path
from tempfile import gettempdir

import torch

from .paths import known_datasets
from .utils.coding import pil_to_tensor01


def _list_images(dataset):
    root = known_datasets.get(dataset, Path(dataset))
    img_paths = list(Path(root).rglob('*.*'))
    img_paths.sort()
    return img_paths


def _mse(real, fake):
    """mean((real - fake)^2) of one image (evaluation.py:47-49).  On the GPU: the native fp64-accumulating reduction
    (lvae_sqerr_partials_f32: deterministic) on the decoder's output where it lies -- no 12 B/pixel device-to-host copy and CPU pass per image;
    CPU tensors (stub codecs in tests): the reference's expression.  Every caller goes through here, so the sharded and the
    single-process evaluations agree bit for bit."""
    fake = fake.squeeze(0)
    if fake.is_cuda:
        import ctypes
        from . import _native
        a = fake.contiguous()
        b = real.to(a.device, non_blocking=True).contiguous()
        nblk = 512
        out = torch.empty(nblk, dtype=torch.float64, device=a.device)
        with torch.cuda.device(a.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
            _native.check(_native.lib().lvae_sqerr_partials_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), nblk, a.numel(), st), 'sqerr')
        total = 0.0
        for v in out.cpu().tolist():          # fixed order: deterministic across runs and processes
            total += v
        return total / a.numel()
    return (real - fake.cpu()).square().mean().item()


def _eval_one(model, impath, tmp_bits_dir, tag=''):
    from PIL import Image
    tmp_bits_path = tmp_bits_dir / f'{impath.stem}{tag}.bits'
    model.compress_file(impath, tmp_bits_path)
    num_bits = tmp_bits_path.stat().st_size * 8
    fake = model.decompress_file(tmp_bits_path)
    tmp_bits_path.unlink()
    real = pil_to_tensor01(Image.open(impath))
    mse = _mse(real, fake)
    return {'bpp': float(num_bits / float(real.shape[1] * real.shape[2])), 'mse': float(mse),
            'psnr': float(-10 * math.log10(mse))}


@torch.no_grad()
def imcoding_evaluate(model, dataset, progress=False):
    """dict {bpp, mse, psnr}: dataset means of per-image values (evaluation.py:59-66)."""
    assert hasattr(model, 'compress_file') and hasattr(model, 'decompress_file')
    img_paths = _list_images(dataset)
    tmp_bits_dir = Path(gettempdir())
    sums, n = defaultdict(float), 0
    it = img_paths
    if progress:
        from tqdm import tqdm
        it = tqdm(img_paths, ascii=True)
    for impath in it:
        stats = _eval_one(model, impath, tmp_bits_dir)
        n += 1
        for k, v in stats.items():      # timm AverageMeter: running sum / count
            sums[k] += v
    return {k: v / n for k, v in sums.items()}


def shard_paths(img_paths, rank, world):
    """Rank r of W codes sorted(img_paths)[r::W] -- the partition for same-size sets."""
    return img_paths[rank::world]


def padded_pixels(path, div=64):
    """Pixel count of an image after padding to multiples of `div` (what the codec actually processes); header read only."""
    from PIL import Image
    with Image.open(path) as img:
        h, w = img.height, img.width
    return (div * math.ceil(h / div)) * (div * math.ceil(w / div)), (div * math.ceil(h / div), div * math.ceil(w / div))


def lpt_partition(costs, world):
    """Longest-processing-time-first partition of items with the given costs over `world` ranks (mixed-size sets
    such as CLIC-2022, where rank::world leaves the ranks with the portrait/landscape 2048-wide images late).  Deterministic on
    every rank: items sorted by (-cost, index), each given to the least-loaded rank (ties -> lowest rank).  Returns a list of
    `world` index lists, each in ascending index order."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads, parts = [0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i)
        loads[r] += costs[i]
    return [sorted(p) for p in parts]


def batch_same_size(indices, shapes, max_batch=8, max_pixels=8 * 512 * 768 * 4):
    """Group a rank's images by padded size into batches (<= max_batch images, <= max_pixels padded pixels in total): the GPU part
    of a batch runs batched and its rANS streams are coded by parallel host threads, instead of one image at a time."""
    groups, out = {}, []
    for i in indices:
        groups.setdefault(shapes[i], []).append(i)
    for shape, idxs in groups.items():
        per = max(1, min(max_batch, max_pixels // (shape[0] * shape[1])))
        for o in range(0, len(idxs), per):
            out.append(idxs[o:o + per])
    out.sort(key=lambda b: b[0])
    return out


def _decode_images(paths):
    """PIL images of `paths`, fully decoded (run on a helper thread: PNG decoding releases the GIL)."""
    from PIL import Image
    imgs = []
    for p in paths:
        img = Image.open(p)
        img.load()
        imgs.append(img)
    return imgs


def _eval_batch(model, paths, tmp_bits_dir, tag='', images=None):
    """_eval_one for a batch of same-padded-size images through the model's batched file API (bit-identical per image); every
    PNG is decoded once (`images`: already decoded by the prefetch thread)."""
    if not hasattr(model, 'compress_files'):
        return [_eval_one(model, p, tmp_bits_dir, tag) for p in paths]
    imgs = images if images is not None else _decode_images(paths)
    bits = [tmp_bits_dir / f'{p.stem}{tag}.{k}.bits' for k, p in enumerate(paths)]
    model.compress_files(paths, bits, images=imgs)
    fakes = model.decompress_files(bits)
    out = []
    for img, b, fake in zip(imgs, bits, fakes):
        num_bits = b.stat().st_size * 8
        b.unlink()
        real = pil_to_tensor01(img)
        mse = _mse(real, fake)
        out.append({'bpp': float(num_bits / float(real.shape[1] * real.shape[2])), 'mse': float(mse),
                    'psnr': float(-10 * math.log10(mse))})
    return out


def gather_stats(local, world, device=None):
    """all_gather of per-image (index, bpp, mse, psnr) rows as float64; returns rows sorted by image index so that the
    mean is computed in exactly the single-process order."""
    import torch.distributed as dist
    t = torch.tensor(local, dtype=torch.float64).reshape(-1, 4)
    if device is not None:
        t = t.to(device)
    counts = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
    mx = max(int(c.item()) for c in counts)
    pad = torch.zeros(mx, 4, dtype=torch.float64, device=t.device)
    pad[:t.shape[0]] = t
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    rows = torch.cat([b[:int(c.item())] for b, c in zip(bufs, counts)], 0).cpu()
    return rows[torch.argsort(rows[:, 0])]


@torch.no_grad()
def imcoding_evaluate_sharded(model, dataset, partition='lpt', max_batch=8):
    """Same result as imcoding_evaluate (to the last bit: per-image values do not depend on batching, means are formed in image
    order), with the image list sharded over torch.distributed ranks: LPT by padded pixel count (`partition='stride'`:
    rank::world), and inside a rank same-size images coded as batches of up to `max_batch`."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    img_paths = _list_images(dataset)
    tmp_bits_dir = Path(gettempdir())
    meta = [padded_pixels(p, getattr(model, 'max_stride', 64)) for p in img_paths]
    if partition == 'lpt':
        mine = lpt_partition([m[0] for m in meta], world)[rank]
    else:
        mine = list(range(rank, len(img_paths), world))
    local = []
    batches = batch_same_size(mine, [m[1] for m in meta], max_batch=max_batch)
    # the next batch's PNGs are decoded on a helper thread while the GPU and the coder threads work on the current one
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = pool.submit(_decode_images, [img_paths[i] for i in batches[0]]) if batches else None
        for bi, batch in enumerate(batches):
            imgs = nxt.result()
            nxt = pool.submit(_decode_images, [img_paths[i] for i in batches[bi + 1]]) if bi + 1 < len(batches) else None
            stats = _eval_batch(model, [img_paths[i] for i in batch], tmp_bits_dir, tag=f'.r{rank}',
                                images=imgs if hasattr(model, 'compress_files') else None)
            for idx, s in zip(batch, stats):
                local.append([float(idx), s['bpp'], s['mse'], s['psnr']])
    dev = next(model.parameters()).device
    rows = gather_stats(local, world, dev if dist.get_backend() == 'nccl' else None)
    assert rows.shape[0] == len(img_paths)
    out = {}
    for j, k in enumerate(('bpp', 'mse', 'psnr')):
        acc = 0.0
        for v in rows[:, j + 1].tolist():
            acc += v
        out[k] = acc / rows.shape[0]
    return out
