#!/usr/bin/env python
"""Multi-GPU evaluation (BASELINE.json config 4): one process per GPU, the image list sharded rank::world,
one all_gather of per-image (index, bpp, mse, psnr) rows over RCCL, means formed in image order on rank 0.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/eval-sharded.py \
        -m qarv_base -a "pretrained='qarv_base.pt'" -n clic2022-test -l 16 2048 -s 8
With --synthetic N it writes N seeded PNGs of CLIC-like mixed sizes to a temp folder (no datasets offline) and uses seeded
weights."""
import argparse
import math
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lossy-vae_amd'))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import lvae  # noqa: E402
from lvae.evaluation import imcoding_evaluate_sharded  # noqa: E402

CLIC_SIZES = [(1365, 2048), (2048, 1365), (1152, 2048), (2048, 1536)]      # (h, w) drawn round-robin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', type=str, default='qarv_base')
    ap.add_argument('-a', '--model_args', type=str, default='pretrained=True')
    ap.add_argument('-n', '--dataset_name', type=str, default='clic2022-test')
    ap.add_argument('-l', '--lmb_range', type=float, default=None, nargs='+')
    ap.add_argument('-s', '--steps', type=int, default=4)
    ap.add_argument('--synthetic', type=int, default=0)
    ap.add_argument('--backend', type=str, default='nccl')
    ap.add_argument('--max-batch', type=int, default=8, help='same-size images coded per batch inside a rank (1 = one image at a time)')
    ap.add_argument('--partition', type=str, default='lpt', choices=['lpt', 'stride'], help='LPT by padded pixels, or rank::world')
    args = ap.parse_args()
    rank, world, local = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
    if os.environ.get('LVAE_SINGLE_GPU_TEST') == '1':      # rehearsal of the N > 1 path on a 1-GPU box: every rank on cuda:0 (use --backend gloo)
        local = 0
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    from lvae.utils.numa import pin_ranks_collectively
    try:
        ncpu = None if os.environ.get('LVAE_SINGLE_GPU_TEST') == '1' else pin_ranks_collectively(local, dist, local, world)
    except Exception as e:
        print(f'[rank {rank}] NUMA pinning skipped: {e!r}', file=sys.stderr)
        ncpu = None
    kwargs = eval(f'dict({args.model_args})')
    dataset = args.dataset_name
    if args.synthetic:
        import seeded_init
        from PIL import Image
        kwargs['pretrained'] = False
        dataset = os.path.join(tempfile.gettempdir(), f'lvae_synth_{args.synthetic}')
        if rank == 0:
            os.makedirs(dataset, exist_ok=True)
            for i in range(args.synthetic):
                h, w = CLIC_SIZES[i % len(CLIC_SIZES)]
                Image.fromarray(seeded_init.synthetic_image_u8(h, w, seed=500 + i)).save(os.path.join(dataset, f'im{i:03d}.png'))
        dist.barrier()
    model = lvae.get_model(args.model, **kwargs)
    if args.synthetic:
        import seeded_init
        sd = model.state_dict()
        for k in sd:
            a = seeded_init.seeded_tensor(k, tuple(sd[k].shape), 0, profile='typical')
            if a is not None:
                sd[k] = torch.from_numpy(a)
        model.load_state_dict(sd)
    model = model.to(dev).eval()
    model.compress_mode()
    model.coder_threads = max(4, ncpu or (os.cpu_count() or 8) // max(1, world))     # coder / launch threads on the GPU's socket
    start, end = args.lmb_range or getattr(model, 'lmb_range', (0, 0))
    lambdas = torch.linspace(math.log(start), math.log(end), steps=args.steps).exp().tolist() if hasattr(model, 'default_lmb') else [None]
    for lmb in lambdas:
        if lmb is not None:
            model.default_lmb = lmb
        torch.cuda.synchronize(dev)
        t0 = time.time()
        res = imcoding_evaluate_sharded(model, dataset, partition=args.partition, max_batch=args.max_batch)
        dt = time.time() - t0
        if rank == 0:
            print(f'lambda={lmb}: {res}', flush=True)
            print(f'  [{world} rank(s), partition={args.partition}, max_batch={args.max_batch}] {dt:.2f} s for the set '
                  f'(file reads, PNG decode, PSNR and the all_gather included)', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
