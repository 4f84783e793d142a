"""Debug: encode a seeded image with qarv_base (default arithmetic) and dump the per-block symbols / indexes:
   python tools/debug_dw.py out.npz [H W B]          (run under different LVAE_GROUPS and diff)
   python tools/debug_dw.py --diff a.npz b.npz"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd')); sys.path.insert(0, os.path.join(REPO, 'tests'))
if sys.argv[1] == '--diff':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in sorted(a.files, key=lambda s: (int(s.split('_')[1]), s)):
        d = int((a[k] != b[k]).sum())
        print(f'{k:10s} n={a[k].size:7d} diff={d}')
    sys.exit(0)
import torch
import lvae, seeded_init
import lvae.engine as _eng
if os.environ.get('DBG_NOINK'):
    _eng.INKERNEL_REDUCE_MAX_BYTES = 0
if os.environ.get('DBG_NOSPLITK'):
    _eng.KSPLIT_MAX_TILES_PER_IMAGE = 0
if os.environ.get('DBG_NOSIDE'):
    import lvae.models.qarv.model as _qm
    _qm.SIDE_STREAM_MAX_PIXELS = 0
from conftest import load_seeded_into
from oracle import qarv_oracle
H, W, B = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (128, 192, 1)
sd = seeded_init.seeded_state_dict(qarv_oracle.qarv_param_shapes(qarv_oracle.qarv_base_arch()), seed=0)
m = lvae.get_model('qarv_base'); load_seeded_into(m, sd); m = m.to('cuda:0'); m.eval(); m.compress_mode()
im = torch.stack([torch.from_numpy(seeded_init.synthetic_image_u8(H, W, 5 + i)).permute(2, 0, 1).float() / 255 for i in range(B)]).cuda()
tr = m.encode_trace(im, 2048)
tr2 = m.encode_trace(im, 2048)
out = {}
for i, (d, d2) in enumerate(zip(tr, tr2)):
    out[f'sym_{i}'] = d['symbols']; out[f'idx_{i}'] = d['indexes']
    print(i, d['symbols'].shape, 'run-to-run diff sym', int((d['symbols'] != d2['symbols']).sum()), 'idx', int((d['indexes'] != d2['indexes']).sum()))
np.savez(sys.argv[1], **out)
