"""Debug: qres34m 512x768 (seeded 'wide' weights) vs the CPU oracle, per-block flip counts.  LVAE_DW_CL=0 selects the earlier depthwise forms."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd')); sys.path.insert(0, os.path.join(REPO, 'tests'))
import torch
import lvae, seeded_init
from lvae import _native
if os.environ.get('LVAE_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])
from oracle import qres_oracle
sd = seeded_init.seeded_state_dict(qres_oracle.qres_param_shapes(qres_oracle.qres34m_arch()), seed=0)
m = lvae.get_model('qres34m'); full = m.state_dict()
for k, v in sd.items(): full[k] = torch.from_numpy(v)
m.load_state_dict(full); m.compress_mode(); m = m.to('cuda:0').eval()
orc = qres_oracle.QresOracle(sd); orc.compress_mode()
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 768)
im = torch.from_numpy(seeded_init.synthetic_image_u8(H, W, 31)).permute(2, 0, 1).float().div(255).unsqueeze(0)
tr = m.encode_trace(im.cuda()); otr = orc.encode_trace(im, code=False)
for i, (a, b) in enumerate(zip(tr, otr['blocks'])):
    f = int((a['symbols'].reshape(-1) != b['symbols'].numpy().reshape(-1)).sum()); g = int((a['indexes'].reshape(-1) != b['indexes'].numpy().reshape(-1)).sum())
    print(f'block {i:2d} n={a["symbols"].size:7d} sym flips {f:5d} idx flips {g:5d}')
