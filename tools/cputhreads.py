import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/lossy-vae_amd')
import torch, bench
m_sd = None
import seeded_init
from oracle import qarv_oracle
arch=qarv_oracle.qarv_base_arch()
sd=seeded_init.seeded_state_dict(qarv_oracle.qarv_param_shapes(arch),0,'typical')
for th in (8,16,32,64):
    r=bench.cpu_baseline(sd,512,768,n_images=1,threads=th)
    print(th, r['value'], r['sample'][-8:])
