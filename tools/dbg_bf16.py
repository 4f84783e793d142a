import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/lossy-vae_amd')
import torch, bench
dev=torch.device('cuda',0)
bench.PROFILE='typical'
m,sd=bench.build_model(dev)
m.set_gemm_precision('bf16')
ims=bench.synth_batch(8,512,768,0).to(dev)
for g in (1,2):
    m.pipeline_groups=g
    s1=m.compress_batch(ims); s2=m.compress_batch(ims)
    print('groups',g,'deterministic enc:', s1==s2, [a==b for a,b in zip(s1,s2)])
    try:
        x=m.decompress_batch(s1); print(' decode ok')
    except Exception as e: print(' decode failed', e)
m.pipeline_groups=1
s_all=m.compress_batch(ims)
single=[m.compress(ims[i:i+1]) for i in range(8)]
print('batch==single', [a==b for a,b in zip(s_all,single)])
