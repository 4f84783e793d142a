R=$GRAFT_REPO_ROOT
cd $R
run() { echo "== $*"; env "$@" LVAE_PREC=4 python tools/microbench.py gemmx 2>&1 | grep -v amdgpu | grep -E "K=  384|K=  768|K=  192"; }
for st in 0 1 2 3 4; do run LVAE_H2P=22 LVAE_H2P_STAGGER=$st; done
for st in 0 2 4; do run LVAE_H2P=42 LVAE_H2P_STAGGER=$st; done
