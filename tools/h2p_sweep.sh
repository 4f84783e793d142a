R=$GRAFT_REPO_ROOT
cd $R
run() { echo "== $*"; env "$@" LVAE_PREC=4 python tools/microbench.py gemmx 2>&1 | grep -v amdgpu | grep -E "K= 4096|N=  768 K=  384"; }
run LVAE_H2P=22
run LVAE_H2P=22 LVAE_H2P_LDSPAD=40000
run LVAE_H2P=42
