# microbench of the in-tree library vs experimental builds: tools/exp_ab.sh <what> <B> name1 name2 ...
R=$GRAFT_REPO_ROOT
what=$1; B=$2; shift 2
echo "== base"; (cd $R && LVAE_PREC=2 python tools/microbench.py $what $B 2>&1 | grep -v amdgpu)
for n in "$@"; do
  echo "== $n"; (cd $R && LVAE_PREC=2 LVAE_LIB=_bin/$n/liblvae_hip.so python tools/microbench.py $what $B 2>&1 | grep -v amdgpu)
done
