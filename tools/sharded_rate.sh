# per-GPU rate of the sharded evaluation on a synthetic CLIC-like set (1 rank): one image at a time vs same-size batches
R=$GRAFT_REPO_ROOT
cd $R
python scripts/eval-sharded.py --synthetic 16 --backend gloo -s 2 -l 64 1024 --max-batch 1 --partition stride 2>&1 | grep -v amdgpu
python scripts/eval-sharded.py --synthetic 16 --backend gloo -s 2 -l 64 1024 2>&1 | grep -v amdgpu
