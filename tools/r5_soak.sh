R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r5
mkdir -p $O
cd $R
python tools/soak.py --seconds ${SOAK_S:-600} 2>/dev/null | tail -1 > $O/soak_determinism.json; cat $O/soak_determinism.json
PROF_PART=b bash tools/prof_r5.sh > /dev/null 2>&1
cat $O/pmc_hbm_traffic.txt | head -20
head -20 $O/pmc_gemm_mfma_util.txt
