for cfg in 2 10 1 11 0; do echo "== LVAE_GEMM_CFG=$cfg (forced)"; LVAE_GEMM_CFG=$cfg python tools/microbench.py gemm 1 2>&1 | grep -E "^s16|^s32|^s64|total" ; done
