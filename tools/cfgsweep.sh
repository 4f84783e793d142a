for cfg in -1 3 4 7 8 0; do echo "== LVAE_GEMM_CFG=$cfg"; LVAE_GEMM_CFG=$cfg python tools/microbench.py gemm 8 2>&1 | grep -E "^s4 |^s8 |^s16|total" ; done
