for cfg in -1 4 8 7 3 0; do echo "== LVAE_GEMM_CFG=$cfg"; LVAE_GEMM_CFG=$cfg python tools/microbench.py gemm 8 2>&1 | grep -E "^s4 |^s8 |total" ; done
