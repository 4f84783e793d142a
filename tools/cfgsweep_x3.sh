for cfg in -1 0 3 4 6 8 1; do echo "== x3 LVAE_GEMM_CFG=$cfg"; LVAE_PREC=2 LVAE_GEMM_CFG=$cfg python tools/microbench.py gemm 8 2>&1 | grep -E "^s4 |^s8 |^s16|total" ; done
