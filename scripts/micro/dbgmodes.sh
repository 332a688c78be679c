for m in 0 102 100 101; do echo "mode $m"; DL3P_GEMM_BK64_MIN_K=100000 DL3P_GEMM_STAGGER=$m python scripts/gemm_sweep.py fwd big 2>&1 | grep -E "K= 304|K= 256 N= 256|K=  16 N=  96"; done
