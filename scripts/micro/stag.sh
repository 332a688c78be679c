for s in 0 2 4 6 8; do echo "stagger $s"; DL3P_GEMM_STAGGER=$s python scripts/gemm_sweep.py all big 2>&1 | grep -E "K= 304|K= 256 N= 256|K=  16" ; done
