# band-count sweep for the depthwise window kernels (DL3P_DW_BALANCE=2: fewest equal bands reaching DL3P_DW_WANT)
for cfg in "1 384 16" "2 256 16" "2 384 16" "2 512 16" "2 768 16" "2 384 24" "2 512 32"; do
  set -- $cfg
  echo "BALANCE=$1 WANT=$2 MAXTH=$3"
  for i in 1 2; do DL3P_DW_BALANCE=$1 DL3P_DW_WANT=$2 DL3P_DW_MAXTH=$3 python bench.py --steps 20 --warmup 5 --cpu-steps 0 2>&1 | tail -1 | cut -c100-200; done
done
