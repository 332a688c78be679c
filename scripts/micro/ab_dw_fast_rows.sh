for v in 1 0 1 0; do
DL3P_DW_FAST_ROWS=$v python bench.py --steps 40 --warmup 10 --no-other-configs --no-cpu-baseline --no-streaming 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fast_rows=$v', d['ms_per_step'], d['value'])"
done
