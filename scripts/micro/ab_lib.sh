# usage: bash scripts/micro/ab_lib.sh <other libdl3p.so> [repeats]  -- bench.py (headline + configs[2..4]) with the in-tree library and with another build, alternating on one box
OTHER=$(realpath $1); R=${2:-2}
for i in $(seq $R); do for v in tree other; do
if [ $v = other ]; then export DL3P_LIB_OVERRIDE=$OTHER; else unset DL3P_LIB_OVERRIDE; fi
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-streaming 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], [o['ms_per_step'] for o in d.get('other_configs', [])])"
done; done
