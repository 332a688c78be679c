# usage: bash scripts/micro/ab_env.sh VAR a b [repeats]  -- the headline step with VAR=a and VAR=b alternating on one box
VAR=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do for v in $A $B; do
env $VAR=$v python bench.py --steps 40 --warmup 10 --no-other-configs --no-cpu-baseline --no-streaming 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['value'])"
done; done
