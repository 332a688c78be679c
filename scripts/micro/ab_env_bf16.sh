# usage: bash scripts/micro/ab_env_bf16.sh VAR a b [repeats]  -- configs[4] (MobileNetV3-Large 1024 x 2048, bf16, batch 1) with VAR alternating
VAR=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do for v in $A $B; do
env $VAR=$v python bench.py --model mobilenetv3large --size 1024 --width 2048 --classes 19 --batch 1 --dtype bf16 --steps 60 --warmup 15 --no-other-configs --no-cpu-baseline --no-streaming 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['value'], d['config'].get('launches_per_step'))"
done; done
