B="--steps 50 --warmup 15 --no-cpu-baseline --no-streaming --no-other-configs"
for f in 0 1; do echo "== WGRAD_SIDE=$f"; DL3P_WGRAD_SIDE=$f timeout 300 python bench.py --model mobilenetv3large --batch 1 --size 1024 --width 2048 --classes 19 --dtype bf16 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('final_loss'))"; done
DL3P_WGRAD_SIDE=1 timeout 300 python bench.py --model mobilenetv2 --batch 16 --dtype bf16 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mnv2 bf16 side=1', d['ms_per_step'], d['value'])"
DL3P_WGRAD_SIDE=0 timeout 300 python bench.py --model mobilenetv2 --batch 16 --dtype bf16 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mnv2 bf16 side=0', d['ms_per_step'], d['value'])"
timeout 600 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -4
