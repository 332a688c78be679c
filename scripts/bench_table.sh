#!/bin/bash
# GPU box, repo root: the DESIGN.md §8 table (one bench.py line per configuration) -> gpurun_out/bench_table.txt
B="python3 bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 4"
run() { echo "== $*"; $B "$@" 2>&1 | grep '"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s  %.2f ms  %s launches' % (d['value'], d['ms_per_step'], d['config'].get('launches_per_step')))"; }
{
run --model mobilenetv2
run --model mobilenetv3large
run --model xception --batch 4
run --model xception --batch 4 --size 769 --classes 19
run --model mobilenetv2_lite
run --model mobilenetv2 --os 8
run --model mobilenetv3large --os 8
run --model xception --os 8 --size 769 --classes 19 --batch 2
run --model mobilenetv3large --size 1024 --width 2048 --classes 19 --batch 1
run --model mobilenetv3large --size 1024 --width 2048 --classes 19 --batch 2
run --model resnet50
run --model mobilenetv3small
run --model mobilenetv3small_lite
run --model mobilenetv3large_lite
} > gpurun_out/bench_table.txt 2>&1
cat gpurun_out/bench_table.txt
