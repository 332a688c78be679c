#!/bin/bash
set -uo pipefail
# where do the waves of the fused inverted-residual kernels spend their cycles?  SQ wait / issue counters, two passes
# usage: irb_pmc.sh [shape key of scripts/micro/irb_bench.py, default 257]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
S=${1:-257}
O=gpurun_out/irb_pmc
rm -rf $O; mkdir -p $O
export IRB_CT=${IRB_CT:-1} IRB_WAVES=${IRB_WAVES:-8192} IRB_A_WAVES=${IRB_A_WAVES:-8192} IRB_B_WAVES=${IRB_B_WAVES:-6144}
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/a -- python3 scripts/micro/irb_bench.py $S > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/b -- python3 scripts/micro/irb_bench.py $S > $O/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c -- python3 scripts/micro/irb_bench.py $S > $O/c.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ('a', 'b'):
    f = glob.glob('$O/%s/**/*counter_collection.csv' % tag, recursive=True)
    if not f:
        print(tag, 'no counters', open('$O/%s.log' % tag).read()[-400:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        kn = r['Kernel_Name']
        if 'irb_' not in kn: continue
        agg[kn[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn, c in agg.items():
        print(kn)
        for k, v in sorted(c.items()):
            print('   %-28s %14.0f  (n=%d)' % (k, sum(v) / len(v), len(v)))
f = glob.glob('$O/c/**/*kernel_stats.csv', recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if 'irb_' in r['Name'] or 'pw_small' in r['Name'] or 'dw_fwd' in r['Name']:
            print('%-80s calls %s avg %.1f us' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3))
PY
