#!/bin/bash
set -euo pipefail
# where do the waves of the tiled GEMM spend their cycles?  SQ wait / issue counters of one shape (separate passes)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
S=${1:-"266256 304 256"}
O=gpurun_out/gemm_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/a -- python3 ${GEMM_PMC_SCRIPT:-scripts/micro/gemm_shape.py} $S > $O/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/b -- python3 ${GEMM_PMC_SCRIPT:-scripts/micro/gemm_shape.py} $S > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ('a', 'b'):
    f = glob.glob('$O/%s/**/*counter_collection.csv' % tag, recursive=True)
    if not f:
        print(tag, 'no counters', open('$O/%s.log' % tag).read()[-400:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        kn = r['Kernel_Name']
        if 'pw_gemm' not in kn and 'pw_wgrad' not in kn: continue
        agg[kn[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn, c in agg.items():
        print(kn)
        for k, v in sorted(c.items()):
            print('   %-28s %14.0f  (n=%d)' % (k, sum(v) / len(v), len(v)))
PY
