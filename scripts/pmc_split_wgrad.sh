#!/bin/bash
# LDS / MFMA counters of the split-bf16 weight-gradient kernel on the decoder shape ("what bounds it"), each profiler run under
# its own timeout:  pmc_split_wgrad.sh  -> gpurun_out/pmc_split_wgrad/r03_split_wgrad_counters.csv
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
O=gpurun_out/pmc_split_wgrad
mkdir -p $O
export SB_SHAPES=266256x256x256,266256x304x256,17424x1280x256
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/a -- python3 scripts/micro/sb_wgrad.py > $O/a.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/b -- python3 scripts/micro/sb_wgrad.py > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ('a', 'b'):
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'wgrad' in r['Kernel_Name']:
                agg[(r['Kernel_Name'][:60], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for v in agg.values() for c in v})
out = open('$O/r03_split_wgrad_counters.csv', 'w')
out.write('kernel,grid,calls,' + ','.join(names) + '\n')
for (kn, g), c in sorted(agg.items()):
    n = max(len(v) for v in c.values())
    out.write('"%s",%s,%d,' % (kn, g, n) + ','.join('%.0f' % (sum(c[x]) / len(c[x])) if c.get(x) else '' for x in names) + '\n')
out.close()
print(open('$O/r03_split_wgrad_counters.csv').read())
PY
tail -3 $O/a.log $O/b.log
