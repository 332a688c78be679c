#!/bin/bash
set -euo pipefail
# usage: pmc_kernel.sh "<counters>" <kernel substring> -- <python args>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
if [ $# -lt 4 ]; then echo 'usage: pmc_kernel.sh "<counters>" <kernel substring> -- <python args>' >&2; exit 2; fi
C="$1"; K="$2"; shift 3
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_tmp
# (the profiled run's output goes to a log; under `set -e` a failing run would otherwise end the script without a word)
if ! rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_tmp -- python3 "$@" > gpurun_out/pmc_kernel.log 2>&1; then
  echo "pmc_kernel.sh: the profiled run failed; tail of gpurun_out/pmc_kernel.log:" >&2; tail -20 gpurun_out/pmc_kernel.log >&2; exit 1
fi
python3 - "$K" <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
f = glob.glob('gpurun_out/pmc_tmp/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(list); dur = []
for r in csv.DictReader(open(f)):
    if k in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in agg.items():
    v = v[len(v)//2:]
    print('%-28s mean %.4g (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
