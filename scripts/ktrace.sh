#!/bin/bash
set -euo pipefail
# usage: ktrace.sh <kernel substring> -- <python args> : mean kernel duration (us) from rocprofv3 kernel trace
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
if [ $# -lt 3 ]; then echo 'usage: ktrace.sh <kernel substring> -- <python args>' >&2; exit 2; fi
K="$1"; shift 2
mkdir -p gpurun_out
rm -rf gpurun_out/kt_tmp
if ! rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_tmp -- python3 "$@" > gpurun_out/ktrace.log 2>&1; then
  echo "ktrace.sh: the profiled run failed; tail of gpurun_out/ktrace.log:" >&2; tail -20 gpurun_out/ktrace.log >&2; exit 1
fi
python3 - "$K" <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
f = glob.glob('gpurun_out/kt_tmp/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if k in r['Kernel_Name']:
        d[r['Kernel_Name'][:50]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, v in d.items():
    v2 = sorted(v[len(v)//4:])
    print('%-50s n=%d mean %.2f us  median %.2f  min %.2f' % (n, len(v), sum(v2)/len(v2), v2[len(v2)//2], v2[0]))
PY
