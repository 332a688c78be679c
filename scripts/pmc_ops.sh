#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / MFMA-pipe counters + kernel durations for the top launches of configs[2] and configs[4] (scripts/pmc_ops.py),
# every profiler run under its own timeout (a wedged profiler must not eat the lease):  pmc_ops.sh r03
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
R=${1:-r03}
O=gpurun_out/pmc_ops_$R
mkdir -p $O
for cfg in ${PMC_CFGS:-xception xception769 headline_gemms bf16}; do
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${cfg}_kt -- python3 scripts/pmc_ops.py $cfg > $O/${cfg}_kt.log 2>&1
  timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${cfg}_fetch -- python3 scripts/pmc_ops.py $cfg > $O/${cfg}_fetch.log 2>&1
  timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${cfg}_write -- python3 scripts/pmc_ops.py $cfg > $O/${cfg}_write.log 2>&1
  timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/${cfg}_mfma -- python3 scripts/pmc_ops.py $cfg > $O/${cfg}_mfma.log 2>&1
  python3 - <<PY
import csv, glob, collections
cfg, O, R = '$cfg', '$O', '$R'
def rows(tag):
    f = glob.glob('%s/%s_%s/**/*counter_collection.csv' % (O, cfg, tag), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ('fetch', 'write', 'mfma'):
    for r in rows(tag):
        agg[(r['Kernel_Name'][:90], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
f = glob.glob('%s/%s_kt/**/*kernel_trace.csv' % (O, cfg), recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        g = str(int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']))
        dur[(r['Kernel_Name'][:90], g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = open('%s/%s_%s_op_counters.csv' % (O, R, cfg), 'w')
out.write('kernel,grid,calls,avg_us,fetch_bytes_x2_gfx950,write_bytes,hbm_bytes,achieved_GBps,mfma_pipe_utilisation\n')
lines = []
for key, c in agg.items():
    kn, g = key
    if not any(s in kn for s in ('pw_', 'pwb_', 'dw_', 'dwb_', 'bn_bwd_apply')):
        continue
    mean = lambda v: sum(v[len(v) // 2:]) / max(1, len(v[len(v) // 2:]))        # second half: warm
    fe = mean(c['FETCH_SIZE']) * 1024 * 2 if c.get('FETCH_SIZE') else 0.0
    wr = mean(c['WRITE_SIZE']) * 1024 if c.get('WRITE_SIZE') else 0.0
    us = mean(dur[key]) if dur.get(key) else 0.0
    mb, bc = c.get('SQ_VALU_MFMA_BUSY_CYCLES'), c.get('SQ_BUSY_CU_CYCLES')
    util = (mean(mb) / mean(bc) / 4) if (mb and bc and mean(bc) > 0) else 0.0
    lines.append((us * len(dur.get(key, [])), '"%s",%s,%d,%.1f,%.0f,%.0f,%.0f,%.0f,%.3f\n' % (
        kn, g, len(dur.get(key, [])), us, fe, wr, fe + wr, (fe + wr) / us / 1e3 if us else 0.0, util)))
for _, l in sorted(lines, reverse=True):
    out.write(l)
out.close()
PY
  rm -rf $O/${cfg}_kt $O/${cfg}_fetch $O/${cfg}_write $O/${cfg}_mfma
done
ls -la $O; for cfg in ${PMC_CFGS:-xception xception769 headline_gemms bf16}; do head -8 $O/${R}_${cfg}_op_counters.csv; done
