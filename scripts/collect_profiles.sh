#!/bin/bash
# run on the GPU box: bench line + rocprofv3 kernel stats + HBM / MFMA counters (separate passes).
#   collect_profiles.sh r03                                   the headline config (BASELINE configs[1])
#   collect_profiles.sh r03 xception --model xception --batch 4 ...      another config: files get the tag in their names,
#                                                                        the roofline-traffic json is headline-only
# (no `set -e`: a pass that fails must not lose the others)
set -uo pipefail
set -x
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
R0=${1:-r01}
TAG=${2:-}
if [ -n "$TAG" ]; then shift 2; R=${R0}_$TAG; else shift $(( $# > 0 ? 1 : 0 )); R=$R0; fi
ARGS="$* --no-other-configs"
MFMA_INSTS=SQ_INSTS_VALU_MFMA_F32
case " $ARGS " in *" bf16 "*) MFMA_INSTS=;; esac      # (pipe utilisation = busy cycles / CU busy cycles needs no instruction count)
O=gpurun_out/profiles_$R
mkdir -p $O
if [ -z "$TAG" ]; then
  python3 bench.py > $O/bench_$R.log 2>&1
else
  python3 bench.py $ARGS --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$R.log 2>&1
fi
grep '"metric"' $O/bench_$R.log > $O/bench_$R.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline > $O/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES $MFMA_INSTS --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-graph > $O/pmc_mfma.log 2>&1
cp $O/kt/*/*kernel_stats.csv $O/${R}_kernel_stats.csv
# steps in the trace = launches of the once-per-step loss kernel (warm-up + timed + the bench's extra probe steps)
STEPS=$(python3 - <<PY
import csv
rows = list(csv.DictReader(open('$O/${R}_kernel_stats.csv')))
n = [r['Calls'] for r in rows if 'head_xpass_kernel' in r['Name']] or [r['Calls'] for r in rows if 'head_kernel' in r['Name']]
print(n[0] if n else 1)
PY
)
python3 scripts/prof_summary.py $O/kt $STEPS 40 > $O/${R}_kernel_summary.txt
python3 - <<PY
import csv, glob
out = open('$O/${R}_hbm_counters_dw.csv', 'w')
out.write('counter,kernel,grid,calls,mean_value_KB\n')
for tag in ('pmc_fetch', 'pmc_write'):
    f = glob.glob('$O/%s/**/*counter_collection.csv' % tag, recursive=True)
    if not f: continue
    agg = {}
    for r in csv.DictReader(open(f[0])):
        if 'dw_' not in r['Kernel_Name'] and 'dwb_' not in r['Kernel_Name']: continue
        k = (r['Counter_Name'], r['Kernel_Name'][:60], r['Grid_Size'])
        agg.setdefault(k, []).append(float(r['Counter_Value']))
    for (c, kn, g), v in sorted(agg.items()):
        out.write('%s,"%s",%s,%d,%.1f\n' % (c, kn, g, len(v), sum(v) / len(v)))
out.close()
# per-launch HBM traffic of the roofline kernel (guide: FETCH_SIZE x2 on gfx950 for 16-B/lane reads, WRITE_SIZE exact; unit KB)
import json
t = {}
var = {}
for line in open('$O/${R}_hbm_counters_dw.csv').read().splitlines()[1:]:
    c, rest = line.split(',', 1)
    kn = rest.split('"')[1]
    val = float(rest.rsplit(',', 1)[1])
    # <1> = BatchNorm prologue without activation: the launch of aspp rate 18 inside the MobileNetV2 training step, the one
    # bench.py times (<0>: the rate-12 launch; <2>: bench.py's N=256 streaming variant)
    if 'dw_fwd_lattice2<1>' in kn:
        t[c] = val
    if 'dw_fwd_lattice2' in kn:
        var.setdefault(kn.split('(')[0].replace('void ', ''), {})[c] = val
if 'FETCH_SIZE' in t and 'WRITE_SIZE' in t:
    import hashlib, os
    def blob(f):
        data = open(os.path.join('tf-keras-deeplabv3p-model-set_amd', f), 'rb').read()
        return hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()[:12]
    json.dump({'kernel': 'dw_fwd_lattice2', 'source_hash': blob('csrc/dwconv.hip') + '+' + blob('csrc/dw_tuned.h'),
               'fetch_size_kb_raw': t['FETCH_SIZE'], 'write_size_kb': t['WRITE_SIZE'],
               'traffic_bytes': int((2 * t['FETCH_SIZE'] + t['WRITE_SIZE']) * 1024),
               'variants_kb_raw': var,
               'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, eager launches; FETCH_SIZE doubled (gfx950)'},
              open('$O/${R}_roofline_traffic.json', 'w'))
PY
python3 - <<PY
# MFMA utilisation of the pointwise GEMM kernels: busy cycles of the matrix pipe / busy CU cycles (x4: 4 SIMDs per CU
# are counted in SQ_VALU_MFMA_BUSY_CYCLES), and MFMA instructions issued per launch
import csv, glob, collections
f = glob.glob('$O/pmc_mfma/**/*counter_collection.csv', recursive=True)
out = open('$O/${R}_mfma_counters_gemm.csv', 'w')
out.write('kernel,grid,calls,mfma_busy_cycles,busy_cu_cycles,mfma_insts,mfma_pipe_utilisation\n')
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        kn = r['Kernel_Name']
        if 'pw_' not in kn and 'pwb_' not in kn: continue
        agg[(kn[:70], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    rows = []
    for (kn, g), c in agg.items():
        m = c.get('SQ_VALU_MFMA_BUSY_CYCLES', [0]); b = c.get('SQ_BUSY_CU_CYCLES', [0]); i = c.get('$MFMA_INSTS' or 'none', [0])
        mm, bb, ii = sum(m) / len(m), sum(b) / len(b), sum(i) / len(i)
        rows.append((mm, '"%s",%s,%d,%.0f,%.0f,%.0f,%.3f\n' % (kn, g, len(m), mm, bb, ii, mm / bb / 4 if bb else 0)))
    for _, line in sorted(rows, reverse=True)[:40]:
        out.write(line)
out.close()
PY
rm -rf $O/kt $O/pmc_fetch $O/pmc_write $O/pmc_mfma
ls -la $O
