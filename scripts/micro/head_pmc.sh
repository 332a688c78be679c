#!/bin/bash
set -uo pipefail
# SQ counters + durations of the fused training head's two passes at the headline shape (scripts/micro/head_bench.py) -> stdout
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
echo "# scripts/micro/head_pmc.sh: 16 x 129x129x24 -> 513x513, 21 classes; rocprofv3 --pmc, three passes; SQ counters are chip totals per launch"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" \
         "SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  for K in head_xpass head_ypass; do
    echo "## $K"
    bash scripts/pmc_kernel.sh "$C" $K -- scripts/micro/head_bench.py
  done
done
echo "## durations (rocprofv3 --kernel-trace)"
bash scripts/ktrace.sh head_ -- scripts/micro/head_bench.py
python3 scripts/micro/head_bench.py 2>/dev/null | tail -3
