#!/bin/bash
# rocprofv3 passes of the bench command (GPU box): kernel trace + stats, then PMC counters in separate runs
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with trace domains other than kernel-trace).
# usage: tools/profile_round.sh <out-dir under gpurun_out> [bench args...]
set -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-regimes $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 20 --warmup 5 > "$OUT/bench_profiled.json" 2> "$OUT/trace.err" || exit 1
echo "trace done" >&2
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc$i" -- $BENCH --steps 3 --warmup 1 > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" || exit 1
  echo "pmc pass $i done" >&2
done
python3 bench.py $* --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench_line.err" || exit 1
python3 tools/profile_summary.py "$OUT"
