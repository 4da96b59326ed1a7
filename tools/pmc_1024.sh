#!/bin/bash
# PMC counters of the pitch-1024 scan kernels (GPU box): one rocprofv3 --pmc pass per counter group, never combined with a trace domain.
# usage: tools/pmc_1024.sh <out-dir under gpurun_out> <variant>
set -o pipefail
OUT=gpurun_out/$1; V=$2
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_v${V}_$i" -- python3 tools/ab.py "variant=$V" --rows 4194304 --dim 1024 --rounds 1 --iters 2 > "$OUT/pmc_v${V}_$i.log" 2>&1 || exit 1
  echo "pmc pass $i done" >&2
done
