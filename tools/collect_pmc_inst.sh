#!/bin/bash
# GPU-box: instruction-mix counters of the synthesis kernel (own pass: --pmc with --kernel-trace only).
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
for cfg in "256 f32" "4096 f32" "256 f64"; do
  set -- $cfg
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/${tag}_inst_$1_$2 -- python3 bench.py --precision $2 --batch $1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_inst_$1_$2.log 2>&1
done
echo done
