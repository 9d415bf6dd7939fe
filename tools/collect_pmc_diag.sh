#!/bin/bash
# GPU-box: instruction-cache and LDS counters of the synthesis kernel (own passes: --pmc with --kernel-trace only).
# usage: tools/collect_pmc_diag.sh <tag>
set -e
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "4096 500 1 f32" "256 500 1 f32" "4096 500 1 f64"; do
  set -- $cfg
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/${tag}_ic_$1_$4 -- python3 bench.py --precision $4 --batch $1 --frames $2 --delay $3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_ic_$1_$4.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/${tag}_lds_$1_$4 -- python3 bench.py --precision $4 --batch $1 --frames $2 --delay $3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/${tag}_lds_$1_$4.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, sys
for d in sorted(glob.glob("gpurun_out/*_ic_*/") + glob.glob("gpurun_out/*_lds_*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "synth_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: sum(v) / len(v) for k, v in acc.items()})
PY
