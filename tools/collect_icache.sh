#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for p in mixed f64 f32; do
rocprofv3 --kernel-trace --pmc SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d gpurun_out/ic_$p -- python3 bench.py --precision $p --batch 4096 --frames 500 --delay 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/ic_$p.log 2>&1
python3 - gpurun_out/ic_$p $p <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "synth_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done
