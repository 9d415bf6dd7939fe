#!/bin/bash
# GPU-box: vector-instruction count per stage, from library variants that skip one stage each (tools/build_variant.sh
# skip<bit> -DGVTM_TUNE_SKIP=<bit>).  usage: tools/collect_skip_counts.sh <tag> <precision> <delay> -> gpurun_out/<tag>_skip_counts.txt
set -e
tag=${1:-skip}; prec=${2:-f32}; delay=${3:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${tag}_skip_counts.txt
: > $out
for v in none 1 2 4 8 16 32 64 128 256 512 1024 2047; do
  if [ $v = none ]; then unset GVTM_LIBRARY; else export GVTM_LIBRARY=gama_tts_amd/lib_variants/libgama_vtm_skip$v.so; fi
  d=gpurun_out/${tag}_skip_$v
  rm -rf $d
  rocprofv3 --kernel-trace --pmc ${GVTM_SKIP_COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE} --output-format csv -d $d -- python3 bench.py --precision $prec --batch 4096 --frames 500 --delay $delay --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $d.log 2>&1
  python3 - "$d" "$v" >> $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "synth_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("skip", sys.argv[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
done
cat $out
