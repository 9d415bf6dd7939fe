#!/bin/bash
# GPU-box: rocprofv3 evidence for the bench workload (BASELINE configs[3]: 4096 x 7500 frames, SectionDelay 2) in the
# three precisions, plus configs[1] (256 x 500).     usage: tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*
# --pmc passes run on their own (kernel-trace only), FETCH_SIZE and WRITE_SIZE separately (MI355X_MICROARCH.md).
set -e
tag=${1:-r03}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 10 --warmup 2 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "bench done"
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
for p in f32 mixed f64; do
  common="--precision $p --no-cpu-baseline --no-extras --no-end-to-end --no-parity-check"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_$p -- python3 bench.py --steps 3 --warmup 1 $common > $out/${tag}_stats_$p.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch_$p -- python3 bench.py --steps 2 --warmup 1 $common > $out/${tag}_fetch_$p.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write_$p -- python3 bench.py --steps 2 --warmup 1 $common > $out/${tag}_write_$p.log 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out/${tag}_inst_$p -- python3 bench.py --steps 2 --warmup 1 $common > $out/${tag}_inst_$p.log 2>&1
  echo "profiled $p"
done
# configs[1]: 256 x 500 frames, VocalTractModel0 semantics
for p in f32 f64; do
  common="--precision $p --batch 256 --frames 500 --delay 1 --no-cpu-baseline --no-extras --no-end-to-end --no-parity-check"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats256_$p -- python3 bench.py --steps 10 --warmup 2 $common > $out/${tag}_stats256_$p.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch256_$p -- python3 bench.py --steps 3 --warmup 1 $common > $out/${tag}_fetch256_$p.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write256_$p -- python3 bench.py --steps 3 --warmup 1 $common > $out/${tag}_write256_$p.log 2>&1
done
for cfg in "4096 1000 2 2" "4096 1000 1 2" "4096 1000 0 2" "4096 500 2 1" "4096 500 1 1" "4096 500 0 1" "256 500 2 1" "256 500 0 1"; do
  set -- $cfg
  python3 tests/tools/role_cycles.py $cfg > $out/${tag}_role_cycles_b$1_p$3_d$4.txt 2>/dev/null
done
python3 tests/tools/role_cycles_m5.py 256 500 > $out/${tag}_role_cycles_m5_b256.txt 2>/dev/null
python3 bench.py --precision f64 --steps 5 --warmup 1 --no-extras > $out/${tag}_bench_f64.json 2> $out/${tag}_bench_f64.err
python3 bench.py --model 5 --steps 10 --warmup 2 > $out/${tag}_bench_model5.json 2> $out/${tag}_bench_model5.err
python3 bench.py --model 4 --steps 10 --warmup 2 --precision f64 > $out/${tag}_bench_model4.json 2> $out/${tag}_bench_model4.err
python3 tests/tools/parity_report.py > $out/${tag}_config3_parity.json 2> $out/${tag}_config3_parity.err
python3 tests/tools/bench_aux.py > $out/${tag}_bench_aux.json 2> $out/${tag}_bench_aux.err
python3 tests/tools/bench_models.py > $out/${tag}_bench_models.txt 2> $out/${tag}_bench_models.err
echo done
