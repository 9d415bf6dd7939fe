#!/bin/bash
# GPU-box: rocprofv3 evidence for the default bench workload (and its float variant).
# usage: tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*
# --pmc passes run on their own (kernel-trace only), FETCH_SIZE and WRITE_SIZE separately.
set -e
tag=${1:-r01}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 3 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python3 bench.py --steps 20 --warmup 3 --precision f64 > $out/${tag}_bench_f64.json 2>> $out/${tag}_bench.err
for p in f64 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_$p -- python3 bench.py --steps 5 --warmup 2 --precision $p --no-cpu-baseline --no-extras > $out/${tag}_stats_$p.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch_$p -- python3 bench.py --steps 3 --warmup 1 --precision $p --no-cpu-baseline --no-extras > $out/${tag}_fetch_$p.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write_$p -- python3 bench.py --steps 3 --warmup 1 --precision $p --no-cpu-baseline --no-extras > $out/${tag}_write_$p.log 2>&1
done
python3 tests/tools/role_cycles.py 256 500 0 > $out/${tag}_role_cycles_f64_u1.txt 2>/dev/null
python3 tests/tools/role_cycles.py 256 500 2 > $out/${tag}_role_cycles_f32_u1.txt 2>/dev/null
python3 tests/tools/role_cycles.py 4096 500 2 > $out/${tag}_role_cycles_f32_u4.txt 2>/dev/null
for b in 512 1024 4096; do for p in f64 mixed f32; do python3 bench.py --precision $p --batch $b --steps 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1; done; done > $out/${tag}_bench_batches.jsonl
for p in f64 mixed f32; do python3 bench.py --batch 4096 --frames 7500 --delay 2 --precision $p --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1; done > $out/${tag}_bench_config4.jsonl
# reference model 5 (VocalTractModel5<double,1>): bench, rocprof summary, role cycles, parity report
python3 tests/tools/bench_m5.py > $out/${tag}_bench_m5.json 2> $out/${tag}_bench_m5.err
python3 bench.py --model 5 --steps 10 --warmup 2 > $out/${tag}_bench_model5.json 2> $out/${tag}_bench_model5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_m5 -- python3 bench.py --model 5 --steps 5 --warmup 2 --no-cpu-baseline > $out/${tag}_stats_m5.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch_m5 -- python3 bench.py --model 5 --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_fetch_m5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write_m5 -- python3 bench.py --model 5 --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_write_m5.log 2>&1
python3 tests/tools/role_cycles_m5.py 256 500 > $out/${tag}_role_cycles_m5.txt 2>/dev/null
python3 tests/tools/parity_report_m5.py > $out/${tag}_model5_parity.json 2> $out/${tag}_model5_parity.err
python3 tests/tools/parity_report.py > $out/${tag}_config3_parity.json 2> $out/${tag}_config3_parity.err
python3 tests/tools/bench_aux.py > $out/${tag}_bench_aux.json 2> $out/${tag}_bench_aux.err
echo done
