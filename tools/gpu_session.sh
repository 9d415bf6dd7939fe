set -e
mkdir -p gpurun_out/r03b
python -m pytest tests -m gpu -x -q -k "not f32" > gpurun_out/r03b/pytest2.log 2>&1 || { tail -40 gpurun_out/r03b/pytest2.log; exit 1; }
tail -2 gpurun_out/r03b/pytest2.log
for p in mixed f64; do python tools/ab.py 4096 $p default nogate default nogate; done > gpurun_out/r03b/ab_gate.txt 2>&1
cat gpurun_out/r03b/ab_gate.txt
for p in f64 mixed; do for c in "" "--voiced-only"; do
python bench.py --precision $p --steps 4 --warmup 1 --no-extras --no-end-to-end --no-cpu-baseline $c > gpurun_out/r03b/bench_${p}_gate${c}.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03b/bench_${p}_gate${c}.json"))
print("$p", "$c", "kernel %.2f ms  %.3f G" % (d["roofline"]["kernel_ms"], d["value"]/1e9), d["parity_check"]["max_err"])
PY
done; done
