set -e
mkdir -p gpurun_out/r03a
python bench.py --gpus 2 --single-device --dist-backend gloo --batch 512 --frames 500 --steps 3 --warmup 1 --no-extras --no-end-to-end --no-cpu-baseline > gpurun_out/r03a/bench_2rank.json 2> gpurun_out/r03a/bench_2rank.err || { tail -20 gpurun_out/r03a/bench_2rank.err; exit 1; }
cat gpurun_out/r03a/bench_2rank.json | cut -c1-600
for p in f32 mixed f64; do python tools/ab.py 4096 $p default r02 default r02; done > gpurun_out/r03a/ab_r02.txt 2>&1
cat gpurun_out/r03a/ab_r02.txt
