set -e
mkdir -p gpurun_out/r03c
timeout -k 10 600 python -m pytest tests/test_gpu_model5.py tests/test_gpu_overrun.py tests/test_gpu_soak.py -x -q > gpurun_out/r03c/pytest_m5.log 2>&1 || { tail -40 gpurun_out/r03c/pytest_m5.log; exit 1; }
tail -2 gpurun_out/r03c/pytest_m5.log
for b in 256 512 1024 2048; do timeout -k 10 120 python tests/tools/role_cycles_m5.py $b 250; done > gpurun_out/r03c/m5_roles.txt 2>&1
grep -E "kernel|w[0-9]" gpurun_out/r03c/m5_roles.txt
