set -e
mkdir -p gpurun_out/r03c
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_model5.py tests/test_gpu_dropin.py -x -q > gpurun_out/r03c/pytest_m5s.log 2>&1 || { tail -60 gpurun_out/r03c/pytest_m5s.log; exit 1; }
tail -2 gpurun_out/r03c/pytest_m5s.log
