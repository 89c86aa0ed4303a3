mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r03p_pytest_gpu.log 2>&1
tail -15 gpurun_out/r03p_pytest_gpu.log
