python -m pytest tests/test_model_gpu.py tests/test_dp_rccl_gpu.py tests/test_host_api_gpu.py -x -q 2>&1 | tail -5
python tools/ab_libs.py 2 - tools/lib_prev.bin 2>&1 | cut -c1-60
