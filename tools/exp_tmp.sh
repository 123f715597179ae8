python -m pytest tests/test_model_gpu.py -x -q -k overlap 2>&1 | tail -3
python tools/overlap_sweep.py 2
