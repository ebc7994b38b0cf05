export TMPDIR=/tmp
python -m pytest tests/test_gpu_tps.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -25
