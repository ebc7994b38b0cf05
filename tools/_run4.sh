export TMPDIR=/tmp
python -m pytest tests/test_gpu_soft.py -x -q -m gpu 2>&1 | tail -25
