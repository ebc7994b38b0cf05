export TMPDIR=/tmp
python -m pytest tests/test_gpu_rccl.py -x -q -m gpu -k cpp_host 2>&1 | tail -30
