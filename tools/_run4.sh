export TMPDIR=/tmp
python -m pytest tests/test_gpu_blocks.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -25
