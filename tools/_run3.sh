export TMPDIR=/tmp
mkdir -p gpurun_out/r4g
python -m pytest tests/test_gpu_rccl.py tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -15
python tools/stream_bench.py > gpurun_out/r4g/stream_bench.jsonl 2>&1; cat gpurun_out/r4g/stream_bench.jsonl
