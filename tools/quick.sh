python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_segment'])"
