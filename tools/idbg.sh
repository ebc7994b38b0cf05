python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for d in 0 2; do DVBT_INNER_DBG=$d python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inner dbg $d', d['value'], d['stage_ms_per_segment']['inner'])"; done
