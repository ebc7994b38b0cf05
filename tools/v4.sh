python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for k in 4 3; do DVBT_VITERBI_KERNEL=$k python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel $k', d['value'], d['stage_ms_per_segment'])"; done
for dbg in 1 2 3; do DVBT_VITERBI_DBG=$dbg python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg $dbg', d['value'], d['stage_ms_per_segment']['viterbi'])"; done
