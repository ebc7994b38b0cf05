python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for mc in 1792 3000 9000; do for dbg in 0 8; do DVBT_VITERBI_DBG=$dbg DVBT_VITERBI_MAXCHUNK=$mc python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('maxchunk $mc dbg $dbg', d['value'], d['stage_ms_per_segment']['viterbi'])"; done; done
