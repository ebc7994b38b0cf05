for dbg in 0 1 2 3 7; do DVBT_VITERBI_DBG=$dbg python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg $dbg', d['value'], d['stage_ms_per_segment']['viterbi'])"; done
python -m pytest tests/test_gpu_blocks.py -q -m gpu -k viterbi 2>&1 | tail -3
