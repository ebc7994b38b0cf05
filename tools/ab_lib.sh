python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['stage_ms_per_segment'])"
for v in "$@"; do DVBT_HIP_LIB=$PWD/gr_dvbt_amd/lib/libdvbt_hip_$v.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['stage_ms_per_segment'])"; done
