export TMPDIR=/tmp
mkdir -p gpurun_out/r4h
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r4h/bench.json 2> gpurun_out/r4h/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4h/bench.json').read())
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}, d['config']['verified'])
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','solo_launch_ms','traffic')})
for k,v in d['extra_workloads'].items():
    if 'value' in v: print(k, v['value'], v.get('x_realtime'))
    else: print(k, {kk:(vv['value'],vv['x_realtime'],vv['ms_per_run']) for kk,vv in v.items()})
print('stream_abi', [(x['symbols_per_call'], x['value']) for x in d['stream_abi']])
print('per_block', {k:v['value'] for k,v in d['per_block_abi']['variants'].items()}, d['per_block_abi']['host_pointer_ceiling'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores'))
PY
tail -3 gpurun_out/r4h/bench.err
