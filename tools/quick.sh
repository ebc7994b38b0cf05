python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_segment'])"
export TMPDIR=/tmp; rm -rf gpurun_out/prof_q; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -o q -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_q/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r["Name"].split("(")[0].replace("void ","").replace("dvbt::","")[:30].ljust(30), r["Calls"].rjust(4), f"{float(r['AverageNs'])/1e3:10.1f} us", r["Percentage"])
PY
