export TMPDIR=/tmp
mkdir -p gpurun_out/r4f
python -m pytest tests/test_gpu_blocks.py -x -q -m gpu -k wave_peak 2>&1 | tail -15; python -m pytest tests/test_gpu_config5.py tests/test_gpu_channel.py tests/test_gpu_reacquire.py -x -q -m gpu 2>&1 | tail -5
python tools/period_prof.py 8 16 5 > gpurun_out/r4f/period8.json 2>&1; cat gpurun_out/r4f/period8.json
python tools/period_prof.py 9 16 5 > gpurun_out/r4f/period9.json 2>&1; cat gpurun_out/r4f/period9.json
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4f/prof8 -o r -- python $GRAFT_REPO_ROOT/tools/period_prof.py 8 16 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4f/prof8/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(f"{r['Name'][:50]:50s} n={r['Calls']:>5s} tot_ms={int(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.2f}")
PY
