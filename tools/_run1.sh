export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python tools/stream_bench.py > gpurun_out/r4a/stream_bench.jsonl 2>&1; cat gpurun_out/r4a/stream_bench.jsonl
python tools/period_prof.py 8 16 5 > gpurun_out/r4a/period8.json 2>&1; cat gpurun_out/r4a/period8.json
python tools/period_prof.py 9 16 5 > gpurun_out/r4a/period9.json 2>&1; cat gpurun_out/r4a/period9.json
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4a/prof8 -o r -- python $GRAFT_REPO_ROOT/tools/period_prof.py 8 16 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r4a/prof8 -name "*kernel_stats*" | head -1 | xargs head -40
