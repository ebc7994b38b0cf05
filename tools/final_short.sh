#!/bin/bash
# round-end evidence, the part that changes with the front-end / tail kernels: tests, smoke, the bench line (timed), rocprofv3 kernel stats of the default command and of the
# one-step-in-flight variant, the pipeline-depth series, BASELINE config 5 at the prescribed noise.  (tools/final.sh adds the SNR sweep, the RS load series, the MFMA
# microbenchmark, the soft-mode profile and the PMC passes: those kernels are unchanged since they were last taken.)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
T0=$(date +%s); timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench.py default run: $(( $(date +%s) - T0 )) s wall"; cut -c1-400 gpurun_out/bench_final.json
rm -rf gpurun_out/prof_final gpurun_out/prof_solo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o r -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_prof.json 2> /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_solo -o r -- python bench.py --pipeline 1 --no-cpu-baseline --no-extras > gpurun_out/bench_prof_solo.json 2> /dev/null
for pd in 1 2 3 4; do timeout 300 python bench.py --pipeline $pd --steps 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'steps_in_flight': $pd, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'verified': d['config']['verified']}))"; done | tee gpurun_out/pipeline_depth.jsonl
rm -rf gpurun_out/prof_cfg5; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg5 -o r -- python $GRAFT_REPO_ROOT/tools/period_prof.py 8 16 3 > /dev/null 2>&1)
timeout 200 python tools/period_prof.py 8 16 5 2>/dev/null | grep "^{" > gpurun_out/config5_walk.jsonl; timeout 200 python tools/period_prof.py 9 16 5 2>/dev/null | grep "^{" >> gpurun_out/config5_walk.jsonl; cat gpurun_out/config5_walk.jsonl | cut -c1-300
