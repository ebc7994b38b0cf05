#!/bin/bash
# round-end evidence: tests, smoke, bench line, rocprofv3 kernel stats of the same command (and of the one-step-in-flight variant), PMC passes
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-400 gpurun_out/bench_final.json
rm -rf gpurun_out/prof_final gpurun_out/prof_solo
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o r -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_prof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_solo -o r -- python bench.py --pipeline 1 --no-cpu-baseline --no-extras > gpurun_out/bench_prof_solo.json 2> /dev/null
for pd in 1 2 3 4; do python bench.py --pipeline $pd --steps 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'steps_in_flight': $pd, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'verified': d['config']['verified']}))"; done | tee gpurun_out/pipeline_depth.jsonl
rm -rf gpurun_out/prof_cfg5; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg5 -o r -- python $GRAFT_REPO_ROOT/tools/period_prof.py 8 16 3 > /dev/null 2>&1)
python tools/period_prof.py 8 16 5 2>/dev/null | grep "^{" > gpurun_out/config5_walk.jsonl; python tools/period_prof.py 9 16 5 2>/dev/null | grep "^{" >> gpurun_out/config5_walk.jsonl; cat gpurun_out/config5_walk.jsonl
python tools/snr_sweep.py 3 > gpurun_out/snr_sweep.jsonl 2>/dev/null; wc -l gpurun_out/snr_sweep.jsonl
python tools/rs_load.py > gpurun_out/rs_load.jsonl 2>/dev/null; wc -l gpurun_out/rs_load.jsonl
[ -x tools/ubench_mfma ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma tools/ubench_mfma.hip 2>/dev/null
./tools/ubench_mfma > gpurun_out/ubench_mfma.json 2>/dev/null
rm -rf gpurun_out/prof_soft; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_soft -o r -- python tools/soft_prof.py > /dev/null 2>&1
python tools/soft_gain.py > gpurun_out/soft_gain.jsonl 2>/dev/null; wc -l gpurun_out/soft_gain.jsonl
bash tools/pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log | cut -c1-300
