#!/bin/bash
# round-6 evidence on the GPU box (one gpurun call, ~30 min): tests (logged), smoke, the bench line, rocprofv3 kernel stats of the default command and of one step in flight, the
# pipeline-depth series, the proof's A/B, BASELINE config 5 at the prescribed noise, the C++ multi-GPU host's kernel + memory-copy trace, the PMC passes, the randomised sweeps, the
# sharded dropout scan.  tools/save_profiles.py r06 copies the summaries into profiles/.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
T0=$(date +%s); timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench.py default run: $(( $(date +%s) - T0 )) s wall"; cut -c1-400 gpurun_out/bench_final.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_short.json 2> /dev/null; cut -c1-200 gpurun_out/bench_short.json
rm -rf gpurun_out/prof_final gpurun_out/prof_solo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o r -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_prof.json 2> /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_solo -o r -- python bench.py --pipeline 1 --no-cpu-baseline --no-extras > gpurun_out/bench_prof_solo.json 2> /dev/null
for pd in 1 2 3 4 5 6; do timeout 300 python bench.py --pipeline $pd --steps 200 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'steps_in_flight': $pd, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'verified': d['config']['verified']}))"; done | tee gpurun_out/pipeline_depth.jsonl
for v in 1 0 -1; do timeout 300 python bench.py --pipeline 1 --steps 100 --no-cpu-baseline --no-extras --viterbi-verify $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'viterbi_verify': $v, 'steps_in_flight': 1, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'verified': d['config']['verified'], 'viterbi_check': d['config'].get('viterbi_check'), 'stage_ms_solo': d['stage_ms_per_piece_solo']}))"; done | tee gpurun_out/verify_ab.jsonl
rm -rf gpurun_out/prof_cfg5; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg5 -o r -- python $GRAFT_REPO_ROOT/tools/period_prof.py 8 16 3 > /dev/null 2>&1)
timeout 200 python tools/period_prof.py 8 16 5 2>/dev/null | grep "^{" > gpurun_out/config5_walk.jsonl; timeout 200 python tools/period_prof.py 9 16 5 2>/dev/null | grep "^{" >> gpurun_out/config5_walk.jsonl; cat gpurun_out/config5_walk.jsonl | cut -c1-300
rm -rf gpurun_out/prof_cpp_lent; timeout 400 python tools/cpp_prof.py lent 16 > gpurun_out/cpp_prof.log 2>&1; tail -2 gpurun_out/cpp_prof.log | cut -c1-300
bash tools/pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log | cut -c1-300
timeout 500 python tools/sweep.py 120 6005 > gpurun_out/parity_sweep.txt 2>&1; tail -1 gpurun_out/parity_sweep.txt
timeout 400 python tools/sweep4.py 60 6006 > gpurun_out/parity_sweep4.txt 2>&1; tail -1 gpurun_out/parity_sweep4.txt
rm -f gpurun_out/parity_sweep5_*.txt
for seed in 6007 6008; do timeout 400 python tools/sweep5.py 60 $seed > gpurun_out/parity_sweep5_$seed.txt 2>&1; tail -1 gpurun_out/parity_sweep5_$seed.txt; done
timeout 400 python tools/shard_dbg.py scan 2>&1 | grep "^hole" > gpurun_out/shard_scan.txt; cat gpurun_out/shard_scan.txt
