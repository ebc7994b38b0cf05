#!/bin/bash
# round-5 evidence: tools/final_short.sh (tests, smoke, the bench line, rocprofv3 kernel stats of the default command and of one step in flight, the pipeline-depth series, BASELINE
# config 5 at the prescribed noise) + the randomised sweeps (tools/sweep.py, sweep4.py, sweep5.py) + the sharded dropout scan; `pmc` as first argument adds the PMC passes
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/final_short.sh > gpurun_out/final_short.log 2>&1; tail -12 gpurun_out/final_short.log | cut -c1-300
if [ "$1" = "pmc" ]; then bash tools/pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log | cut -c1-300; fi
if [ "$1" = "pmc" ]; then timeout 500 python tools/sweep.py 120 5005 > gpurun_out/parity_sweep.txt 2>&1; tail -1 gpurun_out/parity_sweep.txt; fi
timeout 400 python tools/sweep4.py 60 5006 > gpurun_out/parity_sweep4.txt 2>&1; tail -1 gpurun_out/parity_sweep4.txt
for seed in 5007 5008 5009 5010; do timeout 400 python tools/sweep5.py 60 $seed > gpurun_out/parity_sweep5_$seed.txt 2>&1; tail -1 gpurun_out/parity_sweep5_$seed.txt; done
timeout 400 python tools/shard_dbg.py scan 2>&1 | grep "^hole" > gpurun_out/shard_scan.txt; cat gpurun_out/shard_scan.txt
