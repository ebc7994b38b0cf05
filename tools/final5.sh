#!/bin/bash
# round-5 evidence: tools/final_short.sh (tests, smoke, the bench line, rocprofv3 kernel stats of the default command and of one step in flight, the pipeline-depth series, BASELINE
# config 5 at the prescribed noise) + the PMC passes on the final code + the randomised sweeps (tools/sweep.py, sweep4.py, sweep5.py)
export TMPDIR=/tmp
bash tools/final_short.sh > gpurun_out/final_short.log 2>&1; tail -12 gpurun_out/final_short.log | cut -c1-300
bash tools/pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log | cut -c1-300
timeout 500 python tools/sweep.py 120 5005 > gpurun_out/parity_sweep.txt 2>&1; tail -1 gpurun_out/parity_sweep.txt
timeout 400 python tools/sweep4.py 60 5006 > gpurun_out/parity_sweep4.txt 2>&1; tail -1 gpurun_out/parity_sweep4.txt
timeout 400 python tools/sweep5.py 60 5007 > gpurun_out/parity_sweep5a.txt 2>&1; tail -1 gpurun_out/parity_sweep5a.txt
timeout 400 python tools/sweep5.py 60 5008 > gpurun_out/parity_sweep5b.txt 2>&1; tail -1 gpurun_out/parity_sweep5b.txt
