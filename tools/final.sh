#!/bin/bash
# round-end evidence: tests, smoke, bench line, rocprofv3 kernel stats of the same command, PMC passes
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-400 gpurun_out/bench_final.json
rm -rf gpurun_out/prof_final; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o r -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_prof.json 2> /dev/null
bash tools/pmc.sh > gpurun_out/pmc.log 2>&1; tail -3 gpurun_out/pmc.log | cut -c1-300
