export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python tools/shard_dbg.py scan 2>&1 | grep "^hole" > gpurun_out/shard_scan.txt; cat gpurun_out/shard_scan.txt
timeout 500 python tools/sweep5.py 60 5007 > gpurun_out/parity_sweep5a.txt 2>&1; tail -1 gpurun_out/parity_sweep5a.txt
timeout 500 python tools/sweep5.py 40 5009 > gpurun_out/parity_sweep5c.txt 2>&1; tail -1 gpurun_out/parity_sweep5c.txt
