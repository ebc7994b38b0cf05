#!/bin/bash
# HBM traffic and issue counters per kernel (separate --pmc passes, per MI355X_MICROARCH.md).  The bench's pre-scan launches the same
# kernels on a small head of the stream: every kernel's LARGEST dispatch is the one of the timed steps (tools/save_profiles.py keeps that).
export TMPDIR=/tmp
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc/$tag -o p -- python bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-extras --superframes ${SF:-64} > gpurun_out/pmc/$tag.json 2> gpurun_out/pmc/$tag.err
done
python tools/save_profiles.py --pmc-only
