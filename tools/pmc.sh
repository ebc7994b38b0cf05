#!/bin/bash
# HBM traffic and issue counters per kernel (separate --pmc passes, per MI355X_MICROARCH.md)
export TMPDIR=/tmp
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" ${PMC_EXTRA:+"$PMC_EXTRA"}; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc/$tag -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --superframes ${SF:-64} > gpurun_out/pmc/$tag.json 2> gpurun_out/pmc/$tag.err
done
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmc/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dvbt::', '').split('<')[0]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in acc:
        for c, v in acc[k].items():
            tot[k][c] = sum(v) / len(v)
for k, d in tot.items():
    if d.get('GRBM_GUI_ACTIVE', 0) > 20000 or 'viterbi' in k:
        print(k, {c: round(v, 1) for c, v in sorted(d.items())})
PY
