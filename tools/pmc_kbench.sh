#!/bin/bash
# PMC passes over the standalone tools (fast): vit_kbench builds and the VALU microbenchmark, to compare counters
export TMPDIR=/tmp
out=gpurun_out/pmck; rm -rf $out; mkdir -p $out
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
G2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA"
G3="SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM SQ_WAVE32_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS"
i=0
for g in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  for e in "$@"; do
    rocprofv3 --pmc $g --output-format csv -d $out/g${i}_e$e -o p -- ./tools/vit_kbench_e$e 17475 8467 > $out/g${i}_e$e.log 2>&1
  done
  rocprofv3 --pmc $g --output-format csv -d $out/g${i}_ub -o p -- ./tools/ubench_valu > $out/g${i}_ub.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, re
res = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmck/*/*counter_collection.csv')):
    tag = f.split('/')[-2].split('_', 1)[1]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        name = r['Kernel_Name']
        if 'viterbi3' in name:
            key = tag + ' viterbi3 grid=' + r['Grid_Size']
        elif tag == 'ub' and r['Grid_Size'] == str(2048 * 64) and r['Workgroup_Size'] == '64':
            key = 'ub one-wave W=2 ' + re.sub(r'.*k<\(anonymous enum\)|void |\(.*', '', name)[:40]
        else:
            continue
        res[key].setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k in sorted(res):
    d = {c: max(v) for c, v in res[k].items()}
    print(k, {c: round(v) for c, v in sorted(d.items())})
PY
