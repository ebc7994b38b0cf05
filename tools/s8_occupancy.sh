#!/bin/bash
# VERDICT r04 item 5 ("the symbol kernel with a smaller LDS image so that three workgroups fit a CU"), measured.  What keeps symbol8k_kernel at two 512-thread
# workgroups per CU is not only the 64 KB image: at 128 VGPRs two workgroups own every register of the CU's four SIMDs.  A third workgroup needs <= 80 VGPRs.
#   build: the library with  (a) __launch_bounds__(512, 5 | 6)  = 96 | 80 VGPRs, everything else the product's (correct output: the price of the register cap alone,
#                                 still two workgroups per CU because of the LDS);
#                            (b) S8_EXP=128: half of the image allocated (accesses to the other half fall outside the allocation), the demapper's slow path off --
#                                 wrong output, the kernel's instruction stream otherwise unchanged: with two workgroups per CU the control, with THREE (80 VGPRs) the
#                                 best case of the proposal (the image's other half costs nothing at all here);
#          prints registers / spills / scratch of symbol8k_kernel<false,false> per variant (compile-time facts, no GPU needed)
#   run (GPU box): the bench's one-step-in-flight stage time of the symbol kernel (HIP events) per variant, 17 superframes of 8k QAM64 7/8
cd "$(dirname "$0")/.."
VARIANTS="4:0:2 5:0:2 6:0:2 4:128:2 6:128:3"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    IFS=: read mw e wg <<< "$v"; n=mw${mw}_e${e}_wg${wg}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DS8_MIN_WAVES=$mw -DS8_EXP=$e -DS8_WG_PER_CU=$wg \
      -Rpass-analysis=kernel-resource-usage -o gr_dvbt_amd/lib/libdvbt_hip_s8o_$n.so gr_dvbt_amd/csrc/dvbt_hip.hip 2> /tmp/s8o_$n.txt || exit 1
    python - "$n" /tmp/s8o_$n.txt <<'PY'
import json, re, sys
t = open(sys.argv[2]).read()
i = t.find("Function Name: _ZN4dvbt15symbol8k_kernelILb0ELb0EE")
blk = t[i:i + 3000]
g = lambda k: int(re.search(k + r": (\d+)", blk).group(1))
print(json.dumps({"variant": sys.argv[1], "vgprs": g("VGPRs"), "vgpr_spills": g("VGPRs Spill"), "scratch_bytes_per_lane": g(r"ScratchSize \[bytes/lane\]"), "waves_per_simd_by_registers": g(r"Occupancy \[waves/SIMD\]")}))
PY
  done
else
  for v in $VARIANTS; do
    IFS=: read mw e wg <<< "$v"; n=mw${mw}_e${e}_wg${wg}
    W=""; [ "$e" != 0 ] && W="--wrong-output"
    timeout 90 python tools/ab_bench.py $PWD/gr_dvbt_amd/lib/libdvbt_hip_s8o_$n.so $W --pipeline 1 --steps 10 --warmup 2 --superframes 16 --no-cpu-baseline --no-extras 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'variant': '$n', 'launch_bounds_waves_per_simd': $mw, 's8_exp': $e, 'workgroups_per_cu': $wg, 'symbol_kernel_ms_17sf': d['stage_ms_per_piece_solo']['fft'], 'step_ms': d['ms_per_step'], 'verified': d['config'].get('verified')}))" \
      || echo "{\"variant\": \"$n\", \"error\": \"run failed or timed out\"}"
  done
fi
