#!/bin/bash
# cost attribution of viterbi3_kernel (8k QAM64 7/8, 17,475 OFDM symbols = the bench's launch): builds tools/vit_kbench.hip with the
# compile-time experiment bits of k_viterbi3.hpp (V3_EXP) and prints one JSON line per build and chunk size.  Run on the GPU box:
#   bash tools/vit_attribution.sh run > gpurun_out/vit_attribution.jsonl     (build first, here: bash tools/vit_attribution.sh build)
cd "$(dirname "$0")"
BITS="0 1 2 4 8 15 16 31 48"
if [ "$1" = build ]; then
  for e in $BITS; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DV3_EXP=$e -I ../gr_dvbt_amd/csrc -o vit_kbench_e$e vit_kbench.hip || exit 1; done
else
  for e in $BITS; do ./vit_kbench_e$e 17475 8467 2850; done
fi
