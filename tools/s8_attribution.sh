#!/bin/bash
# cost attribution of symbol8k_kernel: builds the library with the compile-time experiment bits of k_symbol8k.hpp (S8_EXP) into
# gr_dvbt_amd/lib/libdvbt_hip_s8e<bits>.so and prints the bench's stage times for each (on the GPU box: bash tools/s8_attribution.sh run)
cd "$(dirname "$0")/.."
BITS="${BITS:-0 1 3 35 4 8 16 32}"
if [ "$1" = build ]; then
  for e in $BITS; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DS8_EXP=$e ${EXTRA} -o gr_dvbt_amd/lib/libdvbt_hip_s8e$e.so gr_dvbt_amd/csrc/dvbt_hip.hip || exit 1; done
else
  for e in $BITS; do
    python tools/ab_bench.py $PWD/gr_dvbt_amd/lib/libdvbt_hip_s8e$e.so --wrong-output --pipeline 1 --steps 10 --warmup 2 --superframes 16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'s8_exp': $e, 'fft_stage_ms_17sf': d['stage_ms_per_piece']['fft']}))"
  done
fi
