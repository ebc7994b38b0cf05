#!/bin/bash
# cost attribution of derot_fft_demod_kernel: builds the library with the compile-time experiment bits of k_symbol.hpp (SYM_EXP) into
# gr_dvbt_amd/lib/libdvbt_hip_sym<bits>.so and prints the bench's stage times for each (run on the GPU box: bash tools/sym_attribution.sh run)
cd "$(dirname "$0")/.."
BITS="${BITS:-0 1 2 4 8 16 31}"
if [ "$1" = build ]; then
  for e in $BITS; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DSYM_EXP=$e -o gr_dvbt_amd/lib/libdvbt_hip_sym$e.so gr_dvbt_amd/csrc/dvbt_hip.hip || exit 1; done
else
  for e in $BITS; do
    DVBT_HIP_LIB=$PWD/gr_dvbt_amd/lib/libdvbt_hip_sym$e.so BENCH_SKIP_VERIFY=1 python bench.py --pipeline 1 --steps 10 --warmup 2 --superframes 16 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'sym_exp': $e, 'fft_stage_ms_17sf': d['stage_ms_per_piece']['fft'], 'stage_ms': d['stage_ms_per_piece']}))"
  done
fi
