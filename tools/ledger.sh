#!/bin/bash
# Compile one fused translation unit of the product build with --save-temps and print the per-iteration ledger of the
# stage kernels' main loops:   tools/ledger.sh [extra hipcc flags]   (output dir /tmp/ledger)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/ledger && cd /tmp/ledger
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I"$OLDPWD/include" -fapprox-func -freciprocal-math \
  --save-temps "$@" -c "$OLDPWD/athenapk_amd/csrc/fused_mhd_hlld.hip" -o f.o -Rpass-analysis=kernel-resource-usage 2> res.txt || { tail -30 res.txt; exit 1; }
python "$OLDPWD/tools/isa_loop_ledger.py" fused_mhd_hlld-hip-amdgcn-amd-amdhsa-gfx950.s 'fused_m12f_kernel<2, 3, 5, 2, 1, false, false>' 'fused_m12f_kernel<2, 3, 5, 0, 1, false, false>' \
  'fused_march_kernel<2, 3, 5, 3, false, 0, false>' 'fused_dc3_kernel<2, 5, 1, true>' 'fused_dc3r2_kernel<2, 5, 1, true, false, 1>' --json ledger.json
