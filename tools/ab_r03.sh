#!/bin/bash
# A/B of product-build variants (csrc/Makefile `variant`) on one fused stage of 8 x 128^3, run on the GPU box:
#   [AB_ARGS="--recon wenoz --gam0 0.5 --fill 2 --dt"] tools/ab_r03.sh var1 var2 ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
args="${AB_ARGS:---gam0 0.5 --fill 2 --dt}"
for rep in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then unset APK_LIB_PATH; else export APK_LIB_PATH=athenapk_amd/libapk_amd_$v.so; fi
  echo "== $v | $args" | tee -a $O/ab.txt
  python tools/stage_time.py $args --reps 10 2>/dev/null | tail -2 | tee -a $O/ab.txt
done
done
