#!/bin/bash
# A/B of product-build variants on the north-star stage (run on the GPU box): tools/ab_r03.sh var1 var2 ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; mkdir -p $O
for rep in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then unset APK_LIB_PATH; else export APK_LIB_PATH=athenapk_amd/libapk_amd_$v.so; fi
  for args in "--gam0 0.5 --fill 2 --dt" "--gam0 0.5 --fill 2 --dt --generic"; do
    echo "== $v | $args" | tee -a $O/ab.txt
    python tools/stage_time.py $args --reps 10 2>/dev/null | tail -1 | tee -a $O/ab.txt
  done
done
done
