#!/bin/bash
# Round 5, GPU call 24: boundary-plane fluxes beside the stage kernels, medians
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r05_ab24.txt
for v in 1 "" 1 "" 1 ""; do
  if [ -n "$v" ]; then export APK_AMR_PLANES_INLINE=1; else unset APK_AMR_PLANES_INLINE; fi
  echo "APK_AMR_PLANES_INLINE=${v:-unset} $(python tools/amr_ab.py 7 2>&1 | grep median)" >> gpurun_out/r05_ab24.txt
done
cat gpurun_out/r05_ab24.txt
