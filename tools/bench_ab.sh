#!/bin/bash
# Same-box A/B of product-build variants (csrc/Makefile `variant`) on the whole bench cycle, run on the GPU box:
#   tools/bench_ab.sh var1 var2 ...   ->  value, ms per cycle, finishing march / x3 sweep / donor-cell march / general stage
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in default "$@"; do
 if [ $v = default ]; then unset APK_LIB_PATH; else export APK_LIB_PATH=athenapk_amd/libapk_amd_$v.so; fi
 python bench.py --no-cpu-baseline --no-copies-base 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d['roofline']['per_kernel_avg_ms']; print('%-10s'%'$v', '%.4e'%d['value'], '%.3f'%d['ms_per_step'], 'K2 %.3f K1 %.3f DC %.3f general %.3f'%(k['fused_x1'],k['fused_x3'],k['fused_dc_x1'], d['roofline']['general_stage']['ms_per_stage']))"
done; done
