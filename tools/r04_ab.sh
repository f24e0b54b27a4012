#!/bin/bash
# Same-box A/B of run-time switches / product-build variants on the whole bench cycle (GPU box):
#   tools/r04_ab.sh "label:ENV=1 ENV2=x" "label2:APK_LIB_PATH=athenapk_amd/libapk_amd_var.so" ...
# prints value, ms per cycle and the per-kernel times of every configuration, twice (interleaved)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "default:" "$@"; do
 label="${cfg%%:*}"; envs="${cfg#*:}"
 env $envs python bench.py --no-cpu-baseline --no-copies-base ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['roofline']['per_kernel_avg_ms']
g=d['roofline'].get('general_stage',{}).get('ms_per_stage',0)
print('%-14s'%'$label', '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'K2 %.3f K1 %.3f DC %.3f general %.3f'%(k['fused_x1'],k['fused_x3'],k['fused_dc_x1'],g))"
done; done
