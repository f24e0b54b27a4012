#!/bin/bash
# Same-box A/B of the RK cycle without stored primitives (GPU box):  tools/rk_prim_free_ab.sh [ENV=value ...]
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for w in hydro_plm_hllc_rk2_256 mhd_wenoz_hlld_rk3_256; do for e in APK_RK_PRIM_FREE=0 APK_RK_PRIM_FREE=1 "$@"; do
 env $e python bench.py --workload $w --no-cpu-baseline --no-copies-base --no-rehearsal --no-other-workloads 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['roofline']['per_kernel_avg_ms']
print('$w $e', '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'K2 %.3f K1 %.3f'%(k['fused_x1'],k['fused_x3']), 'cycle frac %.3f'%d['roofline']['whole_cycle']['frac'])"
done; done; done
