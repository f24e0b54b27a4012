cd $GRAFT_REPO_ROOT
for rep in 1 2; do for e in APK_M12F_GRID_WAVES=2 APK_M12F_GRID_WAVES=1; do
 env $e python bench.py --workload hydro_plm_hllc_rk2_256 --no-cpu-baseline --no-copies-base --no-rehearsal --no-other-workloads 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); k=d['roofline']['per_kernel_avg_ms']
print('$e', '%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'K2 %.3f K1 %.3f'%(k['fused_x1'],k['fused_x3']), 'cycle frac %.3f'%d['roofline']['whole_cycle']['frac'])"
done; done
