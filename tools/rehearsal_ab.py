import json,sys,os
sys.path.insert(0,'.')
import bench
r=bench.rehearsal_8gpu_rank("mhd_ppm_hlld_vl2_256", float(sys.argv[1]))
print(os.environ.get("APK_OVERLAP_DC"), "overlapped %.3f sync %.3f eff %.3f / %.3f" % (r["overlapped"]["ms_per_step"], r["synchronous"]["ms_per_step"], r["predicted_weak_scaling_efficiency_if_wire_hidden"], r["predicted_weak_scaling_efficiency_if_wire_fully_exposed"]), r["overlapped"]["overlapped_exchanges_per_cycle"])
