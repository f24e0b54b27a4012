import json,sys,os
sys.path.insert(0,'.')
import bench
r=bench.rehearsal_8gpu_rank("mhd_ppm_hlld_vl2_256", float(sys.argv[1]))
print(os.environ.get("APK_OVERLAP_DC"), "overlapped %.3f sync %.3f eff %.3f / %.3f" % (r["overlapped"]["ms_per_step"], r["synchronous"]["ms_per_step"], r["predicted_weak_scaling_efficiency"], r["predicted_weak_scaling_efficiency_if_no_wire_time_is_hidden"]), r["overlapped"]["pack_unpack_copy_kernels_ms_per_cycle"], r["overlapped"]["one_layer_exchanges_per_cycle"], r["one_layer_message_MB_per_peer"][:1])
