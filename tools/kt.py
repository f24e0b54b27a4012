"""print the per-kernel average times of a bench line (json on stdin or file arg)"""
import json, sys
d = json.load(open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin)
r = d["roofline"]
k = r["per_kernel_avg_ms"]
print("value %.4e ms/step %.3f | x1 %.3f x2 %.3f x3 %.3f dc %.3f copy %.3f | stage %.3f general %.3f frac %.4f gfrac %.4f" % (
    d["value"], d["ms_per_step"], k["fused_x1"], k["fused_x2"], k["fused_x3"], k["fused_dc_x1"], k["copy_regions"],
    r["stage_ms"], r["general_stage"]["ms_per_stage"], r["frac"], r["general_stage"]["frac"]))
