"""Turn the raw output of tools/profile_r02.sh (gpurun_out/r02/) into the committed summaries under profiles/."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r02")
P = os.path.join(ROOT, "profiles")


def run(args, **kw):
    return subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, check=True, **kw).stdout


shutil.copy(os.path.join(O, "stats", "s_kernel_stats.csv"), os.path.join(P, "r02_kernel_stats.csv"))
shutil.copy(os.path.join(O, "bench_plain.json"), os.path.join(P, "r02_bench_mhd_ppm_hlld.json"))
shutil.copy(os.path.join(O, "bench_under_rocprof.json"), os.path.join(P, "r02_bench_under_rocprof.json"))
tmp = os.path.join(O, "traffic_in")
os.makedirs(tmp, exist_ok=True)
shutil.copy(os.path.join(O, "fetch", "s_counter_collection.csv"), os.path.join(tmp, "fetch_counter_collection.csv"))
shutil.copy(os.path.join(O, "write", "s_counter_collection.csv"), os.path.join(tmp, "write_counter_collection.csv"))
with open(os.path.join(P, "r02_hbm_traffic.json"), "w") as f:
    f.write(run(["profiles/pmc_traffic.py", tmp]))
with open(os.path.join(P, "r02_pmc_sq.json"), "w") as f:
    f.write(run(["tools/pmc_csv_summary.py", os.path.join(O, "sq", "s_counter_collection.csv"), "fused", "copy_regions"]))
clk = {"bench_kernels": json.loads(run(["tools/pmc_csv_summary.py", os.path.join(O, "clk", "s_counter_collection.csv"), "fused", "copy_regions"])),
       "ubench_kernels_under_rocprof": json.loads(run(["tools/pmc_csv_summary.py", os.path.join(O, "clk_ubench", "s_counter_collection.csv")])),
       "ubench_rates": [json.loads(l) for l in open(os.path.join(O, "ubench_fp64.jsonl")) if l.startswith("{")],
       "ubench_kernel_ids": {"loop_kernel<0>": "v_fma_f64", "loop_kernel<1>": "v_mul_f64", "loop_kernel<2>": "v_add_f64",
                             "loop_kernel<3>": "v_max_f64", "loop_kernel<4>": "v_cndmask_b32", "loop_kernel<5>": "v_cmp_lt_f64+v_cndmask_b32",
                             "loop_kernel<6>": "v_mov_b32", "loop_kernel<7>": "v_rcp_f64", "loop_kernel<8>": "v_rsq_f64",
                             "loop_kernel<9>": "v_sqrt_f64", "loop_kernel<10>": "v_fma_f32", "loop_kernel<11>": "v_mov_b32 dpp"},
       "how": "effective_clock_GHz = GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 / kernel duration; "
              "ubench_rates: tools/ubench/ubench_fp64.hip, register-only loops of one instruction kind, 8 independent chains per lane"}
with open(os.path.join(P, "r02_clock_and_issue_rate.json"), "w") as f:
    json.dump(clk, f, indent=1)
print("profiles written:", sorted(x for x in os.listdir(P) if x.startswith("r02")))
