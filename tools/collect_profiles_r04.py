"""Turn the raw output of tools/profile_r04.sh (gpurun_out/r04/) into the committed summaries under profiles/."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r04")
P = os.path.join(ROOT, "profiles")


def run(args, **kw):
    return subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, check=True, **kw).stdout


def first_json_line(path):
    with open(path) as f:
        for line in f:
            if line.startswith("{"):
                return line
    raise SystemExit("no JSON line in " + path)


for src, dst in (("bench_plain.json", "r04_bench_mhd_ppm_hlld.json"), ("bench_under_rocprof.json", "r04_bench_under_rocprof.json"),
                 ("bench_wenoz.json", "r04_bench_mhd_wenoz_rk3.json"), ("bench_hydro.json", "r04_bench_hydro_plm_hllc.json"),
                 ("bench_amr_extra.json", "r04_bench_mhd_amr_extra.json")):
    with open(os.path.join(P, dst), "w") as f:
        f.write(first_json_line(os.path.join(O, src)))
shutil.copy(os.path.join(O, "stats", "s_kernel_stats.csv"), os.path.join(P, "r04_kernel_stats.csv"))
shutil.copy(os.path.join(O, "amr_stats", "s_kernel_stats.csv"), os.path.join(P, "r04_amr_blast_mhd_kernel_stats.csv"))
shutil.copy(os.path.join(O, "turb_stats", "s_kernel_stats.csv"), os.path.join(P, "r04_turbulence_wenoz_rk3_kernel_stats.csv"))
tmp = os.path.join(O, "traffic_in")
os.makedirs(tmp, exist_ok=True)
shutil.copy(os.path.join(O, "fetch", "s_counter_collection.csv"), os.path.join(tmp, "fetch_counter_collection.csv"))
shutil.copy(os.path.join(O, "write", "s_counter_collection.csv"), os.path.join(tmp, "write_counter_collection.csv"))
with open(os.path.join(P, "r04_hbm_traffic.json"), "w") as f:
    f.write(run(["profiles/pmc_traffic.py", tmp]))
with open(os.path.join(P, "r04_pmc_sq.json"), "w") as f:
    f.write(run(["tools/pmc_csv_summary.py", os.path.join(O, "sq", "s_counter_collection.csv"), "fused", "copy_regions"]))
with open(os.path.join(P, "r04_pmc_sq_wenoz.json"), "w") as f:
    f.write(run(["tools/pmc_csv_summary.py", os.path.join(O, "sq_wenoz", "s_counter_collection.csv"), "fused", "copy_regions"]))
mix = json.loads(run(["tools/pmc_csv_summary.py", os.path.join(O, "mix", "s_counter_collection.csv"), "fused"]))
other = json.loads(run(["tools/pmc_csv_summary.py", os.path.join(O, "other", "s_counter_collection.csv"), "fused"]))
for k, r in mix.items():
    if r.get("SQ_INSTS_VALU"):
        # SQ_THREAD_CYCLES_VALU counts active lanes per VALU instruction (summed): / 64 / instructions = lane activity
        r["active_lanes_per_valu_inst"] = r["SQ_THREAD_CYCLES_VALU"] / r["SQ_INSTS_VALU"]
        arith = r["SQ_INSTS_VALU_ADD_F64"] + r["SQ_INSTS_VALU_MUL_F64"] + r["SQ_INSTS_VALU_FMA_F64"]
        r["fp64_add_mul_fma_share"] = arith / r["SQ_INSTS_VALU"]
with open(os.path.join(P, "r04_pmc_instruction_mix.json"), "w") as f:
    json.dump({"command": "tools/stage_time.py --gam0 0.5 --fill 2 --dt (one pack of 8 x 128^3, general PPM+HLLD stage with FillDerived + dt)",
               "mix": mix, "other": other,
               "how": "two rocprofv3 --pmc passes; SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 do not include v_max/min/cmp/cndmask/mov/dpp"}, f, indent=1)
clk = {"bench_kernels": json.loads(run(["tools/pmc_csv_summary.py", os.path.join(O, "clk", "s_counter_collection.csv"), "fused", "copy_regions"])),
       "how": "effective_clock_GHz = GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 / kernel duration"}
with open(os.path.join(P, "r04_clock.json"), "w") as f:
    json.dump(clk, f, indent=1)
for name in ("amr_prof.txt", "turb_prof.txt"):
    shutil.copy(os.path.join(O, name), os.path.join(P, "r04_" + name))
shutil.copy(os.path.join(O, "ubench_march_traffic.jsonl"), os.path.join(P, "r04_ubench_march_traffic.jsonl"))
with open(os.path.join(P, "r04_bench_sustained_500_cycles.json"), "w") as f:
    f.write(first_json_line(os.path.join(O, "bench_sustained_500.json")))
print("profiles written:", sorted(x for x in os.listdir(P) if x.startswith("r04")))
