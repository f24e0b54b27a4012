"""Turn the raw output of tools/profile_r06.sh (gpurun_out/r06/) into the committed summaries under profiles/."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r06")
P = os.path.join(ROOT, "profiles")


def run(args, **kw):
    return subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, check=True, **kw).stdout


def first_json_line(path):
    with open(path) as f:
        for line in f:
            if line.startswith("{"):
                return line
    raise SystemExit("no JSON line in " + path)


for src, dst in (("bench_plain.json", "r06_bench_mhd_ppm_hlld.json"), ("bench_under_rocprof.json", "r06_bench_under_rocprof.json"),
                 ("bench_wenoz.json", "r06_bench_mhd_wenoz_rk3.json"), ("bench_hydro.json", "r06_bench_hydro_plm_hllc.json"),
                 ("bench_hydro_under_rocprof.json", "r06_bench_hydro_under_rocprof.json"),
                 ("bench_amr_extra.json", "r06_bench_mhd_amr_extra.json")):
    with open(os.path.join(P, dst), "w") as f:
        f.write(first_json_line(os.path.join(O, src)))
shutil.copy(os.path.join(O, "stats", "s_kernel_stats.csv"), os.path.join(P, "r06_kernel_stats.csv"))
shutil.copy(os.path.join(O, "hydro_stats", "s_kernel_stats.csv"), os.path.join(P, "r06_hydro_plm_hllc_kernel_stats.csv"))
shutil.copy(os.path.join(O, "amr_stats", "s_kernel_stats.csv"), os.path.join(P, "r06_amr_blast_mhd_kernel_stats.csv"))
shutil.copy(os.path.join(O, "turb_stats", "s_kernel_stats.csv"), os.path.join(P, "r06_turbulence_wenoz_rk3_kernel_stats.csv"))
for tag, dst in (("", "r06_hbm_traffic.json"), ("_hydro", "r06_hbm_traffic_hydro.json")):
    tmp = os.path.join(O, "traffic_in" + tag)
    os.makedirs(tmp, exist_ok=True)
    shutil.copy(os.path.join(O, "fetch" + tag, "s_counter_collection.csv"), os.path.join(tmp, "fetch_counter_collection.csv"))
    shutil.copy(os.path.join(O, "write" + tag, "s_counter_collection.csv"), os.path.join(tmp, "write_counter_collection.csv"))
    # (hydro: the calibration kernel is the hydro time-step kernel of the same run, all 5 primitives of 8 x 128^3 cells)
    extra = ["--calib-kernel", "min_dt_kernel<1>", "--known-bytes", str(8 * 5 * 128 ** 3 * 8.0)] if tag else []
    with open(os.path.join(P, dst), "w") as f:
        f.write(run(["profiles/pmc_traffic.py", tmp] + extra))
for tag, dst in (("sq", "r06_pmc_sq.json"), ("sq_wenoz", "r06_pmc_sq_wenoz.json"), ("sq_hydro", "r06_pmc_sq_hydro.json")):
    with open(os.path.join(P, dst), "w") as f:
        f.write(run(["tools/pmc_csv_summary.py", os.path.join(O, tag, "s_counter_collection.csv"), "fused", "copy_regions"]))


def mix_of(tag):
    mix = json.loads(run(["tools/pmc_csv_summary.py", os.path.join(O, tag, "s_counter_collection.csv"), "fused"]))
    for k, r in mix.items():
        if r.get("SQ_INSTS_VALU"):
            # SQ_THREAD_CYCLES_VALU counts live lanes per PASS of a VALU instruction through the pipe, summed: an fp64
            # transcendental makes four passes, everything else one -- live lanes per instruction = the sum over
            # (instructions + 3 x transcendentals), which cannot exceed 64 (round-5 review: the plain quotient gave 68.9
            # for the hydro march, whose instructions are 6 % transcendentals)
            r["active_lanes_per_valu_inst"] = r["SQ_THREAD_CYCLES_VALU"] / (r["SQ_INSTS_VALU"] + 3.0 * r.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
            r["active_lanes_per_valu_inst_uncorrected"] = r["SQ_THREAD_CYCLES_VALU"] / r["SQ_INSTS_VALU"]
            arith = r["SQ_INSTS_VALU_ADD_F64"] + r["SQ_INSTS_VALU_MUL_F64"] + r["SQ_INSTS_VALU_FMA_F64"]
            r["fp64_add_mul_fma_share"] = arith / r["SQ_INSTS_VALU"]
    return mix


other = json.loads(run(["tools/pmc_csv_summary.py", os.path.join(O, "other", "s_counter_collection.csv"), "fused"]))
with open(os.path.join(P, "r06_pmc_instruction_mix.json"), "w") as f:
    json.dump({"command": "tools/stage_time.py --gam0 0.5 --fill 2 --dt (one pack of 8 x 128^3, general PPM+HLLD stage with FillDerived + dt)",
               "mix": mix_of("mix"), "other": other,
               "how": "two rocprofv3 --pmc passes; SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 do not include v_max/min/cmp/cndmask/mov/dpp"}, f, indent=1)
with open(os.path.join(P, "r06_pmc_instruction_mix_hydro.json"), "w") as f:
    json.dump({"command": "bench.py --workload hydro_plm_hllc_rk2_256 (RK2: two single-march PLM+HLLC stages per cycle, fused_s3_kernel)",
               "mix": mix_of("mix_hydro")}, f, indent=1)
clk = {"bench_kernels": json.loads(run(["tools/pmc_csv_summary.py", os.path.join(O, "clk", "s_counter_collection.csv"), "fused", "copy_regions"])),
       "how": "effective_clock_GHz = GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 / kernel duration"}
with open(os.path.join(P, "r06_clock.json"), "w") as f:
    json.dump(clk, f, indent=1)
with open(os.path.join(P, "r06_pmc_instruction_mix_scheme_floor.json"), "w") as f:
    mixf = json.loads(run(["tools/pmc_csv_summary.py", os.path.join(O, "mix_floor", "s_counter_collection.csv"), "scheme_floor"]))
    for k, r in mixf.items():
        if r.get("SQ_INSTS_VALU"):
            # 2048 waves x 256 sweep steps per launch: vector instructions per wave-step (9 PPM reconstructions + 1 HLLD solve)
            r["valu_insts_per_wave_step"] = r["SQ_INSTS_VALU"] / r["SQ_WAVES"] / 256.0
            r["active_lanes_per_valu_inst"] = r["SQ_THREAD_CYCLES_VALU"] / (r["SQ_INSTS_VALU"] + 3.0 * r.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
    json.dump({"command": "tools/floor_run.py (bench.scheme_floor: csrc/bench_floor.hip, 256 sweep steps per lane, two waves per SIMD)", "mix": mixf,
               "floor": json.loads(first_json_line(os.path.join(O, "floor.txt")))}, f, indent=1)
shutil.copy(os.path.join(O, "reh_stats", "s_kernel_stats.csv"), os.path.join(P, "r06_rehearsal_8gpu_rank_kernel_stats.csv"))
shutil.copy(os.path.join(O, "reh_packed_stats", "s_kernel_stats.csv"), os.path.join(P, "r06_rehearsal_8gpu_rank_x1_packed_kernel_stats.csv"))
shutil.copy(os.path.join(O, "ot_stats", "s_kernel_stats.csv"), os.path.join(P, "r06_orszag_tang_512x512x4_kernel_stats.csv"))
for name in ("reh_prof.txt", "reh_packed_prof.txt", "ot_prof.txt"):
    shutil.copy(os.path.join(O, name), os.path.join(P, "r06_" + name))
for name in ("amr_prof.txt", "turb_prof.txt"):
    shutil.copy(os.path.join(O, name), os.path.join(P, "r06_" + name))
with open(os.path.join(P, "r06_kernel_resources.txt"), "w") as f:
    for tu in ("fused_mhd_hlld.hip", "fused_euler_hllc.hip"):
        f.write("## " + tu + "\n" + run(["tools/kernel_resources.py", tu]) + "\n")
print("profiles written:", sorted(x for x in os.listdir(P) if x.startswith("r06")))
