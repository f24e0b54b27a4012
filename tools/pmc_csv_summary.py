"""Per-kernel summary of a rocprofv3 `--pmc ... --kernel-trace --output-format csv` counter_collection.csv:
calls, average duration, average of every counter per dispatch (summed over its instances) and a few
derived figures (VALU instructions per wave; effective shader clock from GRBM_GUI_ACTIVE, which rocprofv3
sums over the 8 XCDs).   python tools/pmc_csv_summary.py counter_collection.csv [substr ...] > out.json"""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(apk::PackView.*", "", name)
    name = re.sub(r"\(apk_copy_region.*", "", name)
    return name.replace("void apk::", "").replace("(anonymous namespace)::", "").replace("apk::", "")


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    disp = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            d = disp.setdefault(r["Dispatch_Id"], {"name": short(r["Kernel_Name"]), "ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"]),
                                                   "vgpr": int(r["VGPR_Count"]), "scratch": int(r["Scratch_Size"]), "lds": int(r["LDS_Block_Size"]),
                                                   "ctr": collections.defaultdict(float)})
            d["ctr"][r["Counter_Name"]] += float(r["Counter_Value"])
    agg = {}
    for d in disp.values():
        a = agg.setdefault(d["name"], {"calls": 0, "ns": 0.0, "vgpr": d["vgpr"], "scratch_bytes_per_lane": d["scratch"], "lds_bytes": d["lds"],
                                       "ctr": collections.defaultdict(float)})
        a["calls"] += 1
        a["ns"] += d["ns"]
        for k, v in d["ctr"].items():
            a["ctr"][k] += v
    out = {}
    for name, a in sorted(agg.items(), key=lambda x: -x[1]["ns"]):
        if pats and not any(p in name for p in pats):
            continue
        if "rocclr" in name or "at::native" in name:
            continue
        # (rocprofv3's VGPR_Count column is the ARCH-VGPR granule count of the dispatch packet, not the kernel's register
        # budget, and LDS_Block_Size its static LDS only -- the marches allocate their rings dynamically: the registers,
        # spills and LDS of every kernel are in profiles/rNN_kernel_resources.txt, from the compiler's own remarks)
        rec = {"calls": a["calls"], "avg_us": a["ns"] / a["calls"] / 1e3, "scratch_bytes_per_lane": a["scratch_bytes_per_lane"],
               "resources": "profiles/r06_kernel_resources.txt"}
        c = {k: v / a["calls"] for k, v in a["ctr"].items()}
        rec.update(c)
        if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES"):
            rec["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        if "SQ_ACTIVE_INST_VALU" in c:
            # quad-cycles of VALU issue summed over waves -> seconds of a fully busy chip (1024 SIMDs)
            rec["valu_busy_ms_at_2.06GHz"] = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2.06e9 * 1e3
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if k in c:
                    rec[k + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
        if "GRBM_GUI_ACTIVE" in c:
            rec["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (a["ns"] / a["calls"])
        out[name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
