"""HBM traffic per kernel launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected
in separate runs as MI355X_MICROARCH.md prescribes: `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
together with `--kernel-trace` only).

On gfx950 FETCH_SIZE under-reports streaming reads (the guide measures exactly 1/2 for
16-B/lane loads and says other widths must be calibrated on a known byte count in the same
access pattern).  Our kernels read 8 B/lane, 512-B coalesced rows; the calibration kernel is
the full-block ConsToPrim of the same run, whose read traffic is known exactly
(nblocks * nvar * Nk*Nj*Ni * 8 B, every cell read once).  WRITE_SIZE needed no correction
(ConsToPrim writes the same number of bytes it reads; measured within 5 %).

  python profiles/pmc_traffic.py <dir with fetch_counter_collection.csv, write_counter_collection.csv> \
         --nblocks 8 --nvar 9 --ncell 134 > profiles/rNN_hbm_traffic.json
"""
import argparse
import collections
import csv
import json
import os
import re


def short(name):
    name = re.sub(r"\(apk::PackView.*", "", name)
    return name.replace("void apk::", "").replace("(anonymous namespace)::", "")


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--nblocks", type=int, default=8)
    ap.add_argument("--nvar", type=int, default=9)
    ap.add_argument("--ncell", type=int, default=134, help="cells per block edge incl. ghosts")
    ap.add_argument("--calib-kernel", default="cons_to_prim_kernel<2>")
    ap.add_argument("--calib-fraction", type=float, default=1.0,
                    help="fraction of the block cells the calibration kernel reads (1 = full ConsToPrim)")
    a = ap.parse_args()
    fetch = per_kernel(os.path.join(a.dir, "fetch_counter_collection.csv"), "FETCH_SIZE")  # KiB
    write = per_kernel(os.path.join(a.dir, "write_counter_collection.csv"), "WRITE_SIZE")  # KiB
    known = a.nblocks * a.nvar * a.ncell ** 3 * 8.0 * a.calib_fraction
    factor = known / (fetch[a.calib_kernel] * 1024.0)
    out = {"calibration": {"kernel": a.calib_kernel, "known_read_bytes": known,
                           "FETCH_SIZE_KiB": fetch[a.calib_kernel], "read_correction_factor": factor,
                           "WRITE_SIZE_KiB": write.get(a.calib_kernel),
                           "write_vs_known": write.get(a.calib_kernel, 0) * 1024.0 / known},
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if "rocclr" in k:
            continue
        rd = fetch.get(k, 0.0) * 1024.0 * factor
        wr = write.get(k, 0.0) * 1024.0
        out["kernels"][k] = {"FETCH_SIZE_KiB": fetch.get(k), "WRITE_SIZE_KiB": write.get(k),
                             "hbm_read_GB": rd / 1e9, "hbm_write_GB": wr / 1e9, "hbm_total_GB": (rd + wr) / 1e9}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
