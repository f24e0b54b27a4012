"""HBM traffic per kernel launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected
in separate runs as MI355X_MICROARCH.md prescribes: `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
together with `--kernel-trace` only).

On gfx950 FETCH_SIZE under-reports streaming reads (the guide measures exactly 1/2 for
16-B/lane loads and says other widths must be calibrated on a known byte count in the same
access pattern).  Our kernels read 8 B/lane, 512-B coalesced rows; the calibration kernel is
the hyperbolic-dt kernel of the same run (it runs once, at initialisation), a pure streaming
read of 8 of the 9 primitive variables over the interior cells whose traffic is known exactly.
WRITE_SIZE needed no correction (full-block ConsToPrim in the first-run profile wrote 1.448 GB
for 1.386 GB of output; the march kernels' WRITE_SIZE matches their output arrays to 1 %).

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
    ap.add_argument("--calib-kernel", default="min_dt_kernel<2>")
    ap.add_argument("--known-bytes", type=float, default=None,
                    help="exact read traffic of the calibration kernel; default: min_dt over the interior = "
                         "nblocks * (nvar-1) * nx^3 * 8 B (it reads rho,v,p,B but not psi), nx = ncell - 6")
    a = ap.parse_args()
    fetch = per_kernel(os.path.join(a.dir, "fetch_counter_collection.csv"), "FETCH_SIZE")  # KiB
    write = per_kernel(os.path.join(a.dir, "write_counter_collection.csv"), "WRITE_SIZE")  # KiB
    known = a.known_bytes if a.known_bytes else a.nblocks * (a.nvar - 1) * (a.ncell - 6) ** 3 * 8.0
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
