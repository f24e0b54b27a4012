"""Summarise a rocprofv3 results .db (kernel-trace + pmc): per kernel name the call count, average
duration and the per-dispatch average of every collected counter (summed over its instances, e.g.
the 8 XCDs), plus counter / duration.  For GRBM_GUI_ACTIVE (summed over XCDs by rocprofv3: / 8) that
ratio is the effective shader clock.

    python tools/pmc_summary.py results.db [name-substring ...]
"""
import collections
import json
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    pats = sys.argv[2:]
    rows = con.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events").fetchall()
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, nm = {}, {}
    for name, did, d, cn, cv in rows:
        per[did][cn] += cv
        dur[did] = d
        nm[did] = name
    agg = collections.defaultdict(lambda: {"calls": 0, "dur": 0.0, "ctr": collections.defaultdict(float)})
    for did in per:
        a = agg[nm[did]]
        a["calls"] += 1
        a["dur"] += dur[did]
        for cn, cv in per[did].items():
            a["ctr"][cn] += cv
    out = []
    for name, a in sorted(agg.items(), key=lambda x: -x[1]["dur"]):
        if pats and not any(p in name for p in pats):
            continue
        rec = {"kernel": name[:110], "calls": a["calls"], "avg_us": a["dur"] / a["calls"] / 1e3}
        for cn, cv in a["ctr"].items():
            rec[cn] = cv / a["calls"]
            rec[cn + "_per_ns"] = cv / a["dur"]
        out.append(rec)
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
