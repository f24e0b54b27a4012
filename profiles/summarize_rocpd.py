"""Per-kernel statistics from a rocprofv3 rocpd SQLite database (the default output of
`rocprofv3 --kernel-trace --stats`), as a markdown/CSV table.

  python profiles/summarize_rocpd.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.csv
"""
import re
import sqlite3
import sys


def demangle(names):
    """c++filt when it is there (rocpd stores the mangled symbols)"""
    import shutil
    import subprocess
    if not shutil.which("c++filt"):
        return {n: n for n in names}
    clean = [n[:-3] if n.endswith(".kd") else n for n in names]
    out = subprocess.run(["c++filt"], input="\n".join(clean), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def short(name):
    name = re.sub(r"\(apk::PackView.*", "", name)
    name = name.replace("void apk::", "").replace("(anonymous namespace)::", "")
    return name


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    pretty = demangle([r[0] for r in rows])
    print("kernel,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,sgpr,lds_bytes,scratch_bytes")
    for name, n, tot, mn, mx, vg, ag, sg, lds, scr in rows:
        print('"%s",%d,%.3f,%.1f,%.1f,%.1f,%.1f,%s,%s,%s,%s,%s' % (
            short(pretty.get(name, name)), n, tot / 1e6, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, ag, sg, lds, scr))


if __name__ == "__main__":
    main(sys.argv[1])
