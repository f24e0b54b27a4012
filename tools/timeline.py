"""Print the kernel timeline of the last cycles of a rocprofv3 --kernel-trace CSV: start offset, duration,
gap since the previous kernel's end (idle device time) and the kernel's short name.
  python tools/timeline.py <dir or kernel_trace.csv> [n_kernels_to_show]"""
import csv, glob, os, re, sys

p = sys.argv[1]
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip_tail = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "").replace("apk::", "")
    n = re.sub(r"at::native::.*?(\w+_kernel\w*).*", r"torch:\1", n)
    return n[:60]


tot_busy = tot_gap = 0
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) if prev_end is not None else 0
    out.append((s, e - s, gap, short(r["Kernel_Name"])))
    prev_end = max(prev_end, e) if prev_end is not None else e
tail = out[-n_show - skip_tail:len(out) - skip_tail]
t0 = tail[0][0]
for s, d, g, n in tail:
    print("%10.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, d / 1e3, g / 1e3, n))
busy = sum(d for _, d, _, _ in tail)
gaps = sum(g for _, _, g, _ in tail[1:] if g > 0)
print("last %d kernels: busy %.1f us, idle gaps %.1f us (%.1f %%)" % (len(tail), busy / 1e3, gaps / 1e3, 100.0 * gaps / (busy + gaps)))
