"""Approximate the instruction mix of the COMMON path of a kernel's main loop: walk the assembly from the
loop header, skip every block guarded by `s_cbranch_execz` (taken when no lane needs the rare case) and
count the rest by category.   python tools/isa_hotpath.py file.s 'kernel substring' """
import collections, re, subprocess, sys
sys.path.insert(0, "tools")
from isa_budget import kernels, classify

path, pat = sys.argv[1], sys.argv[2]
ks = kernels(path)
name = [k for k in ks if pat in k][0]
# re-read raw lines with labels for this kernel
mangled = None
txt = open(path).read().split("\n")
import subprocess
start = None
for i, l in enumerate(txt):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if pat in d:
            start = i
            break
end = next(i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end"))
lines = txt[start:end]
labels = {re.match(r"^(\.LBB\w+):", l).group(1): i for i, l in enumerate(lines) if re.match(r"^(\.LBB\w+):", l)}
# find the innermost big loop: the label with a backward branch farthest apart ... take all, pick the longest body
best = None
for i, l in enumerate(lines):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)", l) or re.search(r"s_branch\s+(\.LBB\w+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = i - labels[m.group(1)]
        if best is None or span > best[0]:
            best = (span, labels[m.group(1)], i)
_, lo, hi = best
cnt = collections.Counter()
i = lo
skipped = 0
while i <= hi:
    t = lines[i].strip()
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        i += 1
        continue
    ins = t.split(";")[0].strip()
    m = re.match(r"s_cbranch_execz\s+(\.LBB\w+)", ins)
    if m and m.group(1) in labels and labels[m.group(1)] > i:
        cnt[classify(ins)] += 1
        skipped += labels[m.group(1)] - i
        i = labels[m.group(1)]
        continue
    cnt[classify(ins)] += 1
    i += 1
valu = sum(v for k, v in cnt.items() if k in ("trans64", "div64 expansion", "fp64 arith", "fp64 compare", "int compare", "select (cndmask b32)", "dpp move", "move", "other 32-bit valu (address, int, logic)"))
print("loop lines %d..%d of %s; common path: %d instructions, %d VALU (%d lines of rare blocks skipped)" % (lo, hi, name.split("(")[0], sum(cnt.values()), valu, skipped))
for k, v in sorted(cnt.items(), key=lambda x: -x[1]):
    print("   %-42s %6d" % (k, v))
