"""register / scratch / LDS usage per kernel from a --save-temps .s file:  python tools/isa_regs.py file.s [substr ...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S)
rows = []
for b in blocks:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    d = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))
    rows.append((name, d("vgpr_count"), d("sgpr_count"), d("private_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for n, r in zip(names, rows):
    if len(sys.argv) < 3 or any(p in n for p in sys.argv[2:]):
        print("%-64s vgpr %3d sgpr %3d scratch %3d" % (n.split("(")[0].replace("void apk::", ""), r[1], r[2], r[3]))
