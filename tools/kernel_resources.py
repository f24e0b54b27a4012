"""Register / scratch use of the kernels of one fused translation unit (hipcc -Rpass-analysis=kernel-resource-usage):

    python tools/kernel_resources.py [fused_mhd_hlld.hip] [name substring ...] [-- extra hipcc flags]
"""
import re
import subprocess
import sys

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
src = args[0] if args and args[0].endswith(".hip") else "fused_mhd_hlld.hip"
pats = [a for a in args if not a.endswith(".hip")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-Iathenapk_amd/csrc", "-fapprox-func",
       "-freciprocal-math", "-c", "athenapk_amd/csrc/" + src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
if "error:" in out:
    print(out[-3000:])
    sys.exit(1)
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|TotalSGPRs|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = {"name": m.group(2)}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(1)] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("# static LDS only: the marches allocate their ring / stash dynamically (fused_march / m12f: 2 H x nvar x 512 B per wave = 18432 B for PPM GLM-MHD;\n"
      "# dc3: 9216 B, dc3r2: 18432 B); occ = waves per SIMD by registers (two per SIMD also by the 20 KB of LDS a wave may take)")
print("%-58s %5s %5s %5s %7s %7s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "s-spill", "v-spill", "LDS", "occ"))
for r, n in zip(rows, names):
    n = re.sub(r"\(apk::PackView.*", "", n).replace("void apk::", "").replace("apk::", "")
    if pats and not any(p in n for p in pats):
        continue
    print("%-58s %5s %5s %5s %7s %7s %7s %6s %4s" % (n[:58], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
                                                    r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
