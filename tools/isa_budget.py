"""Per-category static instruction budget of one kernel in a hipcc --save-temps .s file.

    python tools/isa_budget.py file.s 'fused_x1_kernel<2, 3, 5, false>' [more kernel substrings...]

Static counts (every instruction once, both sides of every branch); the dynamic count per wave is
what rocprofv3's SQ_INSTS_VALU reports.  Categories are chosen for an fp64 stencil kernel:
fp64 arithmetic issues at 4 cycles per wave64 instruction on gfx950's SIMD-32 (78.6 TFLOP/s),
32-bit VALU (moves, selects, integer, DPP) at 2, transcendentals (v_rcp/rsq/sqrt_f64) at 16.
"""
import collections
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return out


def kernels(path):
    """{demangled name: [instruction lines]}"""
    cur, body, res = None, [], {}
    for line in open(path):
        m = re.match(r"^(_Z\w+|k_\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur and line.startswith(".Lfunc_end"):
            res[cur] = body
            cur = None
            continue
        if cur:
            t = line.strip()
            if t and not t.startswith((";", ".", "//")) and not t.endswith(":"):
                body.append(t.split(";")[0].strip())
    names = list(res)
    return dict(zip(demangle(names), [res[n] for n in names]))


def classify(ins):
    op = ins.split()[0]
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "trans64"
    if op.startswith(("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")):
        return "div64 expansion"
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_fmac_f64", "v_ldexp_f64",
                      "v_frexp", "v_trunc_f64", "v_rndne_f64")):
        return "fp64 arith"
    if op.startswith("v_cmp") and "f64" in op:
        return "fp64 compare"
    if op.startswith("v_cmp"):
        return "int compare"
    if op.startswith("v_cndmask"):
        return "select (cndmask b32)"
    if op.startswith("v_mov") and "dpp" in ins:
        return "dpp move"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane", "v_pk_mov")):
        return "move"
    if op.startswith("v_"):
        return "other 32-bit valu (address, int, logic)"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "vmem store/atomic"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec")):
        return "branch/exec"
    if op.startswith("s_"):
        return "salu/smem"
    return "other"


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    ks = kernels(path)
    for pat in pats:
        hit = [k for k in ks if pat in k]
        for k in hit:
            c = collections.Counter(classify(i) for i in ks[k])
            valu = sum(v for kk, v in c.items() if kk in ("trans64", "div64 expansion", "fp64 arith", "fp64 compare", "int compare",
                                                         "select (cndmask b32)", "dpp move", "move",
                                                         "other 32-bit valu (address, int, logic)"))
            print("== %s" % k.split("(")[0])
            print("   total %d instructions, %d VALU" % (sum(c.values()), valu))
            for kk, v in sorted(c.items(), key=lambda x: -x[1]):
                print("   %-42s %6d" % (kk, v))


if __name__ == "__main__":
    main()
