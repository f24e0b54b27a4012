"""Instruction ledger of stage kernels from their ISA text (hipcc --save-temps .s).

    python tools/isa_loop_ledger.py file.s 'fused_m12f_kernel<2, 3, 5, 2>' [more kernel substrings ...] [--json out.json]

Static counts over the whole kernel (prologue + march loop; the loop is > 90 % of it).  Regions guarded by
`s_cbranch_execz L` (a divergent branch, entered when SOME lane takes it) are split by size: the small ones (< 100 VALU
instructions) are PPM's limiter branches and HLLD's double-star block -- masked code that runs with a few live lanes --
the large ones (Riemann solves, finish) run in every iteration.  `always` = straight-line path + large guarded regions,
`masked` = the small guarded regions (static: how often they are entered depends on the data; SQ_INSTS_VALU has the
dynamic total).  With -DAPK_PHASE_MARKERS=1 the finishing march carries `;;APK_PHASE n` comments at its phase
boundaries (the APK_TICK points of fused2_kernel.hpp) and the ledger is also given per phase, in layout order.
Categories as tools/isa_budget.py, with v_readlane / v_writelane (SGPR spills) and v_mov split out of `move`.
"""
import collections
import json
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
import isa_budget as ib  # noqa: E402

VALU = ("trans64", "div64 expansion", "fp64 arith", "fp64 min/max", "fp64 compare", "int compare", "select (cndmask b32)", "dpp move", "move",
        "sgpr spill (readlane/writelane)", "64-bit address add", "other 32-bit valu (int, logic)")
SMALL = 100
PHASES = {0: "loop control / other", 1: "x1 reconstruction", 2: "x1 Riemann + flux difference", 3: "x2 reconstruction + ring + loads",
          4: "x2 Riemann", 5: "d3/u1 wait", 6: "finish (update, Dedner, ConsToPrim, dt, stores)"}


def classify(ins):
    op = ins.split()[0]
    if op.startswith(("v_readlane", "v_writelane")):
        return "sgpr spill (readlane/writelane)"
    if op.startswith(("v_max_f64", "v_min_f64")):
        return "fp64 min/max"
    if op.startswith(("v_lshl_add_u64", "v_add_co_u32", "v_addc_co_u32", "v_mad_u64_u32", "v_mad_i64_i32")):
        return "64-bit address add"
    c = ib.classify(ins)
    return "other 32-bit valu (int, logic)" if c.startswith("other 32-bit") else c


def kernel_lines(path):
    """{demangled name: raw lines (labels kept)}"""
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
            body.append(line.rstrip("\n"))
    names = list(res)
    return dict(zip(ib.demangle(names), [res[n] for n in names]))


def ledger(lines):
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i

    def ins_of(i):
        t = lines[i].strip()
        if not t or t.startswith((";", ".", "//")) or re.match(r"^\.LBB", t):
            return None
        return t.split(";")[0].strip()

    regions = []
    for i, l in enumerate(lines):
        g = re.search(r"s_cbranch_execz\s+(\.LBB\w+)", l)
        if g and g.group(1) in labels and labels[g.group(1)] > i:
            regions.append((i + 1, labels[g.group(1)]))

    def count(a, b, skip=()):
        c = collections.Counter()
        for i in range(a, b):
            if any(x <= i < y for x, y in skip):
                continue
            ins = ins_of(i)
            if ins:
                c[classify(ins)] += 1
        return c

    def nvalu(c):
        return sum(c[k] for k in VALU)
    small = [(a, b) for a, b in regions if nvalu(count(a, b)) < SMALL]
    small = [r for r in small if not any(o != r and o[0] <= r[0] and r[1] <= o[1] for o in small)]
    lim = collections.Counter()
    for a, b in small:
        lim.update(count(a, b))
    alw = count(0, len(lines), skip=small)
    res = {"always": dict(alw), "masked": dict(lim), "always_valu": nvalu(alw), "masked_valu": nvalu(lim), "masked_regions": len(small)}
    marks = [(i, int(re.search(r";;APK_PHASE (\d+)", l).group(1))) for i, l in enumerate(lines) if ";;APK_PHASE" in l]
    if marks:
        # a marker closes the phase it names (APK_TICK(n) accounts the time since the previous tick to slot n)
        per = {}
        prev = marks[0][0]
        for i, ph in marks[1:]:
            a, m = per.setdefault(PHASES.get(ph, str(ph)), [collections.Counter(), collections.Counter()])
            a.update(count(prev, i, skip=small))
            for x, y in small:
                if prev <= x < i:
                    m.update(count(x, y))
            prev = i
        res["phases"] = {k: {"always_valu": nvalu(a), "masked_valu": nvalu(m), "always": dict(a), "masked": dict(m)} for k, (a, m) in per.items()}
    return res


if __name__ == "__main__":
    args = sys.argv[1:]
    out = None
    if "--json" in args:
        i = args.index("--json")
        out = args[i + 1]
        del args[i:i + 2]
    ks = kernel_lines(args[0])
    res = {}
    for pat in args[1:]:
        for k in [k for k in ks if pat in k]:
            name = k.split("(")[0].replace("void apk::", "")
            r = ledger(ks[k])
            res[name] = r
            a, g = r["always"], r["masked"]
            print("== %s  (%d masked regions)" % (name, r["masked_regions"]))
            print("   %-44s %8s %8s" % ("category", "always", "masked"))
            for c in sorted(set(a) | set(g), key=lambda c: -(a.get(c, 0) + g.get(c, 0))):
                print("   %-44s %8d %8d" % (c, a.get(c, 0), g.get(c, 0)))
            print("   %-44s %8d %8d" % ("VALU total", r["always_valu"], r["masked_valu"]))
            for ph, v in r.get("phases", {}).items():
                print("   phase %-46s always %5d  masked %5d" % (ph, v["always_valu"], v["masked_valu"]))
    if out:
        json.dump(res, open(out, "w"), indent=1)
