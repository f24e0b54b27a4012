"""Development aid: what does packing the faces of 128^3 meshblocks (nghost 3, 9 variables) cost per direction?"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from athenapk_amd import hydro, lib as L

ctx = hydro.Context()
lib = ctx.lib
n, ng, nv, nb = 128, 3, 9, 8
nt = n + 2 * ng
sn = nt ** 3
src = torch.randn(nb * nv * sn, dtype=torch.float64, device="cuda")
dst = torch.zeros(nb * nv * 3 * nt * nt + 1024, dtype=torch.float64, device="cuda")


def plan(ext_of, off_of, reverse=False):
    regs = (L.CopyRegion * nb)()
    cells = ext_of[0] * ext_of[1] * ext_of[2]
    for b in range(nb):
        r = regs[b]
        a = src.data_ptr() + 8 * (b * nv * sn + off_of[2] * nt * nt + off_of[1] * nt + off_of[0])
        m = dst.data_ptr() + 8 * (b * nv * cells)
        r.src, r.dst = (m, a) if reverse else (a, m)
        r.ext[:] = ext_of
        r.nvar = nv
        big, small = (1, nt, nt * nt, sn), (1, ext_of[0], ext_of[0] * ext_of[1], cells)
        r.src_stride[:] = small if reverse else big
        r.dst_stride[:] = big if reverse else small
        r.flip_var = -1
    h = C.c_void_p()
    assert lib.apk_copy_plan_create(ctx.h, regs, nb, C.byref(h)) == 0
    return h, cells * nv * 8 * nb


def timeit(p, reps=20):
    for _ in range(3):
        lib.apk_copy_plan_run(ctx.h, p, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.apk_copy_plan_run(ctx.h, p, None)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, ext, off in (("x1 face 3 x 128 x 128", (3, n, n), (ng, ng, ng)), ("x2 face 128 x 3 x 128", (n, 3, n), (ng, ng, ng)),
                       ("x3 face 128 x 128 x 3", (n, n, 3), (ng, ng, ng)), ("x1 ghost 3 x 134 x 134", (3, nt, nt), (0, 0, 0)),
                       ("x3 ghost 134 x 134 x 3", (nt, nt, 3), (0, 0, 0))):
    for rev in (False, True):
        p, nbytes = plan(ext, off, rev)
        ms = timeit(p)
        print("%-24s %-6s %6.1f MB  %7.1f us  %6.2f TB/s (read + write)" % (name, "unpack" if rev else "pack", nbytes / 1e6, ms * 1e3, 2 * nbytes / ms / 1e9), flush=True)
        lib.apk_copy_plan_destroy(p)
