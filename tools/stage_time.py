"""Time one fused stage on a pack of nb smooth n^3 GLM-MHD blocks through the C-ABI (the north-star
kernel benchmark of SURVEY 8(d)), for A/B work on the stage kernels.

    python tools/stage_time.py [--gam0 0.5] [--fill 0|2] [--dt] [--recon ppm] [--riemann hlld] [--nb 8] [--n 128] [--reps 10] [--x3only]

Prints ms per stage and the per-kernel times of the handle's own event timing.  State is rebuilt
from the same smooth data before every timed batch, results are not checked (tests do that)."""
import argparse
import math
import sys

sys.path.insert(0, ".")
import torch

from athenapk_amd import hydro, lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--gam0", type=float, default=0.5)
ap.add_argument("--fill", type=int, default=0)
ap.add_argument("--dt", action="store_true")
ap.add_argument("--recon", default="ppm")
ap.add_argument("--riemann", default="hlld")
ap.add_argument("--nb", type=int, default=8)
ap.add_argument("--n", type=int, default=128)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--generic", action="store_true", help="no variable exactly constant along an axis (PPM's extremum test fires on exact ties)")
a = ap.parse_args()
ng = 3 if a.recon in ("ppm", "wenoz") else 2
nb, n = a.nb, a.n
dev = torch.device("cuda")
N = n + 2 * ng
ctx = hydro.Context(strict=False)
ax = torch.arange(N, dtype=torch.float64, device=dev) * (2.0 * math.pi / n)
k, j, i = torch.meshgrid(ax, ax, ax, indexing="ij")
w = torch.empty((nb, 9, N, N, N), dtype=torch.float64, device=dev)
for b in range(nb):
    ph = 0.37 * b
    w[b, 0] = 1.0 + 0.2 * torch.sin(i + 2 * j + k + ph)
    w[b, 1] = 0.3 * torch.sin(j - k + ph)
    w[b, 2] = 0.3 * torch.cos(i + k)
    w[b, 3] = 0.3 * torch.sin(i - 2 * j + ph)
    w[b, 4] = 1.0 + 0.1 * torch.cos(2 * i + j - k)
    w[b, 5] = 0.5 * torch.sin(j + ph)
    w[b, 6] = 0.5 * torch.cos(k - i)
    w[b, 7] = 0.5 * torch.sin(i + j + ph)
    w[b, 8] = 0.01 * torch.sin(i + j + k)
if a.generic:
    for v in range(9):
        w[:, v] += 0.003 * torch.sin(i + 1.3 * j + 0.7 * k + 0.9 * v)
gamma = 5.0 / 3.0
u = w.clone()
u[:, 1:4] = w[:, 0:1] * w[:, 1:4]
u[:, 4] = (w[:, 4] / (gamma - 1.0) + 0.5 * w[:, 0] * (w[:, 1:4] ** 2).sum(1) + 0.5 * (w[:, 5:8] ** 2).sum(1) + 0.5 * w[:, 8] ** 2)
dx = (1.0 / n,) * 3
m0 = hydro.MeshData(ctx, (n, n, n), ng, 9, dx=dx, nblocks=nb, cons=u, prim=w, with_flux=False)
m1 = hydro.MeshData(ctx, (n, n, n), ng, 9, dx=dx, nblocks=nb, cons=u.clone(), prim=w.clone() if a.fill == 2 else None, with_flux=False)
eos = L.make_eos(gamma)


def stage():
    hydro.StageFused(m0, m1, "glmmhd", a.recon, a.riemann, eos, 2.0, a.gam0, 1.0 - a.gam0 if a.gam0 else 1.0, 1e-7, dedner=1,
                     glmmhd_alpha=0.1, mindx=dx[0], fill_derived=a.fill, estimate_dt=a.dt)


for _ in range(2):
    stage()
import ctypes as C
ctx.lib.apk_kernel_timing_enable(ctx.h, 1)
st = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(a.reps):
    stage()
e1.record(st)
e1.synchronize()
ms = e0.elapsed_time(e1) / a.reps
per = []
for slot, name in enumerate(L.TIMING_SLOTS):
    tms, cnt = C.c_double(0.0), C.c_longlong(0)
    ctx.lib.apk_kernel_timing_read(ctx.h, slot, C.byref(tms), C.byref(cnt))
    if cnt.value:
        per.append("%s %.3f" % (name, tms.value / cnt.value))
ctx.lib.apk_kernel_timing_enable(ctx.h, 0)
print("per kernel (ms):", ", ".join(per))
cells = nb * n ** 3
bytes_per = 288.0 if a.gam0 else 216.0
print("stage %.3f ms | %.3e cell-stage-updates/s | %.1f GB/s at %d B = %.2f %% of 8 TB/s" % (
    ms, cells / (ms * 1e-3), bytes_per * cells / (ms * 1e-3) / 1e9, int(bytes_per), 100 * bytes_per * cells / (ms * 1e-3) / 8e12))
