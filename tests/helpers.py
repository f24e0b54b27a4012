"""Shared test helpers: seeded block data and thin numpy wrappers over the oracle's
block-level functions.  (Test infrastructure; the oracle is only ever the checker.)"""
import ctypes as C

import numpy as np

from oracle import oracle as O

# registry of compiled-in flux functions, src/hydro/hydro.cpp:386-416
REGISTRY = (
    [("euler", rc, rs) for rs in ("hlle", "hllc") for rc in ("dc", "plm", "ppm", "weno3", "limo3", "wenoz")]
    + [("glmmhd", rc, rs) for rs in ("hlle", "hlld") for rc in ("dc", "plm", "ppm", "weno3", "limo3", "wenoz")]
    + [("euler", "dc", "none"), ("euler", "dc", "llf"), ("glmmhd", "dc", "none"), ("glmmhd", "dc", "llf")]
)
FUSABLE = [c for c in REGISTRY if c[2] not in ("none", "llf")]

NGHOST = {"dc": 1, "plm": 2, "weno3": 2, "limo3": 2, "ppm": 3, "wenoz": 3}
NHYDRO = {"euler": 5, "glmmhd": 9}


def block_shape(nx, ng, nvar):
    ni = nx[0] + 2 * ng
    nj = nx[1] + 2 * ng if nx[1] > 1 else 1
    nk = nx[2] + 2 * ng if nx[2] > 1 else 1
    return (nvar, nk, nj, ni)


def random_prim(fluid, nx, ng, nscalars=0, seed=0, kind="smooth", nblocks=1):
    """Admissible primitive states on whole blocks incl. ghosts.
    kind: 'smooth' (sines + noise), 'rough' (uniform random), 'shock' (discontinuities)."""
    rng = np.random.default_rng(seed)
    nh = NHYDRO[fluid]
    shp = (nblocks,) + block_shape(nx, ng, nh + nscalars)
    _, _, nk, nj, ni = shp
    k, j, i = np.meshgrid(np.arange(nk), np.arange(nj), np.arange(ni), indexing="ij")
    w = np.empty(shp)
    for b in range(nblocks):
        ph = rng.uniform(0, 2 * np.pi, size=16)
        if kind == "smooth":
            f = lambda a, q: a * np.sin(2 * np.pi * (i / ni + 0.7 * j / max(nj, 2) + 0.4 * k / max(nk, 2)) + ph[q])
            w[b, 0] = 1.0 + f(0.3, 0) + 0.01 * rng.standard_normal((nk, nj, ni))
            w[b, 1] = f(0.4, 1)
            w[b, 2] = f(0.3, 2)
            w[b, 3] = f(0.2, 3)
            w[b, 4] = 1.0 + f(0.4, 4) + 0.01 * rng.standard_normal((nk, nj, ni))
            if nh == 9:
                w[b, 5] = f(0.5, 5)
                w[b, 6] = f(0.5, 6)
                w[b, 7] = f(0.5, 7)
                w[b, 8] = f(0.05, 8)
        elif kind == "rough":
            w[b, 0] = rng.uniform(0.1, 2.0, (nk, nj, ni))
            w[b, 1:4] = rng.uniform(-1.5, 1.5, (3, nk, nj, ni))
            w[b, 4] = rng.uniform(0.05, 3.0, (nk, nj, ni))
            if nh == 9:
                w[b, 5:8] = rng.uniform(-1.2, 1.2, (3, nk, nj, ni))
                w[b, 8] = rng.uniform(-0.3, 0.3, (nk, nj, ni))
        elif kind == "shock":
            left = (i + j + k) < (ni + nj + nk) / 2.2
            w[b, 0] = np.where(left, 1.0, 0.125)
            w[b, 1] = np.where(left, 0.75, -0.3)
            w[b, 2] = np.where(left, 0.0, 0.4)
            w[b, 3] = np.where(left, -0.2, 0.0)
            w[b, 4] = np.where(left, 1.0, 0.1)
            if nh == 9:
                w[b, 5] = 0.75
                w[b, 6] = np.where(left, 1.0, -1.0)
                w[b, 7] = np.where(left, 0.0, 0.3)
                w[b, 8] = np.where(left, 0.01, -0.02)
            w[b, :nh] += 1e-3 * rng.standard_normal((nh, nk, nj, ni)) * (np.arange(nh)[:, None, None, None] != 0)
        else:
            raise ValueError(kind)
        for n in range(nh, nh + nscalars):
            w[b, n] = rng.uniform(0.0, 1.0, (nk, nj, ni))
    return w


def prim_to_cons(fluid, w, gamma):
    """Inverse of ConsToPrim for building consistent cons/prim pairs in tests."""
    nh = NHYDRO[fluid]
    u = np.array(w, copy=True)
    d = w[..., 0, :, :, :]
    u[..., 1, :, :, :] = d * w[..., 1, :, :, :]
    u[..., 2, :, :, :] = d * w[..., 2, :, :, :]
    u[..., 3, :, :, :] = d * w[..., 3, :, :, :]
    ke = 0.5 * d * (w[..., 1, :, :, :] ** 2 + w[..., 2, :, :, :] ** 2 + w[..., 3, :, :, :] ** 2)
    e = w[..., 4, :, :, :] / (gamma - 1.0) + ke
    if nh == 9:
        e = e + 0.5 * (w[..., 5, :, :, :] ** 2 + w[..., 6, :, :, :] ** 2 + w[..., 7, :, :, :] ** 2)
    u[..., 4, :, :, :] = e
    for n in range(nh, w.shape[-4]):
        u[..., n, :, :, :] = d * w[..., n, :, :, :]
    return u


def geom(fluid, nx, ng, nscalars=0, dx=(1.0, 1.0, 1.0)):
    return O.make_geom(nx, ng, NHYDRO[fluid], nscalars, dx)


# ---- oracle block-level wrappers (loop over pack blocks) ----------------------------------------
def orc_fluxes(fluid, recon, riemann, g, prim, gamma, c_h, tight=False):
    lib = O.load()
    eos = O.make_eos(gamma)
    nb = prim.shape[0]
    fl = [np.zeros_like(prim) for _ in range(3)]
    for b in range(nb):
        p = np.ascontiguousarray(prim[b])
        f = [np.zeros_like(p) for _ in range(3)]
        if tight or riemann == "llf":
            lib.orc_calculate_fluxes_tight(C.byref(g), O.FLUID[fluid], C.byref(eos), c_h, O.dp(p),
                                           O.dp(f[0]), O.dp(f[1]), O.dp(f[2]))
        else:
            lib.orc_calculate_fluxes(C.byref(g), O.FLUID[fluid], O.RECON[recon], O.RIEMANN[riemann],
                                     C.byref(eos), c_h, O.dp(p), O.dp(f[0]), O.dp(f[1]), O.dp(f[2]))
        for d in range(3):
            fl[d][b] = f[d]
    return fl


def orc_update(g, u0, u1, fl, gam0, gam1, beta_dt):
    lib = O.load()
    out = np.array(u0, copy=True)
    for b in range(u0.shape[0]):
        o = np.ascontiguousarray(out[b])
        lib.orc_update_flux_div(C.byref(g), O.dp(o), O.dp(np.ascontiguousarray(u1[b])),
                                O.dp(np.ascontiguousarray(fl[0][b])), O.dp(np.ascontiguousarray(fl[1][b])),
                                O.dp(np.ascontiguousarray(fl[2][b])), gam0, gam1, beta_dt)
        out[b] = o
    return out


def orc_dedner(g, cons, prim, extended, alpha, c_h, mindx, beta_dt):
    lib = O.load()
    out = np.array(cons, copy=True)
    for b in range(cons.shape[0]):
        o = np.ascontiguousarray(out[b])
        lib.orc_dedner_source(C.byref(g), int(extended), alpha, c_h, mindx, beta_dt, O.dp(o),
                              O.dp(np.ascontiguousarray(prim[b])))
        out[b] = o
    return out


def orc_c2p(fluid, g, cons, eos):
    lib = O.load()
    cons_out = np.array(cons, copy=True)
    prim = np.zeros_like(cons)
    bad = 0
    for b in range(cons.shape[0]):
        c = np.ascontiguousarray(cons_out[b])
        p = np.zeros_like(c)
        bad += lib.orc_cons_to_prim(C.byref(g), O.FLUID[fluid], C.byref(eos), O.dp(c), O.dp(p))
        cons_out[b] = c
        prim[b] = p
    return cons_out, prim, bad


def orc_min_dt(fluid, g, prim, gamma):
    lib = O.load()
    eos = O.make_eos(gamma)
    return min(lib.orc_estimate_dt_hyp(C.byref(g), O.FLUID[fluid], C.byref(eos),
                                       O.dp(np.ascontiguousarray(prim[b]))) for b in range(prim.shape[0]))


def orc_history(fluid, g, cons):
    lib = O.load()
    tot = np.zeros(8)
    for b in range(cons.shape[0]):
        o = np.zeros(8)
        lib.orc_history(C.byref(g), O.FLUID[fluid], O.dp(np.ascontiguousarray(cons[b])), O.dp(o))
        tot += o
    return tot


def orc_turb_perturb(g, cons, acc, dt, accel_rms, box_volume):
    """in-place on copies; returns (cons, acc)"""
    lib = O.load()
    nb = cons.shape[0]
    u = [np.ascontiguousarray(cons[b]).copy() for b in range(nb)]
    a = [np.ascontiguousarray(acc[b]).copy() for b in range(nb)]
    P = C.POINTER(C.c_double)
    up = (P * nb)(*[O.dp(x) for x in u])
    ap = (P * nb)(*[O.dp(x) for x in a])
    lib.orc_turb_perturb(nb, C.byref(g), up, ap, dt, accel_rms, box_volume)
    return np.stack(u), np.stack(a)


def orc_turb_history(fluid, g, prim, gamma):
    lib = O.load()
    out = np.zeros(3)
    for b in range(prim.shape[0]):
        lib.orc_turb_history(C.byref(g), O.FLUID[fluid], gamma, O.dp(np.ascontiguousarray(prim[b])), O.dp(out))
    return out


def orc_fofc(fluid, g, u0c, u0p, u1c, fl, gamma, c_h, gam0, gam1, beta_dt):
    lib = O.load()
    eos = O.make_eos(gamma)
    out = [np.array(f, copy=True) for f in fl]
    total = 0
    for b in range(u0c.shape[0]):
        f = [np.ascontiguousarray(out[d][b]) for d in range(3)]
        total += lib.orc_first_order_flux_correct(
            C.byref(g), O.FLUID[fluid], C.byref(eos), c_h, O.dp(np.ascontiguousarray(u0c[b])),
            O.dp(np.ascontiguousarray(u0p[b])), O.dp(np.ascontiguousarray(u1c[b])), O.dp(f[0]), O.dp(f[1]),
            O.dp(f[2]), gam0, gam1, beta_dt)
        for d in range(3):
            out[d][b] = f[d]
    return out, total


def orc_stage(fluid, recon, riemann, g, u0, u1, prim, gamma, c_h, gam0, gam1, beta_dt, dedner=0,
              alpha=0.1, mindx=1.0):
    """CalculateFluxes -> UpdateWithFluxDivergence -> DednerSource (hydro_driver.cpp:510-544)."""
    fl = orc_fluxes(fluid, recon, riemann, g, prim, gamma, c_h)
    out = orc_update(g, u0, u1, fl, gam0, gam1, beta_dt)
    if dedner:
        out = orc_dedner(g, out, prim, dedner == 2, alpha, c_h, mindx, beta_dt)
    return out


def interior(a, nx, ng):
    """Slice the interior cells of [..., nvar, Nk, Nj, Ni]."""
    sk = slice(ng, ng + nx[2]) if nx[2] > 1 else slice(0, 1)
    sj = slice(ng, ng + nx[1]) if nx[1] > 1 else slice(0, 1)
    return a[..., sk, sj, ng:ng + nx[0]]


def max_rel(a, b):
    """max |a-b| / max(|b|, tiny) over the array (scale-aware)."""
    scale = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / np.maximum(scale, np.max(np.abs(b)) * 1e-3 + 1e-300)))
