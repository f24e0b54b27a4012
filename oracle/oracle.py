"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the athenapk_amd package (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# enums (mirror src/main.hpp:35-38 of the reference via apk_oracle.h)
FLUID = {"euler": 1, "glmmhd": 2}
RECON = {"dc": 1, "plm": 2, "ppm": 3, "wenoz": 4, "weno3": 5, "limo3": 6}
RIEMANN = {"none": 1, "hlle": 2, "llf": 3, "hllc": 4, "hlld": 5}
INTEGRATOR = {"rk1": 1, "rk2": 2, "vl2": 3, "rk3": 4}
BC = {"periodic": 0, "outflow": 1, "reflecting": 2}

c_dp = C.POINTER(C.c_double)


class Eos(C.Structure):
    _fields_ = [("gamma", C.c_double), ("pfloor", C.c_double), ("dfloor", C.c_double),
                ("efloor", C.c_double), ("vceil", C.c_double), ("eceil", C.c_double)]


def make_eos(gamma, pfloor=-1.0, dfloor=-1.0, efloor=-1.0, vceil=float("inf"),
             eceil=float("inf")):
    return Eos(gamma, pfloor, dfloor, efloor, vceil, eceil)


class Geom(C.Structure):
    _fields_ = [("nx", C.c_int * 3), ("ng", C.c_int), ("nvar", C.c_int), ("nhydro", C.c_int),
                ("dx", C.c_double * 3)]

    @property
    def shape(self):
        ng = self.ng
        ni = self.nx[0] + 2 * ng
        nj = self.nx[1] + 2 * ng if self.nx[1] > 1 else 1
        nk = self.nx[2] + 2 * ng if self.nx[2] > 1 else 1
        return (self.nvar, nk, nj, ni)


def make_geom(nx, ng, nhydro, nscalars=0, dx=(1.0, 1.0, 1.0)):
    g = Geom()
    g.nx[:] = list(nx)
    g.ng = ng
    g.nhydro = nhydro
    g.nvar = nhydro + nscalars
    g.dx[:] = list(dx)
    return g


class SimParams(C.Structure):
    _fields_ = [("fluid", C.c_int), ("recon", C.c_int), ("riemann", C.c_int),
                ("integrator", C.c_int), ("nx", C.c_int * 3), ("mb", C.c_int * 3),
                ("ng", C.c_int), ("nscalars", C.c_int), ("bc_inner", C.c_int * 3),
                ("bc_outer", C.c_int * 3), ("xmin", C.c_double * 3), ("xmax", C.c_double * 3),
                ("cfl", C.c_double), ("glmmhd_alpha", C.c_double),
                ("dedner_extended", C.c_int), ("first_order_flux_correct", C.c_int),
                ("eos", Eos), ("nthreads", C.c_int)]


def build(fast=False, force=False):
    """Compile the oracle with gcc (strict build by default)."""
    target = "liboracle_fast.so" if fast else "liboracle.so"
    path = os.path.join(_HERE, target)
    srcs = [os.path.join(_HERE, f) for f in
            ("recon.c", "riemann.c", "block.c", "sim.c", "turbulence.c", "apk_oracle.h", "Makefile")]
    stale = (not os.path.exists(path)) or any(
        os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", target], check=True,
                       stdout=subprocess.DEVNULL)
    return path


def _declare(lib):
    d, i, l, p = C.c_double, C.c_int, C.c_long, c_dp
    G, E = C.POINTER(Geom), C.POINTER(Eos)
    sig = {
        "orc_recon_many": (None, [i, l, p, d, i, p, p]),
        "orc_riemann_many": (None, [i, i, i, l, p, p, d, d, p]),
        "orc_recon_many_traced": (None, [i, l, p, d, i, p, p, C.POINTER(C.c_uint)]),
        "orc_riemann_many_traced": (None, [i, i, i, l, p, p, d, d, p, C.POINTER(C.c_uint)]),
        "orc_c2p_many_traced": (None, [i, E, l, p, p, C.POINTER(C.c_int), C.POINTER(C.c_uint)]),
        "orc_sound_speed": (d, [d, d, d]),
        "orc_fast_speed": (d, [d, d, d, d, d, d]),
        "orc_cons_to_prim_cell": (i, [i, E, i, i, p, p]),
        "orc_calculate_fluxes": (None, [G, i, i, i, E, d, p, p, p, p]),
        "orc_calculate_fluxes_tight": (None, [G, i, E, d, p, p, p, p]),
        "orc_update_flux_div": (None, [G, p, p, p, p, p, d, d, d]),
        "orc_dedner_source": (None, [G, i, d, d, d, d, p, p]),
        "orc_cons_to_prim": (l, [G, i, E, p, p]),
        "orc_estimate_dt_hyp": (d, [G, i, E, p]),
        "orc_first_order_flux_correct": (l, [G, i, E, d, p, p, p, p, p, p, d, d, d]),
        "orc_history": (None, [G, i, p, p]),
        "orc_sim_create": (C.c_void_p, [C.POINTER(SimParams)]),
        "orc_sim_destroy": (None, [C.c_void_p]),
        "orc_sim_nblocks": (i, [C.c_void_p]),
        "orc_sim_block_geom": (None, [C.c_void_p, G]),
        "orc_sim_cons": (p, [C.c_void_p, i]),
        "orc_sim_prim": (p, [C.c_void_p, i]),
        "orc_sim_block_origin": (None, [C.c_void_p, i, p]),
        "orc_pgen_linear_wave": (d, [C.c_void_p, i, d, d]),
        "orc_pgen_linear_wave_mhd": (d, [C.c_void_p, i, d, d]),
        "orc_linear_wave_mhd_errors": (d, [C.c_void_p, i, d, d, p, p]),
        "orc_linear_wave_mhd_eigen": (None, [C.c_void_p, p, p]),
        "orc_pgen_sod": (None, [C.c_void_p, d, d, d, d, d, d, d]),
        "orc_pgen_orszag_tang": (None, [C.c_void_p]),
        "orc_pgen_field_loop": (None, [C.c_void_p, d, d, d, d, i]),
        "orc_pgen_kh": (None, [C.c_void_p, i, d, d, d, d, d, d, d]),
        "orc_user_reldivb": (d, [C.c_void_p, d]),
        "orc_pgen_advection": (None, [C.c_void_p, d, d, d, d, d, d, d, d]),
        "orc_pgen_cpaw": (d, [C.c_void_p, d, d, d, d, i, d, d]),
        "orc_cpaw_errors": (d, [C.c_void_p, p]),
        "orc_pgen_lw_implode": (None, [C.c_void_p, d, d, d, d]),
        "orc_pgen_blast": (None, [C.c_void_p, d, d, d, d, d, d, d, d, d]),
        "orc_pgen_synthetic": (None, [C.c_void_p]),
        "orc_sim_initialize": (None, [C.c_void_p]),
        "orc_sim_step": (d, [C.c_void_p, d]),
        "orc_sim_run": (i, [C.c_void_p, d, i]),
        "orc_sim_time": (d, [C.c_void_p]),
        "orc_sim_dt": (d, [C.c_void_p]),
        "orc_sim_c_h": (d, [C.c_void_p]),
        "orc_sim_fofc_count": (l, [C.c_void_p]),
        "orc_sim_history": (None, [C.c_void_p, p]),
        "orc_linear_wave_errors": (d, [C.c_void_p, i, d, d, p, p]),
        "orc_sim_gather_cons": (None, [C.c_void_p, p]),
        "orc_sim_exchange_ghosts": (None, [C.c_void_p]),
        "orc_sim_fill_derived": (None, [C.c_void_p]),
        "orc_integrator_coeffs": (i, [i, p, p, p]),
        "orc_pgen_turbulence": (None, [C.c_void_p, d, d, d, i, i, p, d, d, d, d, C.c_uint32]),
        "orc_sim_turb_history": (None, [C.c_void_p, p]),
        "orc_sim_var_hat": (p, [C.c_void_p]),
        "orc_sim_acc": (p, [C.c_void_p, i]),
        "orc_refine_dims": (None, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_prolongate_minmod": (None, [C.c_void_p, i, p, p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_restrict_average": (None, [C.c_void_p, i, i, p, p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_tag_pressure_gradient": (i, [G, p, d, C.POINTER(d)]),
        "orc_tag_velocity_gradient": (i, [G, p, d, C.POINTER(d)]),
        "orc_tag_max_density": (i, [G, p, d, d, C.POINTER(d)]),
        "orc_mt_seed": (None, [C.c_void_p, C.c_uint32]),
        "orc_mt_next": (C.c_uint32, [C.c_void_p]),
        "orc_uniform_m1_p1": (d, [C.c_void_p]),
        "orc_fmft_create": (C.c_void_p, [i, p, d, d, d, C.c_uint32]),
        "orc_fmft_destroy": (None, [C.c_void_p]),
        "orc_fmft_evolve": (None, [C.c_void_p, d]),
        "orc_fmft_phases": (None, [C.c_void_p, i, i, i, i, p]),
        "orc_fmft_inverse": (None, [C.c_void_p, G, p, p, p, p]),
        "orc_turb_history": (None, [G, i, d, p, p]),
        "orc_turb_perturb": (None, [i, G, C.POINTER(p), C.POINTER(p), d, d, d]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


_LIBS = {}


def load(fast=False):
    key = bool(fast)
    if key not in _LIBS:
        lib = C.CDLL(build(fast=fast))
        _declare(lib)
        _LIBS[key] = lib
    return _LIBS[key]


def dp(a):
    """double* of a C-contiguous float64 numpy array."""
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_dp)


# ---------------------------------------------------------------------------------------
# convenience wrappers
def recon_many(recon, q, dx=1.0, n=0, lib=None):
    lib = lib or load()
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 5)
    m = q.shape[0]
    ql, qr = np.empty(m), np.empty(m)
    lib.orc_recon_many(RECON[recon], m, dp(q), dx, n, dp(ql), dp(qr))
    return ql, qr


def riemann_many(fluid, riemann, ivx, wl, wr, gamma, c_h=0.0, lib=None):
    lib = lib or load()
    nv = 5 if fluid == "euler" else 9
    wl = np.ascontiguousarray(wl, dtype=np.float64).reshape(-1, nv)
    wr = np.ascontiguousarray(wr, dtype=np.float64).reshape(-1, nv)
    out = np.empty_like(wl)
    lib.orc_riemann_many(FLUID[fluid], RIEMANN[riemann], ivx, wl.shape[0], dp(wl), dp(wr),
                         gamma, c_h, dp(out))
    return out


TRACE_BITS = {"ppm_lim_m": 1 << 0, "ppm_lim_p": 1 << 1, "ppm_extremum": 1 << 2, "ppm_roundoff": 1 << 3,
              "ppm_ratio_big": 1 << 4, "ppm_over_m": 1 << 5, "ppm_over_p": 1 << 6,
              "hlld_fl": 1 << 8, "hlld_fr": 1 << 9, "hlld_lstar": 1 << 10, "hlld_ldstar": 1 << 11, "hlld_rdstar": 1 << 12,
              "hlld_rstar": 1 << 13, "hlld_deg_l": 1 << 14, "hlld_deg_r": 1 << 15, "hlld_deg_dst": 1 << 16,
              "hllc_am_pos": 1 << 18, "hllc_cp_clip": 1 << 19, "hllc_ql": 1 << 20, "hllc_qr": 1 << 21,
              "hlle_bp_eq_bm": 1 << 22, "c2p_dfloor": 1 << 24, "c2p_vceil": 1 << 25, "c2p_pfloor": 1 << 26,
              "c2p_efloor": 1 << 27, "c2p_eceil": 1 << 28}


def trace_names(mask):
    return [k for k, b in TRACE_BITS.items() if mask & b]


def recon_many_traced(recon, q, dx=1.0, n=0, lib=None):
    """(ql, qr, branch masks) -- the reference's special-case branches each stencil takes (apk_oracle.h)"""
    lib = lib or load()
    q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1, 5)
    m = q.shape[0]
    ql, qr, masks = np.empty(m), np.empty(m), np.zeros(m, dtype=np.uint32)
    lib.orc_recon_many_traced(RECON[recon], m, dp(q), dx, n, dp(ql), dp(qr), masks.ctypes.data_as(C.POINTER(C.c_uint)))
    return ql, qr, masks


def riemann_many_traced(fluid, riemann, ivx, wl, wr, gamma, c_h=0.0, lib=None):
    lib = lib or load()
    nv = 5 if fluid == "euler" else 9
    wl = np.ascontiguousarray(wl, dtype=np.float64).reshape(-1, nv)
    wr = np.ascontiguousarray(wr, dtype=np.float64).reshape(-1, nv)
    out, masks = np.empty_like(wl), np.zeros(wl.shape[0], dtype=np.uint32)
    lib.orc_riemann_many_traced(FLUID[fluid], RIEMANN[riemann], ivx, wl.shape[0], dp(wl), dp(wr), gamma, c_h, dp(out),
                                masks.ctypes.data_as(C.POINTER(C.c_uint)))
    return out, masks


def c2p_many_traced(fluid, eos, u, lib=None):
    """ConsToPrim of rows [m][nhydro]: (u after floors, w, status, branch masks)"""
    lib = lib or load()
    nv = 5 if fluid == "euler" else 9
    u = np.array(u, dtype=np.float64).reshape(-1, nv).copy()
    w = np.empty_like(u)
    st, masks = np.zeros(u.shape[0], dtype=np.int32), np.zeros(u.shape[0], dtype=np.uint32)
    lib.orc_c2p_many_traced(FLUID[fluid], C.byref(eos), u.shape[0], dp(u), dp(w), st.ctypes.data_as(C.POINTER(C.c_int)),
                            masks.ctypes.data_as(C.POINTER(C.c_uint)))
    return u, w, st, masks


def integrator_coeffs(name, lib=None):
    lib = lib or load()
    b, g0, g1 = np.zeros(4), np.zeros(4), np.zeros(4)
    n = lib.orc_integrator_coeffs(INTEGRATOR[name], dp(b), dp(g0), dp(g1))
    return n, b[:n], g0[:n], g1[:n]


def usable_cores():
    """cores this process may really use (affinity mask and cgroup CPU quota): GPU boxes expose all
    256 hardware threads of the node to a container that is allowed 16 of them, and an OpenMP
    team of 256 on 16 cores is several times slower than a team of 16"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


class Sim:
    """Thin OO wrapper over the orc_sim_* mini-driver."""

    def __init__(self, fluid="euler", recon="plm", riemann="hlle", integrator="vl2",
                 nx=(64, 32, 32), mb=None, ng=2, nscalars=0, bc=("periodic",) * 3,
                 bc_outer=None, xmin=(0.0, 0.0, 0.0), xmax=(1.0, 1.0, 1.0), cfl=0.3,
                 gamma=5.0 / 3.0, glmmhd_alpha=0.1, dedner_extended=False, fofc=False,
                 eos=None, nthreads=0, fast=False):
        self.lib = load(fast=fast)
        p = SimParams()
        p.fluid, p.recon, p.riemann = FLUID[fluid], RECON[recon], RIEMANN[riemann]
        p.integrator = INTEGRATOR[integrator]
        p.nx[:] = list(nx)
        p.mb[:] = list(mb or nx)
        p.ng, p.nscalars = ng, nscalars
        p.bc_inner[:] = [BC[b] for b in bc]
        p.bc_outer[:] = [BC[b] for b in (bc_outer or bc)]
        p.xmin[:] = list(xmin)
        p.xmax[:] = list(xmax)
        p.cfl, p.glmmhd_alpha = cfl, glmmhd_alpha
        p.dedner_extended, p.first_order_flux_correct = int(dedner_extended), int(fofc)
        p.eos = eos or make_eos(gamma)
        cores = usable_cores()
        p.nthreads = cores if (nthreads <= 0 or nthreads > cores) else nthreads
        self.params = p
        self.h = self.lib.orc_sim_create(C.byref(p))
        if not self.h:
            raise ValueError("mesh not divisible by meshblock size")
        self.geom = Geom()
        self.lib.orc_sim_block_geom(self.h, C.byref(self.geom))
        self.fluid = fluid
        self.nx = tuple(nx)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_sim_destroy(self.h)
            self.h = None

    @property
    def nblocks(self):
        return self.lib.orc_sim_nblocks(self.h)

    def cons(self, b):
        return np.ctypeslib.as_array(self.lib.orc_sim_cons(self.h, b), shape=self.geom.shape)

    def prim(self, b):
        return np.ctypeslib.as_array(self.lib.orc_sim_prim(self.h, b), shape=self.geom.shape)

    def pgen(self, name, **kw):
        if name == "linear_wave":
            self._lw = (kw.get("wave_flag", 0), kw.get("amp", 1e-6), kw.get("vflow", 0.0))
            self.period = self.lib.orc_pgen_linear_wave(self.h, *self._lw)
        elif name == "linear_wave_mhd":
            self._lw = (kw.get("wave_flag", 0), kw.get("amp", 1e-6), kw.get("vflow", 0.0))
            self._lw_mhd = True
            self.period = self.lib.orc_pgen_linear_wave_mhd(self.h, *self._lw)
        elif name == "sod":
            self.lib.orc_pgen_sod(self.h, kw.get("rho_l", 1.0), kw.get("pres_l", 1.0),
                                  kw.get("u_l", 0.0), kw.get("rho_r", 0.125),
                                  kw.get("pres_r", 0.1), kw.get("u_r", 0.0),
                                  kw.get("x_discont", 0.5))
        elif name == "orszag_tang":
            self.lib.orc_pgen_orszag_tang(self.h)
        elif name == "synthetic":
            self.lib.orc_pgen_synthetic(self.h)
        elif name == "kh":
            self.lib.orc_pgen_kh(self.h, kw.get("iprob", 4), kw.get("vflow", 1.0), kw.get("amp", 0.01),
                                 kw.get("drho_rho0", 0.0), kw.get("vboost", 0.0), kw.get("a", 0.01), kw.get("sigma", 0.1),
                                 kw.get("drat", 2.0))
        elif name == "field_loop":
            self.lib.orc_pgen_field_loop(self.h, kw.get("rad", 0.3), kw.get("amp", 1e-3), kw.get("vflow", 1.0),
                                         kw.get("drat", 1.0), kw.get("iprob", 1))
        elif name == "advection":
            self.lib.orc_pgen_advection(self.h, kw.get("vx", 0.0), kw.get("vy", 0.0), kw.get("vz", 0.0),
                                        kw.get("rho_ratio", 1.0), kw.get("rho_radius", 0.0),
                                        kw.get("rho_fraction_edge", 0.01), kw.get("rho0", 1.0), kw.get("p0", 1.0))
        elif name == "cpaw":
            self.cpaw_lambda = self.lib.orc_pgen_cpaw(self.h, kw.get("b_par", 1.0), kw.get("b_perp", 0.1),
                                                      kw.get("pres", 0.1), kw.get("v_par", 0.0), kw.get("dir", 1),
                                                      kw.get("ang_2", -999.9), kw.get("ang_3", -999.9))
        elif name == "lw_implode":
            self.lib.orc_pgen_lw_implode(self.h, kw.get("d_in", 0.125), kw.get("p_in", 0.14), kw.get("d_out", 1.0),
                                         kw.get("p_out", 1.0))
        elif name == "blast":
            self.lib.orc_pgen_blast(self.h, kw["radius_outer"], kw.get("radius_inner", kw["radius_outer"]),
                                    kw.get("pressure_ambient", 1.0), kw.get("density_ambient", 1.0),
                                    kw["pressure_ratio"], kw.get("density_ratio", 1.0), kw.get("x1_0", 0.0),
                                    kw.get("x2_0", 0.0), kw.get("x3_0", 0.0))
        elif name == "turbulence":
            kv = np.ascontiguousarray(kw["k_vec"], dtype=np.float64)  # [3][M]
            self._turb_modes = kv.shape[1]
            self.lib.orc_pgen_turbulence(self.h, kw.get("rho0", 1.0), kw.get("p0", 1.0), kw.get("b0", 0.01),
                                         kw.get("b_config", 0), kv.shape[1], dp(kv), kw.get("kpeak", 2.0),
                                         kw.get("sol_weight", 1.0), kw.get("corr_time", 1.0),
                                         kw.get("accel_rms", 0.5), kw.get("rseed", 20190729))
        else:
            raise ValueError(name)
        self.lib.orc_sim_initialize(self.h)
        return self

    def step(self, tlim=1e300):
        return self.lib.orc_sim_step(self.h, tlim)

    def run(self, tlim, nlim=-1):
        return self.lib.orc_sim_run(self.h, tlim, nlim)

    time = property(lambda self: self.lib.orc_sim_time(self.h))
    dt = property(lambda self: self.lib.orc_sim_dt(self.h))
    c_h = property(lambda self: self.lib.orc_sim_c_h(self.h))
    fofc_count = property(lambda self: self.lib.orc_sim_fofc_count(self.h))

    def history(self):
        out = np.zeros(8)
        self.lib.orc_sim_history(self.h, dp(out))
        return out

    def turb_history(self):
        """volume sums of sonic Mach, Alfvenic Mach, plasma beta (TurbulenceHst)"""
        out = np.zeros(3)
        self.lib.orc_sim_turb_history(self.h, dp(out))
        return out

    def var_hat(self):
        return np.ctypeslib.as_array(self.lib.orc_sim_var_hat(self.h), shape=(3, self._turb_modes, 2)).copy()

    def acc(self, b):
        return np.ctypeslib.as_array(self.lib.orc_sim_acc(self.h, b), shape=(3,) + self.geom.shape[1:])

    def user_reldivb(self, B0):
        return self.lib.orc_user_reldivb(self.h, B0)

    def cpaw_errors(self):
        err = np.zeros(8)
        rms = self.lib.orc_cpaw_errors(self.h, dp(err))
        return rms, err

    def linear_wave_mhd_eigen(self):
        """(ev[7], rem[7][7]) of the MHD linear-wave background: columns of rem are the right eigenvectors"""
        ev, rem = np.zeros(7), np.zeros((7, 7))
        self.lib.orc_linear_wave_mhd_eigen(self.h, dp(ev), dp(rem))
        return ev, rem

    def linear_wave_errors(self):
        if getattr(self, "_lw_mhd", False):   # d, M1, M2, M3, E, B1, B2, B3
            l1, mx = np.zeros(8), np.zeros(8)
            rms = self.lib.orc_linear_wave_mhd_errors(self.h, *self._lw, dp(l1), dp(mx))
            return rms, l1, mx
        l1, mx = np.zeros(5), np.zeros(5)
        rms = self.lib.orc_linear_wave_errors(self.h, *self._lw, dp(l1), dp(mx))
        return rms, l1, mx

    def gather_cons(self):
        out = np.zeros((self.geom.nvar, self.nx[2], self.nx[1], self.nx[0]))
        self.lib.orc_sim_gather_cons(self.h, dp(out))
        return out


class RefineGeom(C.Structure):
    _fields_ = [("nx", C.c_int * 3), ("ng", C.c_int), ("cng", C.c_int), ("xmin", C.c_double * 3),
                ("dx", C.c_double * 3)]

    def dims(self):
        """(fine shape, coarse shape) as (nk, nj, ni)"""
        f, c = (C.c_int * 3)(), (C.c_int * 3)()
        load().orc_refine_dims(C.byref(self), f, c)
        return (f[2], f[1], f[0]), (c[2], c[1], c[0])


def make_refine_geom(nx, ng, cng, xmin=(0.0, 0.0, 0.0), dx=(1.0, 1.0, 1.0)):
    r = RefineGeom()
    r.nx[:] = list(nx)
    r.ng, r.cng = ng, cng
    r.xmin[:] = list(xmin)
    r.dx[:] = list(dx)
    return r


def prolongate(r, coarse, fine, lo, hi):
    """in place on `fine`; coarse [nvar][cNk][cNj][cNi]"""
    I3 = C.c_int * 3
    load().orc_prolongate_minmod(C.byref(r), coarse.shape[0], dp(coarse), dp(fine), I3(*lo), I3(*hi))


def restrict(r, el, fine, coarse, lo, hi):
    I3 = C.c_int * 3
    load().orc_restrict_average(C.byref(r), fine.shape[0], el, dp(fine), dp(coarse), I3(*lo), I3(*hi))


def tag(kind, g, prim, p0, p1=0.0):
    lib = load()
    crit = C.c_double(0.0)
    if kind == "pressure_gradient":
        t = lib.orc_tag_pressure_gradient(C.byref(g), dp(prim), p0, C.byref(crit))
    elif kind == "xyvelocity_gradient":
        t = lib.orc_tag_velocity_gradient(C.byref(g), dp(prim), p0, C.byref(crit))
    else:
        t = lib.orc_tag_max_density(C.byref(g), dp(prim), p0, p1, C.byref(crit))
    return t, crit.value


class MT19937(C.Structure):
    """std::mt19937 restated (turbulence.c)"""
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


class _FmftStruct(C.Structure):
    _fields_ = [("num_modes", C.c_int), ("k_peak", C.c_double), ("sol_weight", C.c_double), ("t_corr", C.c_double),
                ("k_vec", C.POINTER(C.c_double)), ("var_hat", C.POINTER(C.c_double)),
                ("var_hat_new", C.POINTER(C.c_double)), ("rng", MT19937)]


class Fmft:
    """FewModesFT spectral state (orc_fmft)"""

    def __init__(self, lib, k_vec, k_peak=2.0, sol_weight=1.0, t_corr=1.0, rseed=20190729):
        self.lib = lib
        kv = np.ascontiguousarray(k_vec, dtype=np.float64)
        self.M = kv.shape[1]
        self.h = lib.orc_fmft_create(self.M, dp(kv), k_peak, sol_weight, t_corr, rseed)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_fmft_destroy(self.h)
            self.h = None

    def evolve(self, dt):
        self.lib.orc_fmft_evolve(self.h, dt)

    def var_hat(self):
        st = C.cast(self.h, C.POINTER(_FmftStruct)).contents
        return np.ctypeslib.as_array(st.var_hat, shape=(3, self.M, 2)).copy()

    def phases(self, axis, n, g0, gn):
        out = np.zeros((n, self.M, 2))
        self.lib.orc_fmft_phases(self.h, axis, n, g0, gn, dp(out))
        return out

    def inverse(self, geom, ph_i, ph_j, ph_k):
        acc = np.zeros((3,) + geom.shape[1:])
        self.lib.orc_fmft_inverse(self.h, C.byref(geom), dp(np.ascontiguousarray(ph_i)),
                                  dp(np.ascontiguousarray(ph_j)), dp(np.ascontiguousarray(ph_k)), dp(acc))
        return acc
