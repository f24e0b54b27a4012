/*
 * block.c -- ORACLE (test infrastructure only; see apk_oracle.h).
 * Block-level restatement of the hot path: flux sweeps, flux-divergence update, Dedner
 * source, cons->prim, hyperbolic dt, first-order flux correction, history sums.
 * Arrays are [nvar][Nk][Nj][Ni] with i fastest (the reference's LayoutRight blocks).
 */
#include "apk_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline double sq(double x) { return x * x; }

int orc_ndim(const orc_geom *g) { return (g->nx[2] > 1) ? 3 : ((g->nx[1] > 1) ? 2 : 1); }
int orc_ni(const orc_geom *g) { return g->nx[0] + 2 * g->ng; }
int orc_nj(const orc_geom *g) { return (g->nx[1] > 1) ? g->nx[1] + 2 * g->ng : 1; }
int orc_nk(const orc_geom *g) { return (g->nx[2] > 1) ? g->nx[2] + 2 * g->ng : 1; }
long orc_ncell(const orc_geom *g) { return (long)orc_ni(g) * orc_nj(g) * orc_nk(g); }

/* interior index bounds */
typedef struct {
  int is, ie, js, je, ks, ke, ni, nj, nk;
  long sj, sk, sn; /* strides */
} bounds_t;

static bounds_t get_bounds(const orc_geom *g) {
  bounds_t b;
  b.ni = orc_ni(g);
  b.nj = orc_nj(g);
  b.nk = orc_nk(g);
  b.is = g->ng;
  b.ie = g->ng + g->nx[0] - 1;
  if (g->nx[1] > 1) {
    b.js = g->ng;
    b.je = g->ng + g->nx[1] - 1;
  } else {
    b.js = b.je = 0;
  }
  if (g->nx[2] > 1) {
    b.ks = g->ng;
    b.ke = g->ng + g->nx[2] - 1;
  } else {
    b.ks = b.ke = 0;
  }
  b.sj = b.ni;
  b.sk = (long)b.ni * b.nj;
  b.sn = b.sk * b.nk;
  return b;
}

#define AT(arr, n, k, j, i) (arr)[(n)*bb.sn + (k)*bb.sk + (j)*bb.sj + (i)]

/* ---------------------------------------------------------------------------------------
 * One pencil of reconstruction along direction dir (1,2,3) for all variables, as the
 * Reconstruct<recon,DIR> wrappers do: fills ql[n][i] (L state at the upper face of the
 * cell) and qr[n][i] (R state at the lower face) for i in [il,iu].
 * For dir==1 the caller applies the i+1 offset (plm_simple.hpp:54-58). */
static void recon_pencil(const orc_geom *g, const bounds_t *pb, int recon, int dir,
                         const double *prim, int k, int j, int il, int iu, double *ql,
                         double *qr) {
  const bounds_t bb = *pb;
  const long st = (dir == 1) ? 1 : ((dir == 2) ? bb.sj : bb.sk);
  const double dx = g->dx[dir - 1];
  for (int n = 0; n < g->nvar; ++n) {
    for (int i = il; i <= iu; ++i) {
      const double *c = &AT(prim, n, k, j, i);
      double q[5];
      q[2] = c[0];
      if (recon == ORC_RC_DC) {
        q[0] = q[1] = q[3] = q[4] = c[0];
      } else if (recon == ORC_RC_PPM || recon == ORC_RC_WENOZ) {
        q[0] = c[-2 * st];
        q[1] = c[-st];
        q[3] = c[st];
        q[4] = c[2 * st];
      } else {
        q[0] = q[4] = 0.0;
        q[1] = c[-st];
        q[3] = c[st];
      }
      double l, r;
      orc_recon_point(recon, q, dx, n, &l, &r);
      if (dir == 1) {
        ql[n * bb.ni + i + 1] = l;
        qr[n * bb.ni + i] = r;
      } else {
        ql[n * bb.ni + i] = l;
        qr[n * bb.ni + i] = r;
      }
    }
  }
}

/* Riemann::Solve over a pencil + passive scalar fluxes (hydro.cpp:1084-1097) */
static void riemann_pencil(const orc_geom *g, const bounds_t *pb, int fluid, int riemann,
                           int ivx, double gamma, double c_h, const double *wl,
                           const double *wr, int k, int j, int il, int iu, double *flux) {
  const bounds_t bb = *pb;
  const int nh = g->nhydro;
  double l[ORC_NGLMMHD], r[ORC_NGLMMHD], f[ORC_NGLMMHD];
  for (int i = il; i <= iu; ++i) {
    for (int n = 0; n < nh; ++n) {
      l[n] = wl[n * bb.ni + i];
      r[n] = wr[n * bb.ni + i];
    }
    orc_riemann_point(fluid, riemann, ivx, l, r, gamma, c_h, f);
    for (int n = 0; n < nh; ++n) AT(flux, n, k, j, i) = f[n];
    for (int n = nh; n < g->nvar; ++n) {
      if (f[ORC_IDN] >= 0.0) {
        AT(flux, n, k, j, i) = f[ORC_IDN] * wl[n * bb.ni + i];
      } else {
        AT(flux, n, k, j, i) = f[ORC_IDN] * wr[n * bb.ni + i];
      }
    }
  }
}

/* src/hydro/hydro.cpp:1025-1208.  Loop extents are the reference's, including the +-1
 * transverse extension of the x1 sweep (:1031-1039). */
void orc_calculate_fluxes(const orc_geom *g, int fluid, int recon, int riemann,
                          const orc_eos *eos, double c_h, const double *prim, double *flux1,
                          double *flux2, double *flux3) {
  const bounds_t bb = get_bounds(g);
  const int ndim = orc_ndim(g);
  const size_t pen = (size_t)g->nvar * bb.ni;
  double *wl = (double *)calloc(pen, sizeof(double));
  double *wr = (double *)calloc(pen, sizeof(double));
  double *wlb = (double *)calloc(pen, sizeof(double));
  const double gamma = eos->gamma;

  int jl = bb.js, ju = bb.je, kl = bb.ks, ku = bb.ke;
  if (g->nx[1] > 1) {
    if (g->nx[2] == 1) {
      jl = bb.js - 1, ju = bb.je + 1, kl = bb.ks, ku = bb.ke;
    } else {
      jl = bb.js - 1, ju = bb.je + 1, kl = bb.ks - 1, ku = bb.ke + 1;
    }
  }
  /* x1 */
  for (int k = kl; k <= ku; ++k) {
    for (int j = jl; j <= ju; ++j) {
      recon_pencil(g, &bb, recon, 1, prim, k, j, bb.is - 1, bb.ie + 1, wl, wr);
      riemann_pencil(g, &bb, fluid, riemann, ORC_IV1, gamma, c_h, wl, wr, k, j, bb.is,
                     bb.ie + 1, flux1);
    }
  }
  /* x2 : march in j (hydro.cpp:1100-1153) */
  if (ndim >= 2) {
    const int il = bb.is - 1, iu = bb.ie + 1;
    if (g->nx[2] == 1) {
      kl = bb.ks, ku = bb.ke;
    } else {
      kl = bb.ks - 1, ku = bb.ke + 1;
    }
    for (int k = kl; k <= ku; ++k) {
      double *pl = wl, *plb = wlb;
      for (int j = bb.js - 1; j <= bb.je + 1; ++j) {
        recon_pencil(g, &bb, recon, 2, prim, k, j, il, iu, plb, wr);
        if (j > bb.js - 1) {
          riemann_pencil(g, &bb, fluid, riemann, ORC_IV2, gamma, c_h, pl, wr, k, j, il, iu,
                         flux2);
        }
        double *t = pl;
        pl = plb;
        plb = t;
      }
    }
  }
  /* x3 : march in k (hydro.cpp:1156-1199) */
  if (ndim >= 3) {
    const int il = bb.is - 1, iu = bb.ie + 1;
    jl = bb.js - 1, ju = bb.je + 1;
    for (int j = jl; j <= ju; ++j) {
      double *pl = wl, *plb = wlb;
      for (int k = bb.ks - 1; k <= bb.ke + 1; ++k) {
        recon_pencil(g, &bb, recon, 3, prim, k, j, il, iu, plb, wr);
        if (k > bb.ks - 1) {
          riemann_pencil(g, &bb, fluid, riemann, ORC_IV3, gamma, c_h, pl, wr, k, j, il, iu,
                         flux3);
        }
        double *t = pl;
        pl = plb;
        plb = t;
      }
    }
  }
  free(wl);
  free(wr);
  free(wlb);
}

/* DC + LLF face flux straight from prim, incl. passive scalars
 * (hydro_dc_llf.hpp:43-142, glmmhd_dc_llf.hpp:46-179) */
static void llf_face(const orc_geom *g, const bounds_t *pb, int fluid, double gamma,
                     double c_h, const double *prim, int k, int j, int i, int ivx,
                     double *flux) {
  const bounds_t bb = *pb;
  const long off = (ivx == 1) ? 1 : ((ivx == 2) ? bb.sj : bb.sk);
  const int nh = g->nhydro;
  double l[ORC_NGLMMHD] = {0}, r[ORC_NGLMMHD] = {0}, f[ORC_NGLMMHD];
  for (int n = 0; n < nh; ++n) {
    r[n] = AT(prim, n, k, j, i);
    l[n] = (&AT(prim, n, k, j, i))[-off];
  }
  orc_riemann_point(fluid, ORC_RS_LLF, ivx, l, r, gamma, c_h, f);
  for (int n = 0; n < nh; ++n) AT(flux, n, k, j, i) = f[n];
  for (int n = nh; n < g->nvar; ++n) {
    if (f[ORC_IDN] >= 0.0) {
      AT(flux, n, k, j, i) = f[ORC_IDN] * (&AT(prim, n, k, j, i))[-off];
    } else {
      AT(flux, n, k, j, i) = f[ORC_IDN] * AT(prim, n, k, j, i);
    }
  }
}

/* src/hydro/hydro.cpp:980-1022 */
void orc_calculate_fluxes_tight(const orc_geom *g, int fluid, const orc_eos *eos, double c_h,
                                const double *prim, double *flux1, double *flux2,
                                double *flux3) {
  const bounds_t bb = get_bounds(g);
  const int ndim = orc_ndim(g);
  /* note: the reference loops to ke+1/je+1 also in collapsed dimensions' absence only
   * when the dimension is active; a collapsed dim has ks=ke=0 and no k+1 plane, so the
   * upper limit is clamped to the allocated extent. */
  const int kup = (ndim >= 3) ? bb.ke + 1 : bb.ke;
  const int jup = (ndim >= 2) ? bb.je + 1 : bb.je;
  for (int k = bb.ks; k <= kup; ++k)
    for (int j = bb.js; j <= jup; ++j)
      for (int i = bb.is; i <= bb.ie + 1; ++i) {
        llf_face(g, &bb, fluid, eos->gamma, c_h, prim, k, j, i, ORC_IV1, flux1);
        if (ndim >= 2) llf_face(g, &bb, fluid, eos->gamma, c_h, prim, k, j, i, ORC_IV2, flux2);
        if (ndim >= 3) llf_face(g, &bb, fluid, eos->gamma, c_h, prim, k, j, i, ORC_IV3, flux3);
      }
}

/* Parthenon Update::FluxDivHelper (un-vendored):
 * du = A1 F1(i+1) - A1 F1(i) [+ A2 ...][+ A3 ...]; returns -du / V.
 * UniformCartesian: A1 = dx2*dx3, A2 = dx1*dx3, A3 = dx1*dx2, V = dx1*dx2*dx3. */
static inline double flux_div(const bounds_t *pb, int ndim, const double *area, double vol,
                              const double *f1, const double *f2, const double *f3, int n,
                              int k, int j, int i) {
  const bounds_t bb = *pb;
  double du = (area[0] * AT(f1, n, k, j, i + 1) - area[0] * AT(f1, n, k, j, i));
  if (ndim >= 2) du += (area[1] * AT(f2, n, k, j + 1, i) - area[1] * AT(f2, n, k, j, i));
  if (ndim == 3) du += (area[2] * AT(f3, n, k + 1, j, i) - area[2] * AT(f3, n, k, j, i));
  return -du / vol;
}

static void areas(const orc_geom *g, double *area, double *vol) {
  area[0] = g->dx[1] * g->dx[2];
  area[1] = g->dx[0] * g->dx[2];
  area[2] = g->dx[0] * g->dx[1];
  *vol = g->dx[0] * g->dx[1] * g->dx[2];
}

/* Parthenon Update::UpdateWithFluxDivergence; call site hydro_driver.cpp:534-537 */
void orc_update_flux_div(const orc_geom *g, double *u0, const double *u1, const double *flux1,
                         const double *flux2, const double *flux3, double gam0, double gam1,
                         double beta_dt) {
  const bounds_t bb = get_bounds(g);
  const int ndim = orc_ndim(g);
  double area[3], vol;
  areas(g, area, &vol);
  for (int n = 0; n < g->nvar; ++n)
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          AT(u0, n, k, j, i) =
              gam0 * AT(u0, n, k, j, i) + gam1 * AT(u1, n, k, j, i) +
              beta_dt * flux_div(&bb, ndim, area, vol, flux1, flux2, flux3, n, k, j, i);
        }
}

/* src/hydro/glmmhd/dedner_source.cpp:17-75 */
void orc_dedner_source(const orc_geom *g, int extended, double alpha, double c_h,
                       double mindx, double beta_dt, double *cons, const double *prim) {
  const bounds_t bb = get_bounds(g);
  const double coeff = exp(-alpha * c_h * beta_dt / mindx);
  int ko = 1;
  if (orc_ndim(g) < 3) ko = 0;
  const int jo = (orc_ndim(g) < 2) ? 0 : 1; /* 1-D blocks have no j neighbours */
  for (int k = bb.ks; k <= bb.ke; ++k)
    for (int j = bb.js; j <= bb.je; ++j)
      for (int i = bb.is; i <= bb.ie; ++i) {
        if (extended) {
          const double divB =
              0.5 * ((AT(prim, ORC_IB1, k, j, i + 1) - AT(prim, ORC_IB1, k, j, i - 1)) /
                         g->dx[0] +
                     (AT(prim, ORC_IB2, k, j + jo, i) - AT(prim, ORC_IB2, k, j - jo, i)) /
                         g->dx[1] +
                     (AT(prim, ORC_IB3, k + ko, j, i) - AT(prim, ORC_IB3, k - ko, j, i)) /
                         g->dx[2]);
          AT(cons, ORC_IM1, k, j, i) -= beta_dt * divB * AT(prim, ORC_IB1, k, j, i);
          AT(cons, ORC_IM2, k, j, i) -= beta_dt * divB * AT(prim, ORC_IB2, k, j, i);
          AT(cons, ORC_IM3, k, j, i) -= beta_dt * divB * AT(prim, ORC_IB3, k, j, i);
          AT(cons, ORC_IEN, k, j, i) -=
              0.5 * beta_dt *
              (AT(prim, ORC_IB1, k, j, i) *
                   (AT(prim, ORC_IPS, k, j, i + 1) - AT(prim, ORC_IPS, k, j, i - 1)) /
                   g->dx[0] +
               AT(prim, ORC_IB2, k, j, i) *
                   (AT(prim, ORC_IPS, k, j + jo, i) - AT(prim, ORC_IPS, k, j - jo, i)) /
                   g->dx[1] +
               AT(prim, ORC_IB3, k, j, i) *
                   (AT(prim, ORC_IPS, k + ko, j, i) - AT(prim, ORC_IPS, k - ko, j, i)) /
                   g->dx[2]);
        }
        AT(cons, ORC_IPS, k, j, i) *= coeff;
      }
}

/* src/eos/adiabatic_hydro.hpp:52-142 and adiabatic_glmmhd.hpp:62-167 (one cell).
 * u/w are the nvar values of the cell; u may be modified (floors/ceilings). */
int orc_cons_to_prim_cell(int fluid, const orc_eos *eos, int nhydro, int nscalars, double *u,
                          double *w) {
  int status = 0;
  const double gm1 = eos->gamma - 1.0;
  const int mhd = (fluid == ORC_FLUID_GLMMHD);
  if (!(u[ORC_IDN] > 0.0 || eos->dfloor > 0.0)) status = 1;
  if (!(u[ORC_IDN] > eos->dfloor)) ORC_TRACE(ORC_TR_C2P_DFLOOR);
  u[ORC_IDN] = (u[ORC_IDN] > eos->dfloor) ? u[ORC_IDN] : eos->dfloor;
  w[ORC_IDN] = u[ORC_IDN];
  const double di = 1.0 / u[ORC_IDN];
  w[ORC_IV1] = u[ORC_IM1] * di;
  w[ORC_IV2] = u[ORC_IM2] * di;
  w[ORC_IV3] = u[ORC_IM3] * di;
  double e_B = 0.0;
  if (mhd) {
    w[ORC_IB1] = u[ORC_IB1];
    w[ORC_IB2] = u[ORC_IB2];
    w[ORC_IB3] = u[ORC_IB3];
    w[ORC_IPS] = u[ORC_IPS];
  }
  double e_k = 0.5 * di * (sq(u[ORC_IM1]) + sq(u[ORC_IM2]) + sq(u[ORC_IM3]));
  if (mhd) {
    e_B = 0.5 * (sq(u[ORC_IB1]) + sq(u[ORC_IB2]) + sq(u[ORC_IB3]));
    w[ORC_IPR] = gm1 * (u[ORC_IEN] - e_k - e_B);
  } else {
    w[ORC_IPR] = gm1 * (u[ORC_IEN] - e_k);
  }
  const double v2 = sq(w[ORC_IV1]) + sq(w[ORC_IV2]) + sq(w[ORC_IV3]);
  if (v2 > sq(eos->vceil)) {
    ORC_TRACE(ORC_TR_C2P_VCEIL);
    const double v = sqrt(v2);
    w[ORC_IV1] *= eos->vceil / v;
    w[ORC_IV2] *= eos->vceil / v;
    w[ORC_IV3] *= eos->vceil / v;
    u[ORC_IM1] *= eos->vceil / v;
    u[ORC_IM2] *= eos->vceil / v;
    u[ORC_IM3] *= eos->vceil / v;
    const double e_k_new = 0.5 * u[ORC_IDN] * sq(eos->vceil);
    u[ORC_IEN] -= e_k - e_k_new;
    e_k = e_k_new;
  }
  if (!(w[ORC_IPR] > 0.0 || eos->pfloor > 0.0 || eos->efloor > 0.0)) {
    if (status == 0) status = 2;
  }
  /* the reference writes (p/gm1) + e_k [+ e_B]: left-to-right addition */
  if ((eos->pfloor > 0.0) && (w[ORC_IPR] < eos->pfloor)) {
    ORC_TRACE(ORC_TR_C2P_PFLOOR);
    u[ORC_IEN] = mhd ? (eos->pfloor / gm1) + e_k + e_B : (eos->pfloor / gm1) + e_k;
    w[ORC_IPR] = eos->pfloor;
  }
  const double eff_floor = gm1 * u[ORC_IDN] * eos->efloor;
  if (w[ORC_IPR] < eff_floor) {
    ORC_TRACE(ORC_TR_C2P_EFLOOR);
    u[ORC_IEN] = mhd ? (u[ORC_IDN] * eos->efloor) + e_k + e_B : (u[ORC_IDN] * eos->efloor) + e_k;
    w[ORC_IPR] = eff_floor;
  }
  const double eff_ceil = gm1 * u[ORC_IDN] * eos->eceil;
  if (w[ORC_IPR] > eff_ceil) {
    ORC_TRACE(ORC_TR_C2P_ECEIL);
    u[ORC_IEN] = mhd ? (u[ORC_IDN] * eos->eceil) + e_k + e_B : (u[ORC_IDN] * eos->eceil) + e_k;
    w[ORC_IPR] = eff_ceil;
  }
  for (int n = nhydro; n < nhydro + nscalars; ++n) w[n] = u[n] * di;
  return status;
}

/* src/eos/adiabatic_hydro.cpp:33-55, adiabatic_glmmhd.cpp:33-56: ENTIRE block */
long orc_cons_to_prim(const orc_geom *g, int fluid, const orc_eos *eos, double *cons,
                      double *prim) {
  const bounds_t bb = get_bounds(g);
  long bad = 0;
  double u[64], w[64];
  for (int k = 0; k < bb.nk; ++k)
    for (int j = 0; j < bb.nj; ++j)
      for (int i = 0; i < bb.ni; ++i) {
        for (int n = 0; n < g->nvar; ++n) u[n] = AT(cons, n, k, j, i);
        const int st = orc_cons_to_prim_cell(fluid, eos, g->nhydro, g->nvar - g->nhydro, u, w);
        if (st) ++bad;
        for (int n = 0; n < g->nvar; ++n) {
          AT(cons, n, k, j, i) = u[n];
          AT(prim, n, k, j, i) = w[n];
        }
      }
  return bad;
}

/* src/hydro/hydro.cpp:828-896 (returns the un-scaled minimum; caller multiplies by cfl) */
double orc_estimate_dt_hyp(const orc_geom *g, int fluid, const orc_eos *eos,
                           const double *prim) {
  const bounds_t bb = get_bounds(g);
  const int ndim = orc_ndim(g);
  double min_dt = 1.7976931348623157e308;
  for (int k = bb.ks; k <= bb.ke; ++k)
    for (int j = bb.js; j <= bb.je; ++j)
      for (int i = bb.is; i <= bb.ie; ++i) {
        const double d = AT(prim, ORC_IDN, k, j, i);
        const double v1 = AT(prim, ORC_IV1, k, j, i);
        const double v2 = AT(prim, ORC_IV2, k, j, i);
        const double v3 = AT(prim, ORC_IV3, k, j, i);
        const double p = AT(prim, ORC_IPR, k, j, i);
        double lx, ly = 0.0, lz = 0.0;
        if (fluid == ORC_FLUID_EULER) {
          lx = orc_sound_speed(eos->gamma, d, p);
          ly = lx;
          lz = lx;
        } else {
          const double b1 = AT(prim, ORC_IB1, k, j, i);
          const double b2 = AT(prim, ORC_IB2, k, j, i);
          const double b3 = AT(prim, ORC_IB3, k, j, i);
          lx = orc_fast_speed(eos->gamma, d, p, b1, b2, b3);
          if (ndim > 1) ly = orc_fast_speed(eos->gamma, d, p, b2, b3, b1);
          if (ndim > 2) lz = orc_fast_speed(eos->gamma, d, p, b3, b1, b2);
        }
        min_dt = fmin(min_dt, g->dx[0] / (fabs(v1) + lx));
        if (ndim > 1) min_dt = fmin(min_dt, g->dx[1] / (fabs(v2) + ly));
        if (ndim > 2) min_dt = fmin(min_dt, g->dx[2] / (fabs(v3) + lz));
      }
  return min_dt;
}

/* src/hydro/hydro.cpp:1223-1342.  The reference kernel is racy by design (:1280-1284,
 * 1311-1314).  The oracle fixes ONE legal interleaving: within an attempt every cell is
 * checked against the fluxes as they stood at the start of the attempt, then the faces of
 * all flagged cells are recomputed (the LLF face value depends on prim only, so write
 * order is irrelevant).  The HIP path uses the same two-phase schedule. */
long orc_first_order_flux_correct(const orc_geom *g, int fluid, const orc_eos *eos,
                                  double c_h, const double *u0_cons, const double *u0_prim,
                                  const double *u1_cons, double *flux1, double *flux2,
                                  double *flux3, double gam0, double gam1, double beta_dt) {
  const bounds_t bb = get_bounds(g);
  const int ndim = orc_ndim(g);
  const int nh = g->nhydro;
  double area[3], vol;
  areas(g, area, &vol);
  unsigned char *mark = (unsigned char *)malloc((size_t)bb.sn);
  long total = 0, num_corrected;
  size_t attempts = 0;
  do {
    num_corrected = 0;
    memset(mark, 0, (size_t)bb.sn);
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          double nc[ORC_NGLMMHD];
          for (int v = 0; v < nh; ++v) {
            nc[v] = gam0 * AT(u0_cons, v, k, j, i) + gam1 * AT(u1_cons, v, k, j, i) +
                    beta_dt * flux_div(&bb, ndim, area, vol, flux1, flux2, flux3, v, k, j, i);
          }
          double new_p = nc[ORC_IEN] -
                         0.5 * (sq(nc[ORC_IM1]) + sq(nc[ORC_IM2]) + sq(nc[ORC_IM3])) / nc[ORC_IDN];
          if (fluid == ORC_FLUID_GLMMHD)
            new_p -= 0.5 * (sq(nc[ORC_IB1]) + sq(nc[ORC_IB2]) + sq(nc[ORC_IB3]));
          if (nc[ORC_IDN] > 0.0 && new_p > 0.0) continue;
          if (attempts > 2 && nc[ORC_IDN] > 0.0 && new_p < 0.0) continue; /* rely on floor */
          mark[k * bb.sk + j * bb.sj + i] = 1;
          num_corrected += 1;
        }
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          if (!mark[k * bb.sk + j * bb.sj + i]) continue;
          llf_face(g, &bb, fluid, eos->gamma, c_h, u0_prim, k, j, i, ORC_IV1, flux1);
          llf_face(g, &bb, fluid, eos->gamma, c_h, u0_prim, k, j, i + 1, ORC_IV1, flux1);
          if (ndim >= 2) {
            llf_face(g, &bb, fluid, eos->gamma, c_h, u0_prim, k, j, i, ORC_IV2, flux2);
            llf_face(g, &bb, fluid, eos->gamma, c_h, u0_prim, k, j + 1, i, ORC_IV2, flux2);
          }
          if (ndim >= 3) {
            llf_face(g, &bb, fluid, eos->gamma, c_h, u0_prim, k, j, i, ORC_IV3, flux3);
            llf_face(g, &bb, fluid, eos->gamma, c_h, u0_prim, k + 1, j, i, ORC_IV3, flux3);
          }
        }
    total += num_corrected;
    attempts += 1;
  } while (num_corrected > 0 && attempts < 4);
  free(mark);
  return total;
}

/* src/hydro/hydro.cpp:145-208 */
void orc_history(const orc_geom *g, int fluid, const double *cons, double *out) {
  const bounds_t bb = get_bounds(g);
  const int three_d = (orc_ndim(g) == 3);
  const double vol = g->dx[0] * g->dx[1] * g->dx[2];
  for (int q = 0; q < 8; ++q) out[q] = 0.0;
  const int jo = (orc_ndim(g) >= 2) ? 1 : 0;
  for (int k = bb.ks; k <= bb.ke; ++k)
    for (int j = bb.js; j <= bb.je; ++j)
      for (int i = bb.is; i <= bb.ie; ++i) {
        out[0] += AT(cons, ORC_IDN, k, j, i) * vol;
        out[1] += AT(cons, ORC_IM1, k, j, i) * vol;
        out[2] += AT(cons, ORC_IM2, k, j, i) * vol;
        out[3] += AT(cons, ORC_IM3, k, j, i) * vol;
        out[4] += 0.5 / AT(cons, ORC_IDN, k, j, i) *
                  (sq(AT(cons, ORC_IM1, k, j, i)) + sq(AT(cons, ORC_IM2, k, j, i)) +
                   sq(AT(cons, ORC_IM3, k, j, i))) *
                  vol;
        out[5] += AT(cons, ORC_IEN, k, j, i) * vol;
        if (fluid == ORC_FLUID_GLMMHD) {
          out[6] += 0.5 *
                    (sq(AT(cons, ORC_IB1, k, j, i)) + sq(AT(cons, ORC_IB2, k, j, i)) +
                     sq(AT(cons, ORC_IB3, k, j, i))) *
                    vol;
          double divb =
              (AT(cons, ORC_IB1, k, j, i + 1) - AT(cons, ORC_IB1, k, j, i - 1)) / g->dx[0] +
              (AT(cons, ORC_IB2, k, j + jo, i) - AT(cons, ORC_IB2, k, j - jo, i)) / g->dx[1];
          if (three_d)
            divb += (AT(cons, ORC_IB3, k + 1, j, i) - AT(cons, ORC_IB3, k - 1, j, i)) / g->dx[2];
          const double abs_b = sqrt(sq(AT(cons, ORC_IB1, k, j, i)) + sq(AT(cons, ORC_IB2, k, j, i)) +
                                    sq(AT(cons, ORC_IB3, k, j, i)));
          out[7] += (abs_b != 0) ? 0.5 *
                                       (sqrt(sq(g->dx[0]) + sq(g->dx[1]) + sq(g->dx[2]))) *
                                       fabs(divb) / abs_b * vol
                                 : 0;
        }
      }
}


/* ConsToPrim of m cells given as [m][nhydro] rows, with the branch mask of each (apk_oracle.h) */
void orc_c2p_many_traced(int fluid, const orc_eos *eos, long m, double *u, double *w, int *status, unsigned *masks) {
  const int nh = (fluid == ORC_FLUID_EULER) ? ORC_NHYDRO : ORC_NGLMMHD;
  for (long s = 0; s < m; ++s) {
    masks[s] = 0u;
    orc_trace_sink = masks + s;
    status[s] = orc_cons_to_prim_cell(fluid, eos, nh, 0, u + nh * s, w + nh * s);
  }
  orc_trace_sink = 0;
}
