/*
 * riemann.c -- ORACLE (test infrastructure only; see apk_oracle.h).
 * Plain-C restatement of the reference's 1-D approximate Riemann solvers.  All solvers
 * work on direction-permuted local states (w[IV1] = normal velocity, w[IB1] = normal
 * field) exactly as the reference does, and un-permute on store.  Floating-point
 * operation order follows the reference; build with -ffp-contract=off.
 */
#include "apk_oracle.h"

#include <math.h>
#include <string.h>

static inline double sq(double x) { return x * x; }
static inline double dmin(double a, double b) { return (b < a) ? b : a; }
static inline double dmax(double a, double b) { return (a < b) ? b : a; }

/* src/eos/adiabatic_hydro.hpp:43-45 */
double orc_sound_speed(double gamma, double d, double p) { return sqrt(gamma * p / d); }

/* src/eos/adiabatic_glmmhd.hpp:47-54 */
double orc_fast_speed(double gamma, double d, double p, double bx, double by, double bz) {
  const double asq = gamma * p;
  const double ct2 = by * by + bz * bz;
  const double qsq = bx * bx + ct2 + asq;
  const double tmp = bx * bx + ct2 - asq;
  return sqrt(0.5 * (qsq + sqrt(tmp * tmp + 4.0 * asq * ct2)) / d);
}

/* index permutation shared by all solvers (e.g. hydro_hlle.hpp:46-47, glmmhd_hlld.hpp:45-49) */
typedef struct {
  int vx, vy, vz, bx, by, bz;
} perm_t;

static perm_t make_perm(int ivx) {
  perm_t p;
  p.vx = ivx;
  p.vy = ORC_IV1 + ((ivx - ORC_IV1) + 1) % 3;
  p.vz = ORC_IV1 + ((ivx - ORC_IV1) + 2) % 3;
  p.bx = p.vx - 1 + ORC_NHYDRO;
  p.by = p.vy - 1 + ORC_NHYDRO;
  p.bz = p.vz - 1 + ORC_NHYDRO;
  return p;
}

static void load_hydro(const perm_t *pm, const double *w, double *loc) {
  loc[ORC_IDN] = w[ORC_IDN];
  loc[ORC_IV1] = w[pm->vx];
  loc[ORC_IV2] = w[pm->vy];
  loc[ORC_IV3] = w[pm->vz];
  loc[ORC_IPR] = w[ORC_IPR];
}

static void load_mhd(const perm_t *pm, const double *w, double *loc) {
  load_hydro(pm, w, loc);
  loc[ORC_IB1] = w[pm->bx];
  loc[ORC_IB2] = w[pm->by];
  loc[ORC_IB3] = w[pm->bz];
  loc[ORC_IPS] = w[ORC_IPS];
}

static void store_hydro(const perm_t *pm, const double *f, double *flux) {
  flux[ORC_IDN] = f[ORC_IDN];
  flux[pm->vx] = f[ORC_IV1];
  flux[pm->vy] = f[ORC_IV2];
  flux[pm->vz] = f[ORC_IV3];
  flux[ORC_IEN] = f[ORC_IEN];
}

static void store_mhd(const perm_t *pm, const double *f, double *flux) {
  store_hydro(pm, f, flux);
  flux[pm->bx] = f[ORC_IB1];
  flux[pm->by] = f[ORC_IB2];
  flux[pm->bz] = f[ORC_IB3];
  flux[ORC_IPS] = f[ORC_IPS];
}

/* ------------------------------------------------------------------------------------ */
/* src/hydro/rsolvers/hydro_hlle.hpp:40-138 */
static void hydro_hlle(const double *wl, const double *wr, double gamma, double *f) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0 / gm1;
  double roe[ORC_NHYDRO], fl[ORC_NHYDRO], fr[ORC_NHYDRO];

  const double sdl = sqrt(wl[ORC_IDN]);
  const double sdr = sqrt(wr[ORC_IDN]);
  const double isum = 1.0 / (sdl + sdr);
  roe[ORC_IDN] = sdl * sdr;
  roe[ORC_IV1] = (sdl * wl[ORC_IV1] + sdr * wr[ORC_IV1]) * isum;
  roe[ORC_IV2] = (sdl * wl[ORC_IV2] + sdr * wr[ORC_IV2]) * isum;
  roe[ORC_IV3] = (sdl * wl[ORC_IV3] + sdr * wr[ORC_IV3]) * isum;

  const double el = wl[ORC_IPR] * igm1 +
                    0.5 * wl[ORC_IDN] * (sq(wl[ORC_IV1]) + sq(wl[ORC_IV2]) + sq(wl[ORC_IV3]));
  const double er = wr[ORC_IPR] * igm1 +
                    0.5 * wr[ORC_IDN] * (sq(wr[ORC_IV1]) + sq(wr[ORC_IV2]) + sq(wr[ORC_IV3]));
  const double hroe = ((el + wl[ORC_IPR]) / sdl + (er + wr[ORC_IPR]) / sdr) * isum;

  const double cl = orc_sound_speed(gamma, wl[ORC_IDN], wl[ORC_IPR]);
  const double cr = orc_sound_speed(gamma, wr[ORC_IDN], wr[ORC_IPR]);
  const double q = hroe - 0.5 * (sq(roe[ORC_IV1]) + sq(roe[ORC_IV2]) + sq(roe[ORC_IV3]));
  const double a = (q < 0.0) ? 0.0 : sqrt(gm1 * q);

  const double al = dmin((roe[ORC_IV1] - a), (wl[ORC_IV1] - cl));
  const double ar = dmax((roe[ORC_IV1] + a), (wr[ORC_IV1] + cr));
  const double bp = ar > 0.0 ? ar : ORC_TINY_NUMBER;
  const double bm = al < 0.0 ? al : ORC_TINY_NUMBER; /* hydro HLLE uses +TINY: :97-98 */

  const double vxl = wl[ORC_IV1] - bm;
  const double vxr = wr[ORC_IV1] - bp;
  fl[ORC_IDN] = wl[ORC_IDN] * vxl;
  fr[ORC_IDN] = wr[ORC_IDN] * vxr;
  fl[ORC_IV1] = wl[ORC_IDN] * wl[ORC_IV1] * vxl;
  fr[ORC_IV1] = wr[ORC_IDN] * wr[ORC_IV1] * vxr;
  fl[ORC_IV2] = wl[ORC_IDN] * wl[ORC_IV2] * vxl;
  fr[ORC_IV2] = wr[ORC_IDN] * wr[ORC_IV2] * vxr;
  fl[ORC_IV3] = wl[ORC_IDN] * wl[ORC_IV3] * vxl;
  fr[ORC_IV3] = wr[ORC_IDN] * wr[ORC_IV3] * vxr;
  fl[ORC_IV1] += wl[ORC_IPR];
  fr[ORC_IV1] += wr[ORC_IPR];
  fl[ORC_IEN] = el * vxl + wl[ORC_IPR] * wl[ORC_IV1];
  fr[ORC_IEN] = er * vxr + wr[ORC_IPR] * wr[ORC_IV1];

  double tmp = 0.0;
  if (bp != bm) tmp = 0.5 * (bp + bm) / (bp - bm);
  else ORC_TRACE(ORC_TR_HLLE_BP_EQ_BM);
  for (int n = 0; n < ORC_NHYDRO; ++n) f[n] = 0.5 * (fl[n] + fr[n]) + (fl[n] - fr[n]) * tmp;
}

/* ------------------------------------------------------------------------------------ */
/* src/hydro/rsolvers/hydro_hllc.hpp:32-157 */
static void hydro_hllc(const double *wl, const double *wr, double gamma, double *f) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0 / gm1;
  double fl[ORC_NHYDRO], fr[ORC_NHYDRO];

  const double cl = orc_sound_speed(gamma, wl[ORC_IDN], wl[ORC_IPR]);
  const double cr = orc_sound_speed(gamma, wr[ORC_IDN], wr[ORC_IPR]);
  const double el = wl[ORC_IPR] * igm1 +
                    0.5 * wl[ORC_IDN] * (sq(wl[ORC_IV1]) + sq(wl[ORC_IV2]) + sq(wl[ORC_IV3]));
  const double er = wr[ORC_IPR] * igm1 +
                    0.5 * wr[ORC_IDN] * (sq(wr[ORC_IV1]) + sq(wr[ORC_IV2]) + sq(wr[ORC_IV3]));
  const double rhoa = .5 * (wl[ORC_IDN] + wr[ORC_IDN]);
  const double ca = .5 * (cl + cr);
  const double pmid =
      .5 * (wl[ORC_IPR] + wr[ORC_IPR] + (wl[ORC_IV1] - wr[ORC_IV1]) * rhoa * ca);

  const double ql =
      (pmid <= wl[ORC_IPR])
          ? 1.0
          : sqrt(1.0 + (gamma + 1) / (2 * gamma) * (pmid / wl[ORC_IPR] - 1.0));
  const double qr =
      (pmid <= wr[ORC_IPR])
          ? 1.0
          : sqrt(1.0 + (gamma + 1) / (2 * gamma) * (pmid / wr[ORC_IPR] - 1.0));

  const double al = wl[ORC_IV1] - cl * ql;
  const double ar = wr[ORC_IV1] + cr * qr;
  const double bp = ar > 0.0 ? ar : (ORC_TINY_NUMBER);
  const double bm = al < 0.0 ? al : -(ORC_TINY_NUMBER);

  double vxl = wl[ORC_IV1] - al;
  double vxr = wr[ORC_IV1] - ar;
  const double tl = wl[ORC_IPR] + vxl * wl[ORC_IDN] * wl[ORC_IV1];
  const double tr = wr[ORC_IPR] + vxr * wr[ORC_IDN] * wr[ORC_IV1];
  const double ml = wl[ORC_IDN] * vxl;
  const double mr = -(wr[ORC_IDN] * vxr);
  const double am = (tl - tr) / (ml + mr);
  double cp = (ml * tr + mr * tl) / (ml + mr);
  if (!(cp > 0.0)) ORC_TRACE(ORC_TR_HLLC_CP_CLIP);
  if (ql != 1.0) ORC_TRACE(ORC_TR_HLLC_QL);
  if (qr != 1.0) ORC_TRACE(ORC_TR_HLLC_QR);
  cp = cp > 0.0 ? cp : 0.0;

  vxl = wl[ORC_IV1] - bm;
  vxr = wr[ORC_IV1] - bp;
  fl[ORC_IDN] = wl[ORC_IDN] * vxl;
  fr[ORC_IDN] = wr[ORC_IDN] * vxr;
  fl[ORC_IV1] = wl[ORC_IDN] * wl[ORC_IV1] * vxl + wl[ORC_IPR];
  fr[ORC_IV1] = wr[ORC_IDN] * wr[ORC_IV1] * vxr + wr[ORC_IPR];
  fl[ORC_IV2] = wl[ORC_IDN] * wl[ORC_IV2] * vxl;
  fr[ORC_IV2] = wr[ORC_IDN] * wr[ORC_IV2] * vxr;
  fl[ORC_IV3] = wl[ORC_IDN] * wl[ORC_IV3] * vxl;
  fr[ORC_IV3] = wr[ORC_IDN] * wr[ORC_IV3] * vxr;
  fl[ORC_IEN] = el * vxl + wl[ORC_IPR] * wl[ORC_IV1];
  fr[ORC_IEN] = er * vxr + wr[ORC_IPR] * wr[ORC_IV1];

  double sl, sr, sm;
  if (am >= 0.0) {
    ORC_TRACE(ORC_TR_HLLC_AM_POS);
    sl = am / (am - bm);
    sr = 0.0;
    sm = -bm / (am - bm);
  } else {
    sl = 0.0;
    sr = -am / (bp - am);
    sm = bp / (bp - am);
  }
  f[ORC_IDN] = sl * fl[ORC_IDN] + sr * fr[ORC_IDN];
  f[ORC_IV1] = sl * fl[ORC_IV1] + sr * fr[ORC_IV1] + sm * cp;
  f[ORC_IV2] = sl * fl[ORC_IV2] + sr * fr[ORC_IV2];
  f[ORC_IV3] = sl * fl[ORC_IV3] + sr * fr[ORC_IV3];
  f[ORC_IEN] = sl * fl[ORC_IEN] + sr * fr[ORC_IEN] + sm * cp * am;
}

/* ------------------------------------------------------------------------------------ */
/* src/hydro/rsolvers/hydro_dc_llf.hpp:43-142 (states are already first-order L/R) */
static void hydro_llf(const double *wl, const double *wr, double gamma, double *f) {
  const double igm1 = 1.0 / (gamma - 1.0);
  double qa = wl[ORC_IDN] * wl[ORC_IV1];
  double qb = wr[ORC_IDN] * wr[ORC_IV1];
  const double fs_d = qa + qb;
  double fs_mx = qa * wl[ORC_IV1] + qb * wr[ORC_IV1];
  const double fs_my = qa * wl[ORC_IV2] + qb * wr[ORC_IV2];
  const double fs_mz = qa * wl[ORC_IV3] + qb * wr[ORC_IV3];
  const double el = wl[ORC_IPR] * igm1 +
                    0.5 * wl[ORC_IDN] * (sq(wl[ORC_IV1]) + sq(wl[ORC_IV2]) + sq(wl[ORC_IV3]));
  const double er = wr[ORC_IPR] * igm1 +
                    0.5 * wr[ORC_IDN] * (sq(wr[ORC_IV1]) + sq(wr[ORC_IV2]) + sq(wr[ORC_IV3]));
  fs_mx += (wl[ORC_IPR] + wr[ORC_IPR]);
  const double fs_e = (el + wl[ORC_IPR]) * wl[ORC_IV1] + (er + wr[ORC_IPR]) * wr[ORC_IV1];

  qa = orc_sound_speed(gamma, wl[ORC_IDN], wl[ORC_IPR]);
  qb = orc_sound_speed(gamma, wr[ORC_IDN], wr[ORC_IPR]);
  const double a = fmax((fabs(wl[ORC_IV1]) + qa), (fabs(wr[ORC_IV1]) + qb));

  const double du_d = a * (wr[ORC_IDN] - wl[ORC_IDN]);
  const double du_mx = a * (wr[ORC_IDN] * wr[ORC_IV1] - wl[ORC_IDN] * wl[ORC_IV1]);
  const double du_my = a * (wr[ORC_IDN] * wr[ORC_IV2] - wl[ORC_IDN] * wl[ORC_IV2]);
  const double du_mz = a * (wr[ORC_IDN] * wr[ORC_IV3] - wl[ORC_IDN] * wl[ORC_IV3]);
  const double du_e = a * (er - el);

  f[ORC_IDN] = 0.5 * (fs_d - du_d);
  f[ORC_IV1] = 0.5 * (fs_mx - du_mx);
  f[ORC_IV2] = 0.5 * (fs_my - du_my);
  f[ORC_IV3] = 0.5 * (fs_mz - du_mz);
  f[ORC_IEN] = 0.5 * (fs_e - du_e);
}

/* ------------------------------------------------------------------------------------ */
/* GLM decoupled 2x2 system shared by all MHD solvers (glmmhd_hlld.hpp:88-92) */
static void glm_interface(const double *wl, const double *wr, double c_h, double *bxi,
                          double *psii) {
  *bxi = 0.5 * (wl[ORC_IB1] + wr[ORC_IB1]) - 0.5 / c_h * (wr[ORC_IPS] - wl[ORC_IPS]);
  *psii = 0.5 * (wl[ORC_IPS] + wr[ORC_IPS]) - 0.5 * c_h * (wr[ORC_IB1] - wl[ORC_IB1]);
}

/* src/hydro/rsolvers/glmmhd_hlle.hpp:27-192 */
static void glmmhd_hlle(const double *wl, const double *wr, double gamma, double c_h,
                        double *f) {
  const double gm1 = gamma - 1.0;
  double roe[ORC_NGLMMHD], fl[ORC_NGLMMHD], fr[ORC_NGLMMHD];
  double bxi, psii;
  glm_interface(wl, wr, c_h, &bxi, &psii);
  f[ORC_IB1] = psii;
  f[ORC_IPS] = sq(c_h) * bxi;

  const double sdl = sqrt(wl[ORC_IDN]);
  const double sdr = sqrt(wr[ORC_IDN]);
  const double isum = 1.0 / (sdl + sdr);
  roe[ORC_IDN] = sdl * sdr;
  roe[ORC_IV1] = (sdl * wl[ORC_IV1] + sdr * wr[ORC_IV1]) * isum;
  roe[ORC_IV2] = (sdl * wl[ORC_IV2] + sdr * wr[ORC_IV2]) * isum;
  roe[ORC_IV3] = (sdl * wl[ORC_IV3] + sdr * wr[ORC_IV3]) * isum;
  roe[ORC_IB2] = (sdr * wl[ORC_IB2] + sdl * wr[ORC_IB2]) * isum;
  roe[ORC_IB3] = (sdr * wl[ORC_IB3] + sdl * wr[ORC_IB3]) * isum;
  const double x = 0.5 * (sq(wl[ORC_IB2] - wr[ORC_IB2]) + sq(wl[ORC_IB3] - wr[ORC_IB3])) /
                   (sq(sdl + sdr));
  const double y = 0.5 * (wl[ORC_IDN] + wr[ORC_IDN]) / roe[ORC_IDN];

  const double pbl = 0.5 * (bxi * bxi + sq(wl[ORC_IB2]) + sq(wl[ORC_IB3]));
  const double pbr = 0.5 * (bxi * bxi + sq(wr[ORC_IB2]) + sq(wr[ORC_IB3]));
  const double el = wl[ORC_IPR] / gm1 +
                    0.5 * wl[ORC_IDN] * (sq(wl[ORC_IV1]) + sq(wl[ORC_IV2]) + sq(wl[ORC_IV3])) +
                    pbl;
  const double er = wr[ORC_IPR] / gm1 +
                    0.5 * wr[ORC_IDN] * (sq(wr[ORC_IV1]) + sq(wr[ORC_IV2]) + sq(wr[ORC_IV3])) +
                    pbr;
  const double hroe =
      ((el + wl[ORC_IPR] + pbl) / sdl + (er + wr[ORC_IPR] + pbr) / sdr) * isum;

  const double cl =
      orc_fast_speed(gamma, wl[ORC_IDN], wl[ORC_IPR], wl[ORC_IB1], wl[ORC_IB2], wl[ORC_IB3]);
  const double cr =
      orc_fast_speed(gamma, wr[ORC_IDN], wr[ORC_IPR], wr[ORC_IB1], wr[ORC_IB2], wr[ORC_IB3]);

  const double btsq = sq(roe[ORC_IB2]) + sq(roe[ORC_IB3]);
  const double vaxsq = bxi * bxi / roe[ORC_IDN];
  const double bt_starsq = (gm1 - (gm1 - 1.0) * y) * btsq;
  const double hp = hroe - (vaxsq + btsq / roe[ORC_IDN]);
  const double vsq = sq(roe[ORC_IV1]) + sq(roe[ORC_IV2]) + sq(roe[ORC_IV3]);
  const double twid_asq = dmax((gm1 * (hp - 0.5 * vsq) - (gm1 - 1.0) * x), 0.0);
  const double ct2 = bt_starsq / roe[ORC_IDN];
  const double tsum = vaxsq + ct2 + twid_asq;
  const double tdif = vaxsq + ct2 - twid_asq;
  const double cf2_cs2 = sqrt(tdif * tdif + 4.0 * twid_asq * ct2);
  const double cfsq = 0.5 * (tsum + cf2_cs2);
  const double a = sqrt(cfsq);

  const double al = dmin((roe[ORC_IV1] - a), (wl[ORC_IV1] - cl));
  const double ar = dmax((roe[ORC_IV1] + a), (wr[ORC_IV1] + cr));
  const double bp = ar > 0.0 ? ar : 0.0; /* MHD HLLE uses 0.0: :134-135 */
  const double bm = al < 0.0 ? al : 0.0;

  const double vxl = wl[ORC_IV1] - bm;
  const double vxr = wr[ORC_IV1] - bp;
  fl[ORC_IDN] = wl[ORC_IDN] * vxl;
  fr[ORC_IDN] = wr[ORC_IDN] * vxr;
  fl[ORC_IV1] = wl[ORC_IDN] * wl[ORC_IV1] * vxl + pbl - sq(bxi);
  fr[ORC_IV1] = wr[ORC_IDN] * wr[ORC_IV1] * vxr + pbr - sq(bxi);
  fl[ORC_IV2] = wl[ORC_IDN] * wl[ORC_IV2] * vxl - bxi * wl[ORC_IB2];
  fr[ORC_IV2] = wr[ORC_IDN] * wr[ORC_IV2] * vxr - bxi * wr[ORC_IB2];
  fl[ORC_IV3] = wl[ORC_IDN] * wl[ORC_IV3] * vxl - bxi * wl[ORC_IB3];
  fr[ORC_IV3] = wr[ORC_IDN] * wr[ORC_IV3] * vxr - bxi * wr[ORC_IB3];
  fl[ORC_IV1] += wl[ORC_IPR];
  fr[ORC_IV1] += wr[ORC_IPR];
  fl[ORC_IEN] = el * vxl + wl[ORC_IV1] * (wl[ORC_IPR] + pbl - bxi * bxi);
  fr[ORC_IEN] = er * vxr + wr[ORC_IV1] * (wr[ORC_IPR] + pbr - bxi * bxi);
  fl[ORC_IEN] -= bxi * (wl[ORC_IB2] * wl[ORC_IV2] + wl[ORC_IB3] * wl[ORC_IV3]);
  fr[ORC_IEN] -= bxi * (wr[ORC_IB2] * wr[ORC_IV2] + wr[ORC_IB3] * wr[ORC_IV3]);
  fl[ORC_IB2] = wl[ORC_IB2] * vxl - bxi * wl[ORC_IV2];
  fr[ORC_IB2] = wr[ORC_IB2] * vxr - bxi * wr[ORC_IV2];
  fl[ORC_IB3] = wl[ORC_IB3] * vxl - bxi * wl[ORC_IV3];
  fr[ORC_IB3] = wr[ORC_IB3] * vxr - bxi * wr[ORC_IV3];

  double tmp = 0.0;
  if (bp != bm) tmp = 0.5 * (bp + bm) / (bp - bm);
  else ORC_TRACE(ORC_TR_HLLE_BP_EQ_BM);
  static const int comps[7] = {ORC_IDN, ORC_IV1, ORC_IV2, ORC_IV3, ORC_IEN, ORC_IB2, ORC_IB3};
  for (int c = 0; c < 7; ++c) {
    const int n = comps[c];
    f[n] = 0.5 * (fl[n] + fr[n]) + (fl[n] - fr[n]) * tmp;
  }
}

/* ------------------------------------------------------------------------------------ */
/* HLLD (src/hydro/rsolvers/glmmhd_hlld.hpp:39-396).  The reference spells out the left
 * and right sides separately; they are the same arithmetic, stated here once per side. */
typedef struct {
  double d, mx, my, mz, e, by, bz;
} c1d; /* glmmhd_hlld.hpp:32-34 */

#define HLLD_SMALL 1.0e-8 /* glmmhd_hlld.hpp:36 */

/* star state of one side: eqns (39),(43)-(48) of Miyoshi & Kusano; glmmhd_hlld.hpp:187-250 */
static void hlld_star_side(const double *w, const c1d *u, double sd, double sdm,
                           double sdm_inv, double sm, double pt, double ptst, double bxi,
                           double bxsq, c1d *ust, double ust_d_inv, double *vbst, unsigned trace_bit) {
  ust->mx = ust->d * sm;
  if (fabs(u->d * sd * sdm - bxsq) < (HLLD_SMALL)*ptst) {
    ORC_TRACE(trace_bit);
    ust->my = ust->d * w[ORC_IV2];
    ust->mz = ust->d * w[ORC_IV3];
    ust->by = u->by;
    ust->bz = u->bz;
  } else {
    double tmp = bxi * (sd - sdm) / (u->d * sd * sdm - bxsq);
    ust->my = ust->d * (w[ORC_IV2] - u->by * tmp);
    ust->mz = ust->d * (w[ORC_IV3] - u->bz * tmp);
    tmp = (u->d * sq(sd) - bxsq) / (u->d * sd * sdm - bxsq);
    ust->by = u->by * tmp;
    ust->bz = u->bz * tmp;
  }
  *vbst = (ust->mx * bxi + (ust->my * ust->by + ust->mz * ust->bz)) * ust_d_inv;
  ust->e = (sd * u->e - pt * w[ORC_IV1] + ptst * sm +
            bxi * (w[ORC_IV1] * bxi + (w[ORC_IV2] * u->by + w[ORC_IV3] * u->bz) - *vbst)) *
           sdm_inv;
}

static void cons_from_prim_1d(const double *w, double igm1, double bxsq, c1d *u, double *pb) {
  /* glmmhd_hlld.hpp:95-118 */
  *pb = 0.5 * (bxsq + (sq(w[ORC_IB2]) + sq(w[ORC_IB3])));
  const double ke = 0.5 * w[ORC_IDN] * (sq(w[ORC_IV1]) + (sq(w[ORC_IV2]) + sq(w[ORC_IV3])));
  u->d = w[ORC_IDN];
  u->mx = w[ORC_IV1] * u->d;
  u->my = w[ORC_IV2] * u->d;
  u->mz = w[ORC_IV3] * u->d;
  u->e = w[ORC_IPR] * igm1 + ke + *pb;
  u->by = w[ORC_IB2];
  u->bz = w[ORC_IB3];
}

static void phys_flux_1d(const double *w, const c1d *u, double pt, double bxi, double bxsq,
                         c1d *fx) {
  /* glmmhd_hlld.hpp:141-155 */
  fx->d = u->mx;
  fx->mx = u->mx * w[ORC_IV1] + pt - bxsq;
  fx->my = u->my * w[ORC_IV1] - bxi * u->by;
  fx->mz = u->mz * w[ORC_IV1] - bxi * u->bz;
  fx->e = w[ORC_IV1] * (u->e + pt - bxsq) - bxi * (w[ORC_IV2] * u->by + w[ORC_IV3] * u->bz);
  fx->by = u->by * w[ORC_IV1] - bxi * w[ORC_IV2];
  fx->bz = u->bz * w[ORC_IV1] - bxi * w[ORC_IV3];
}

/* a <- s * (a - b), componentwise (glmmhd_hlld.hpp:297-327) */
static void jump_scale(c1d *a, const c1d *b, double s) {
  a->d = s * (a->d - b->d);
  a->mx = s * (a->mx - b->mx);
  a->my = s * (a->my - b->my);
  a->mz = s * (a->mz - b->mz);
  a->e = s * (a->e - b->e);
  a->by = s * (a->by - b->by);
  a->bz = s * (a->bz - b->bz);
}

static void glmmhd_hlld(const double *wl, const double *wr, double gamma, double c_h,
                        double *f) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0 / gm1;
  double spd[5];
  c1d ul, ur, ulst, uldst, urdst, urst, fl, fr;
  double bxi, psii;
  glm_interface(wl, wr, c_h, &bxi, &psii);
  f[ORC_IB1] = psii;
  f[ORC_IPS] = sq(c_h) * bxi;

  const double bxsq = bxi * bxi;
  double pbl, pbr;
  cons_from_prim_1d(wl, igm1, bxsq, &ul, &pbl);
  cons_from_prim_1d(wr, igm1, bxsq, &ur, &pbr);

  /* step 2: fast speeds from the RECONSTRUCTED normal field, not bxi (:122-125) */
  const double cfl =
      orc_fast_speed(gamma, wl[ORC_IDN], wl[ORC_IPR], wl[ORC_IB1], wl[ORC_IB2], wl[ORC_IB3]);
  const double cfr =
      orc_fast_speed(gamma, wr[ORC_IDN], wr[ORC_IPR], wr[ORC_IB1], wr[ORC_IB2], wr[ORC_IB3]);
  spd[0] = dmin(wl[ORC_IV1] - cfl, wr[ORC_IV1] - cfr);
  spd[4] = dmax(wl[ORC_IV1] + cfl, wr[ORC_IV1] + cfr);

  /* step 3 */
  const double ptl = wl[ORC_IPR] + pbl;
  const double ptr = wr[ORC_IPR] + pbr;
  phys_flux_1d(wl, &ul, ptl, bxi, bxsq, &fl);
  phys_flux_1d(wr, &ur, ptr, bxi, bxsq, &fr);

  /* step 4 */
  const double sdl = spd[0] - wl[ORC_IV1];
  const double sdr = spd[4] - wr[ORC_IV1];
  spd[2] = (sdr * ur.mx - sdl * ul.mx + (ptl - ptr)) / (sdr * ur.d - sdl * ul.d);
  const double sdml = spd[0] - spd[2];
  const double sdmr = spd[4] - spd[2];
  const double sdml_inv = 1.0 / sdml;
  const double sdmr_inv = 1.0 / sdmr;
  ulst.d = ul.d * sdl * sdml_inv;
  urst.d = ur.d * sdr * sdmr_inv;
  const double ulst_d_inv = 1.0 / ulst.d;
  const double urst_d_inv = 1.0 / urst.d;
  const double sqrtdl = sqrt(ulst.d);
  const double sqrtdr = sqrt(urst.d);
  spd[1] = spd[2] - fabs(bxi) / sqrtdl;
  spd[3] = spd[2] + fabs(bxi) / sqrtdr;

  /* step 5 */
  const double ptstl = ptl + ul.d * sdl * (spd[2] - wl[ORC_IV1]);
  const double ptstr = ptr + ur.d * sdr * (spd[2] - wr[ORC_IV1]);
  const double ptst = 0.5 * (ptstr + ptstl);

  double vbstl, vbstr;
  hlld_star_side(wl, &ul, sdl, sdml, sdml_inv, spd[2], ptl, ptst, bxi, bxsq, &ulst,
                 ulst_d_inv, &vbstl, ORC_TR_HLLD_DEG_L);
  hlld_star_side(wr, &ur, sdr, sdmr, sdmr_inv, spd[2], ptr, ptst, bxi, bxsq, &urst,
                 urst_d_inv, &vbstr, ORC_TR_HLLD_DEG_R);

  if (0.5 * bxsq < (HLLD_SMALL)*ptst) {
    ORC_TRACE(ORC_TR_HLLD_DEG_DST);
    uldst = ulst;
    urdst = urst;
  } else {
    const double invsumd = 1.0 / (sqrtdl + sqrtdr);
    const double bxsig = (bxi > 0.0 ? 1.0 : -1.0);
    uldst.d = ulst.d;
    urdst.d = urst.d;
    uldst.mx = ulst.mx;
    urdst.mx = urst.mx;
    double tmp = invsumd * (sqrtdl * (ulst.my * ulst_d_inv) + sqrtdr * (urst.my * urst_d_inv) +
                            bxsig * (urst.by - ulst.by));
    uldst.my = uldst.d * tmp;
    urdst.my = urdst.d * tmp;
    tmp = invsumd * (sqrtdl * (ulst.mz * ulst_d_inv) + sqrtdr * (urst.mz * urst_d_inv) +
                     bxsig * (urst.bz - ulst.bz));
    uldst.mz = uldst.d * tmp;
    urdst.mz = urdst.d * tmp;
    tmp = invsumd * (sqrtdl * urst.by + sqrtdr * ulst.by +
                     bxsig * sqrtdl * sqrtdr *
                         ((urst.my * urst_d_inv) - (ulst.my * ulst_d_inv)));
    uldst.by = urdst.by = tmp;
    tmp = invsumd * (sqrtdl * urst.bz + sqrtdr * ulst.bz +
                     bxsig * sqrtdl * sqrtdr *
                         ((urst.mz * urst_d_inv) - (ulst.mz * ulst_d_inv)));
    uldst.bz = urdst.bz = tmp;
    tmp = spd[2] * bxi + (uldst.my * uldst.by + uldst.mz * uldst.bz) / uldst.d;
    uldst.e = ulst.e - sqrtdl * bxsig * (vbstl - tmp);
    urdst.e = urst.e + sqrtdr * bxsig * (vbstr - tmp);
  }

  /* step 6: jumps across the waves, in the reference's order (dst before st) */
  jump_scale(&uldst, &ulst, spd[1]);
  jump_scale(&ulst, &ul, spd[0]);
  jump_scale(&urdst, &urst, spd[3]);
  jump_scale(&urst, &ur, spd[4]);

  c1d r;
  if (spd[0] >= 0.0) {
    ORC_TRACE(ORC_TR_HLLD_FL);
    r = fl;
  } else if (spd[4] <= 0.0) {
    ORC_TRACE(ORC_TR_HLLD_FR);
    r = fr;
  } else if (spd[1] >= 0.0) {
    ORC_TRACE(ORC_TR_HLLD_LSTAR);
    r.d = fl.d + ulst.d;
    r.mx = fl.mx + ulst.mx;
    r.my = fl.my + ulst.my;
    r.mz = fl.mz + ulst.mz;
    r.e = fl.e + ulst.e;
    r.by = fl.by + ulst.by;
    r.bz = fl.bz + ulst.bz;
  } else if (spd[2] >= 0.0) {
    ORC_TRACE(ORC_TR_HLLD_LDSTAR);
    r.d = fl.d + ulst.d + uldst.d;
    r.mx = fl.mx + ulst.mx + uldst.mx;
    r.my = fl.my + ulst.my + uldst.my;
    r.mz = fl.mz + ulst.mz + uldst.mz;
    r.e = fl.e + ulst.e + uldst.e;
    r.by = fl.by + ulst.by + uldst.by;
    r.bz = fl.bz + ulst.bz + uldst.bz;
  } else if (spd[3] > 0.0) {
    ORC_TRACE(ORC_TR_HLLD_RDSTAR);
    r.d = fr.d + urst.d + urdst.d;
    r.mx = fr.mx + urst.mx + urdst.mx;
    r.my = fr.my + urst.my + urdst.my;
    r.mz = fr.mz + urst.mz + urdst.mz;
    r.e = fr.e + urst.e + urdst.e;
    r.by = fr.by + urst.by + urdst.by;
    r.bz = fr.bz + urst.bz + urdst.bz;
  } else {
    ORC_TRACE(ORC_TR_HLLD_RSTAR);
    r.d = fr.d + urst.d;
    r.mx = fr.mx + urst.mx;
    r.my = fr.my + urst.my;
    r.mz = fr.mz + urst.mz;
    r.e = fr.e + urst.e;
    r.by = fr.by + urst.by;
    r.bz = fr.bz + urst.bz;
  }
  f[ORC_IDN] = r.d;
  f[ORC_IV1] = r.mx;
  f[ORC_IV2] = r.my;
  f[ORC_IV3] = r.mz;
  f[ORC_IEN] = r.e;
  f[ORC_IB2] = r.by;
  f[ORC_IB3] = r.bz;
}

/* ------------------------------------------------------------------------------------ */
/* src/hydro/rsolvers/glmmhd_dc_llf.hpp:46-179 */
static void glmmhd_llf(const double *wl, const double *wr, double gamma, double c_h,
                       double *f) {
  const double igm1 = 1.0 / (gamma - 1.0);
  double bxi, psii;
  glm_interface(wl, wr, c_h, &bxi, &psii);

  double qa = wl[ORC_IDN] * wl[ORC_IV1];
  double qb = wr[ORC_IDN] * wr[ORC_IV1];
  const double qc = 0.5 * (sq(wl[ORC_IB2]) + sq(wl[ORC_IB3]) - sq(bxi));
  const double qd = 0.5 * (sq(wr[ORC_IB2]) + sq(wr[ORC_IB3]) - sq(bxi));

  const double fs_d = qa + qb;
  double fs_mx = qa * wl[ORC_IV1] + qb * wr[ORC_IV1] + qc + qd;
  const double fs_my = qa * wl[ORC_IV2] + qb * wr[ORC_IV2] - bxi * (wl[ORC_IB2] + wr[ORC_IB2]);
  const double fs_mz = qa * wl[ORC_IV3] + qb * wr[ORC_IV3] - bxi * (wl[ORC_IB3] + wr[ORC_IB3]);
  const double fs_by =
      wl[ORC_IB2] * wl[ORC_IV1] + wr[ORC_IB2] * wr[ORC_IV1] - bxi * (wl[ORC_IV2] + wr[ORC_IV2]);
  const double fs_bz =
      wl[ORC_IB3] * wl[ORC_IV1] + wr[ORC_IB3] * wr[ORC_IV1] - bxi * (wl[ORC_IV3] + wr[ORC_IV3]);

  const double el = wl[ORC_IPR] * igm1 +
                    0.5 * wl[ORC_IDN] * (sq(wl[ORC_IV1]) + sq(wl[ORC_IV2]) + sq(wl[ORC_IV3])) +
                    qc + sq(bxi);
  const double er = wr[ORC_IPR] * igm1 +
                    0.5 * wr[ORC_IDN] * (sq(wr[ORC_IV1]) + sq(wr[ORC_IV2]) + sq(wr[ORC_IV3])) +
                    qd + sq(bxi);
  fs_mx += (wl[ORC_IPR] + wr[ORC_IPR]);
  double fs_e = (el + wl[ORC_IPR] + qc) * wl[ORC_IV1] + (er + wr[ORC_IPR] + qd) * wr[ORC_IV1];
  fs_e -= bxi * (wl[ORC_IB2] * wl[ORC_IV2] + wl[ORC_IB3] * wl[ORC_IV3]);
  fs_e -= bxi * (wr[ORC_IB2] * wr[ORC_IV2] + wr[ORC_IB3] * wr[ORC_IV3]);

  qa = orc_fast_speed(gamma, wl[ORC_IDN], wl[ORC_IPR], wl[ORC_IB1], wl[ORC_IB2], wl[ORC_IB3]);
  qb = orc_fast_speed(gamma, wr[ORC_IDN], wr[ORC_IPR], wr[ORC_IB1], wr[ORC_IB2], wr[ORC_IB3]);
  const double a = fmax((fabs(wl[ORC_IV1]) + qa), (fabs(wr[ORC_IV1]) + qb));

  const double du_d = a * (wr[ORC_IDN] - wl[ORC_IDN]);
  const double du_mx = a * (wr[ORC_IDN] * wr[ORC_IV1] - wl[ORC_IDN] * wl[ORC_IV1]);
  const double du_my = a * (wr[ORC_IDN] * wr[ORC_IV2] - wl[ORC_IDN] * wl[ORC_IV2]);
  const double du_mz = a * (wr[ORC_IDN] * wr[ORC_IV3] - wl[ORC_IDN] * wl[ORC_IV3]);
  const double du_e = a * (er - el);
  const double du_by = a * (wr[ORC_IB2] - wl[ORC_IB2]);
  const double du_bz = a * (wr[ORC_IB3] - wl[ORC_IB3]);

  f[ORC_IDN] = 0.5 * (fs_d - du_d);
  f[ORC_IV1] = 0.5 * (fs_mx - du_mx);
  f[ORC_IV2] = 0.5 * (fs_my - du_my);
  f[ORC_IV3] = 0.5 * (fs_mz - du_mz);
  f[ORC_IEN] = 0.5 * (fs_e - du_e);
  f[ORC_IB1] = psii;
  f[ORC_IB2] = 0.5 * (fs_by - du_by);
  f[ORC_IB3] = 0.5 * (fs_bz - du_bz);
  f[ORC_IPS] = sq(c_h) * bxi;
}

/* ------------------------------------------------------------------------------------ */
void orc_riemann_point(int fluid, int riemann, int ivx, const double *wl, const double *wr,
                       double gamma, double c_h, double *flux) {
  const perm_t pm = make_perm(ivx);
  double l[ORC_NGLMMHD], r[ORC_NGLMMHD], f[ORC_NGLMMHD];
  if (fluid == ORC_FLUID_EULER) {
    load_hydro(&pm, wl, l);
    load_hydro(&pm, wr, r);
    switch (riemann) {
    case ORC_RS_HLLE: hydro_hlle(l, r, gamma, f); break;
    case ORC_RS_HLLC: hydro_hllc(l, r, gamma, f); break;
    case ORC_RS_LLF: hydro_llf(l, r, gamma, f); break;
    case ORC_RS_NONE: /* rsolvers.hpp:35-48 */
      for (int n = 0; n < ORC_NHYDRO; ++n) f[n] = 0.0;
      break;
    default:
      for (int n = 0; n < ORC_NHYDRO; ++n) f[n] = NAN;
    }
    store_hydro(&pm, f, flux);
  } else {
    load_mhd(&pm, wl, l);
    load_mhd(&pm, wr, r);
    switch (riemann) {
    case ORC_RS_HLLE: glmmhd_hlle(l, r, gamma, c_h, f); break;
    case ORC_RS_HLLD: glmmhd_hlld(l, r, gamma, c_h, f); break;
    case ORC_RS_LLF: glmmhd_llf(l, r, gamma, c_h, f); break;
    case ORC_RS_NONE: /* rsolvers.hpp:50-63 */
      for (int n = 0; n < ORC_NGLMMHD; ++n) f[n] = 0.0;
      break;
    default:
      for (int n = 0; n < ORC_NGLMMHD; ++n) f[n] = NAN;
    }
    store_mhd(&pm, f, flux);
  }
}

void orc_riemann_many(int fluid, int riemann, int ivx, long m, const double *wl,
                      const double *wr, double gamma, double c_h, double *flux) {
  const int nv = (fluid == ORC_FLUID_EULER) ? ORC_NHYDRO : ORC_NGLMMHD;
  for (long s = 0; s < m; ++s)
    orc_riemann_point(fluid, riemann, ivx, wl + nv * s, wr + nv * s, gamma, c_h,
                      flux + nv * s);
}


void orc_riemann_many_traced(int fluid, int riemann, int ivx, long m, const double *wl, const double *wr, double gamma,
                             double c_h, double *flux, unsigned *masks) {
  const int nv = (fluid == ORC_FLUID_EULER) ? ORC_NHYDRO : ORC_NGLMMHD;
  for (long s = 0; s < m; ++s) {
    masks[s] = 0u;
    orc_trace_sink = masks + s;
    orc_riemann_point(fluid, riemann, ivx, wl + nv * s, wr + nv * s, gamma, c_h, flux + nv * s);
  }
  orc_trace_sink = 0;
}
