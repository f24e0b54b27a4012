/*
 * amr.c -- ORACLE (test infrastructure only; see apk_oracle.h).
 * Restatement of the mesh-refinement operators AthenaPK registers for its conserved variables
 * (src/hydro/hydro.cpp:780-781) and of its block tagging criteria:
 *   - Hydro::refinement_ops::ProlongateCellMinModMultiD (src/hydro/prolongation/custom_ops.hpp:49-186),
 *     cell-centred case (el == CC): multi-D minmod slopes limited jointly so that no new extrema
 *     appear (Stone et al. 2020 eq. 5 + the AMReX limiter);
 *   - parthenon::refinement_ops::RestrictAverage (Parthenon, UN-VENDORED: volume-weighted
 *     average of the 2^d fine cells, pairwise summation order as in Athena++), also for face
 *     fluxes (area-weighted average of the 2^(d-1) fine faces: the coarse-fine flux correction
 *     of src/hydro/hydro_driver.cpp:527-531);
 *   - refinement::gradient::PressureGradient / VelocityGradient (src/refinement/gradient.cpp:18-99)
 *     and refinement::other::MaxDensity (src/refinement/other.cpp:18-44).
 * PARITY UNPINNED for the Parthenon-side helpers GetGridSpacings / GradMinMod and the
 * UniformCartesian coordinate formula Xc(i) = (xmin - is*dx) + (i + 0.5)*dx, which are recalled
 * from upstream Parthenon; the reference holds no vectors for these operators.
 */
#include "apk_oracle.h"

#include <math.h>

static double sign_(double a) { return (a < 0.) ? -1. : 1.; }

/* Parthenon refinement_ops::util::GradMinMod */
static double grad_minmod(double fc, double fm, double fp, double dxm, double dxp, double *gxm, double *gxp) {
  *gxm = (fc - fm) / dxm;
  *gxp = (fp - fc) / dxp;
  return 0.5 * (sign_(*gxm) + sign_(*gxp)) * fmin(fabs(*gxm), fabs(*gxp));
}

/* UniformCartesian::Xc of an index space whose first interior index is `s` */
static double xc_(double xmin, double dx, int s, int idx) { return (xmin - s * dx) + (idx + 0.5) * dx; }

/* Parthenon refinement_ops::util::GetGridSpacings<DIM, CC> */
static void grid_spacings(const orc_refine_geom *r, int d, int ci, int fi, double *dxm, double *dxp, double *dxfm,
                          double *dxfp) {
  const double cdx = 2.0 * r->dx[d];
  const int cs = r->cng, fs = r->ng;
  const double xm = xc_(r->xmin[d], cdx, cs, ci - 1);
  const double xc = xc_(r->xmin[d], cdx, cs, ci);
  const double xp = xc_(r->xmin[d], cdx, cs, ci + 1);
  *dxm = xc - xm;
  *dxp = xp - xc;
  const double fxm = xc_(r->xmin[d], r->dx[d], fs, fi);
  const double fxp = xc_(r->xmin[d], r->dx[d], fs, fi + 1);
  *dxfm = xc - fxm;
  *dxfp = fxp - xc;
}

int orc_refine_ndim(const orc_refine_geom *r) { return (r->nx[2] > 1) ? 3 : ((r->nx[1] > 1) ? 2 : 1); }
void orc_refine_dims(const orc_refine_geom *r, int fine[3], int coarse[3]) {
  for (int d = 0; d < 3; ++d) {
    const int act = (d == 0) || r->nx[d] > 1;
    fine[d] = act ? r->nx[d] + 2 * r->ng : 1;
    coarse[d] = act ? r->nx[d] / 2 + 2 * r->cng : 1;
  }
}

/* custom_ops.hpp:60-183, for every variable and every coarse cell of the inclusive index box */
void orc_prolongate_minmod(const orc_refine_geom *r, int nvar, const double *coarse, double *fine, const int lo[3],
                           const int hi[3]) {
  int fn[3], cn[3];
  orc_refine_dims(r, fn, cn);
  const int DIM = orc_refine_ndim(r);
  const long csj = cn[0], csk = (long)cn[0] * cn[1], csn = csk * cn[2];
  const long fsj = fn[0], fsk = (long)fn[0] * fn[1], fsn = fsk * fn[2];
  const int cis = r->cng, cjs = (DIM > 1) ? r->cng : 0, cks = (DIM > 2) ? r->cng : 0;
  const int is = r->ng, js = (DIM > 1) ? r->ng : 0, ks = (DIM > 2) ? r->ng : 0;
  for (int v = 0; v < nvar; ++v)
    for (int k = lo[2]; k <= hi[2]; ++k)
      for (int j = lo[1]; j <= hi[1]; ++j)
        for (int i = lo[0]; i <= hi[0]; ++i) {
          const double *c = coarse + v * csn;
          double *f = fine + v * fsn;
#define CO(kk, jj, ii) c[(kk)*csk + (jj)*csj + (ii)]
#define FI(kk, jj, ii) f[(kk)*fsk + (jj)*fsj + (ii)]
          const int fi = (i - cis) * 2 + is;
          const int fj = (DIM > 1) ? (j - cjs) * 2 + js : js;
          const int fk = (DIM > 2) ? (k - cks) * 2 + ks : ks;
          const double fc = CO(k, j, i);
          double dx1fm = 0, dx1fp = 0, gx1m = 0, gx1p = 0, gx1c = 0;
          {
            double dx1m, dx1p;
            grid_spacings(r, 0, i, fi, &dx1m, &dx1p, &dx1fm, &dx1fp);
            gx1c = grad_minmod(fc, CO(k, j, i - 1), CO(k, j, i + 1), dx1m, dx1p, &gx1m, &gx1p);
          }
          double dx2fm = 0, dx2fp = 0, gx2m = 0, gx2p = 0, gx2c = 0;
          if (DIM > 1) {
            double dx2m, dx2p;
            grid_spacings(r, 1, j, fj, &dx2m, &dx2p, &dx2fm, &dx2fp);
            gx2c = grad_minmod(fc, CO(k, j - 1, i), CO(k, j + 1, i), dx2m, dx2p, &gx2m, &gx2p);
          }
          double dx3fm = 0, dx3fp = 0, gx3m = 0, gx3p = 0, gx3c = 0;
          if (DIM > 2) {
            double dx3m, dx3p;
            grid_spacings(r, 2, k, fk, &dx3m, &dx3p, &dx3fm, &dx3fp);
            /* the reference passes dx3p as the last (output) argument; it is not read afterwards */
            gx3c = grad_minmod(fc, CO(k - 1, j, i), CO(k + 1, j, i), dx3m, dx3p, &gx3m, &gx3p);
          }
          (void)gx1m; (void)gx1p; (void)gx2m; (void)gx2p; (void)gx3m; (void)gx3p;
          double dqmax = fabs(gx1c) * fmax(dx1fm, dx1fp);
          int jlim = 0, klim = 0;
          if (DIM > 1) {
            dqmax += fabs(gx2c) * fmax(dx2fm, dx2fp);
            jlim = 1;
          }
          if (DIM > 2) {
            dqmax += fabs(gx3c) * fmax(dx3fm, dx3fp);
            klim = 1;
          }
          double qmin = fc, qmax = fc;
          for (int koff = -klim; koff <= klim; koff++)
            for (int joff = -jlim; joff <= jlim; joff++)
              for (int ioff = -1; ioff <= 1; ioff++) {
                qmin = fmin(qmin, CO(k + koff, j + joff, i + ioff));
                qmax = fmax(qmax, CO(k + koff, j + joff, i + ioff));
              }
          double alpha = 1.0;
          if (dqmax * alpha > (qmax - fc)) alpha = (qmax - fc) / dqmax;
          if (dqmax * alpha > (fc - qmin)) alpha = (fc - qmin) / dqmax;
          gx1c *= alpha;
          gx2c *= alpha;
          gx3c *= alpha;
          FI(fk, fj, fi) = fc - (gx1c * dx1fm + gx2c * dx2fm + gx3c * dx3fm);
          FI(fk, fj, fi + 1) = fc + (gx1c * dx1fp - gx2c * dx2fm - gx3c * dx3fm);
          if (DIM > 1) {
            FI(fk, fj + 1, fi) = fc - (gx1c * dx1fm - gx2c * dx2fp + gx3c * dx3fm);
            FI(fk, fj + 1, fi + 1) = fc + (gx1c * dx1fp + gx2c * dx2fp - gx3c * dx3fm);
          }
          if (DIM > 2) {
            FI(fk + 1, fj, fi) = fc - (gx1c * dx1fm + gx2c * dx2fm - gx3c * dx3fp);
            FI(fk + 1, fj, fi + 1) = fc + (gx1c * dx1fp - gx2c * dx2fm + gx3c * dx3fp);
            FI(fk + 1, fj + 1, fi) = fc - (gx1c * dx1fm - gx2c * dx2fp - gx3c * dx3fp);
            FI(fk + 1, fj + 1, fi + 1) = fc + (gx1c * dx1fp + gx2c * dx2fp + gx3c * dx3fp);
          }
#undef CO
#undef FI
        }
}

/* RestrictAverage for element `el`: 0 = cell centres (volume weights), 1/2/3 = x1/x2/x3 faces
 * (area weights; the face index along `el` is not averaged).  Weights are those of the uniform
 * Cartesian grid, summed pairwise like the upstream operator. */
void orc_restrict_average(const orc_refine_geom *r, int nvar, int el, const double *fine, double *coarse,
                          const int lo[3], const int hi[3]) {
  int fn[3], cn[3];
  orc_refine_dims(r, fn, cn);
  const int DIM = orc_refine_ndim(r);
  /* face arrays carry one extra entry along their own direction */
  if (el >= 1) {
    fn[el - 1] += 1;
    cn[el - 1] += 1;
  }
  const long csj = cn[0], csk = (long)cn[0] * cn[1], csn = csk * cn[2];
  const long fsj = fn[0], fsk = (long)fn[0] * fn[1], fsn = fsk * fn[2];
  const int cs[3] = {r->cng, (DIM > 1) ? r->cng : 0, (DIM > 2) ? r->cng : 0};
  const int fs[3] = {r->ng, (DIM > 1) ? r->ng : 0, (DIM > 2) ? r->ng : 0};
  /* offsets that are averaged over */
  const int oi1 = (el != 1) ? 1 : 0;
  const int oj1 = (DIM > 1 && el != 2) ? 1 : 0;
  const int ok1 = (DIM > 2 && el != 3) ? 1 : 0;
  double w = 1.0; /* Volume<el>: product of the widths of the averaged directions */
  if (el != 1) w *= r->dx[0];
  if (el != 2) w *= r->dx[1];
  if (el != 3) w *= r->dx[2];
  for (int v = 0; v < nvar; ++v)
    for (int k = lo[2]; k <= hi[2]; ++k)
      for (int j = lo[1]; j <= hi[1]; ++j)
        for (int i = lo[0]; i <= hi[0]; ++i) {
          const int fi = (i - cs[0]) * 2 + fs[0];
          const int fj = (DIM > 1) ? (j - cs[1]) * 2 + fs[1] : 0;
          const int fk = (DIM > 2) ? (k - cs[2]) * 2 + fs[2] : 0;
          double vol[2][2][2], terms[2][2][2];
          for (int ok = 0; ok < 2; ++ok)
            for (int oj = 0; oj < 2; ++oj)
              for (int oi = 0; oi < 2; ++oi) {
                vol[ok][oj][oi] = terms[ok][oj][oi] = 0.0;
                if (ok > ok1 || oj > oj1 || oi > oi1) continue;
                vol[ok][oj][oi] = w;
                terms[ok][oj][oi] = w * fine[v * fsn + (fk + ok) * fsk + (fj + oj) * fsj + (fi + oi)];
              }
          const double tvol = ((vol[0][0][0] + vol[0][1][0]) + (vol[0][0][1] + vol[0][1][1])) +
                              ((vol[1][0][0] + vol[1][1][0]) + (vol[1][0][1] + vol[1][1][1]));
          coarse[v * csn + k * csk + j * csj + i] =
              (((terms[0][0][0] + terms[0][1][0]) + (terms[0][0][1] + terms[0][1][1])) +
               ((terms[1][0][0] + terms[1][1][0]) + (terms[1][0][1] + terms[1][1][1]))) /
              tvol;
        }
}

/* tagging: returns +1 refine, -1 derefine, 0 same; *crit receives the reduced criterion */
static int tag_of(double v, double refine_above, double derefine_below) {
  if (v > refine_above) return 1;
  if (v < derefine_below) return -1;
  return 0;
}

/* gradient.cpp:18-61 (threshold, 0.25 threshold) */
int orc_tag_pressure_gradient(const orc_geom *g, const double *prim, double threshold, double *crit) {
  const int ni = orc_ni(g), nj = orc_nj(g);
  const int ndim = (g->nx[2] > 1) ? 3 : ((g->nx[1] > 1) ? 2 : 1);
  const long sn = (long)ni * nj * orc_nk(g);
  const double *p = prim + ORC_IPR * sn;
  double maxeps = 0.0;
  if (ndim == 1) {
    *crit = 0.0;
    return 0;
  }
  const int is = g->ng, js = g->ng, ks = (ndim == 3) ? g->ng : 0;
#define P(k, j, i) p[((long)(k)*nj + (j)) * ni + (i)]
  if (ndim == 3) {
    for (int k = ks - 1; k <= ks + g->nx[2]; ++k)
      for (int j = js - 1; j <= js + g->nx[1]; ++j)
        for (int i = is - 1; i <= is + g->nx[0]; ++i) {
          const double a = 0.5 * (P(k, j, i + 1) - P(k, j, i - 1)), b = 0.5 * (P(k, j + 1, i) - P(k, j - 1, i)),
                       c = 0.5 * (P(k + 1, j, i) - P(k - 1, j, i));
          const double eps = sqrt(a * a + b * b + c * c) / P(k, j, i);
          maxeps = fmax(maxeps, eps);
        }
  } else {
    const int k = ks;
    for (int j = js - 1; j <= js + g->nx[1]; ++j)
      for (int i = is - 1; i <= is + g->nx[0]; ++i) {
        const double a = 0.5 * (P(k, j, i + 1) - P(k, j, i - 1)), b = 0.5 * (P(k, j + 1, i) - P(k, j - 1, i));
        const double eps = sqrt(a * a + b * b) / P(k, j, i);
        maxeps = fmax(maxeps, eps);
      }
  }
#undef P
  *crit = maxeps;
  return tag_of(maxeps, threshold, 0.25 * threshold);
}

/* gradient.cpp:64-96 (threshold, 0.5 threshold); k over the interior only */
int orc_tag_velocity_gradient(const orc_geom *g, const double *prim, double threshold, double *crit) {
  const int ni = orc_ni(g), nj = orc_nj(g);
  const int ndim = (g->nx[2] > 1) ? 3 : ((g->nx[1] > 1) ? 2 : 1);
  const long sn = (long)ni * nj * orc_nk(g);
  const double *v1 = prim + ORC_IV1 * sn, *v2 = prim + ORC_IV2 * sn;
  const int is = g->ng, js = (ndim > 1) ? g->ng : 0, ks = (ndim == 3) ? g->ng : 0;
  double vgmax = 0.0;
  for (int k = ks; k < ks + g->nx[2]; ++k)
    for (int j = js - 1; j <= js + g->nx[1]; ++j)
      for (int i = is - 1; i <= is + g->nx[0]; ++i) {
        const long c = ((long)k * nj + j) * ni + i;
        const double vgy = fabs(v2[c + 1] - v2[c - 1]) * 0.5;
        const double vgx = fabs(v1[c + ni] - v1[c - ni]) * 0.5;
        const double vg = sqrt(vgx * vgx + vgy * vgy);
        if (vg > vgmax) vgmax = vg;
      }
  *crit = vgmax;
  return tag_of(vgmax, threshold, 0.5 * threshold);
}

/* other.cpp:18-44; the i range runs to ib.e + 1 as written there */
int orc_tag_max_density(const orc_geom *g, const double *prim, double refine_above, double deref_below, double *crit) {
  const int ni = orc_ni(g), nj = orc_nj(g);
  const int ndim = (g->nx[2] > 1) ? 3 : ((g->nx[1] > 1) ? 2 : 1);
  const int is = g->ng, js = (ndim > 1) ? g->ng : 0, ks = (ndim == 3) ? g->ng : 0;
  double maxrho = 0.0;
  for (int k = ks; k < ks + g->nx[2]; ++k)
    for (int j = js; j < js + g->nx[1]; ++j)
      for (int i = is; i <= is + g->nx[0]; ++i) maxrho = fmax(maxrho, prim[((long)k * nj + j) * ni + i]);
  *crit = maxrho;
  return tag_of(maxrho, refine_above, deref_below);
}
