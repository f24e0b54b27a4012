/*
 * recon.c -- ORACLE (test infrastructure only; see apk_oracle.h).
 * Plain-C restatement of the reference's pointwise reconstruction operators.  Operation
 * order and grouping parentheses follow the reference exactly (the "KGF" groupings exist
 * for floating-point symmetry); compile with -ffp-contract=off for the canonical result.
 */
#include "apk_oracle.h"

#include <float.h>
#include <math.h>

static inline double sq(double x) { return x * x; }
/* Parthenon SIGN macro (un-vendored): SIGN(0) = +1 */
static inline double sgn(double x) { return (x < 0.0) ? -1.0 : 1.0; }
static inline double dmin(double a, double b) { return (b < a) ? b : a; } /* std::min */
static inline double dmax(double a, double b) { return (a < b) ? b : a; } /* std::max */

/* src/recon/plm_simple.hpp:21-37 */
void orc_plm(double qm1, double q0, double qp1, double *ql_ip1, double *qr_i) {
  const double dl = q0 - qm1;
  const double dr = qp1 - q0;
  const double prod = dl * dr;
  double slope = 0.0;
  if (prod > 0.0) slope = prod / (dl + dr);
  *ql_ip1 = q0 + slope;
  *qr_i = q0 - slope;
}

/* One interface of PPM step 2a (src/recon/ppm_simple.hpp:66-98): the i-1/2 and i+1/2
 * blocks of the reference are the same arithmetic on (lo,hi) = (q_im1,q_i) resp.
 * (q_i,q_ip1).  Returns the possibly limited interface value. */
static double ppm_limit_interface(double qlo, double qhi, double face, double d2lo,
                                  double d2hi, unsigned trace_bit) {
  const double C2 = 1.25;
  const double below = face - qlo; /* (CD eq 84a) */
  const double above = qhi - face; /* (CD eq 84b) */
  const double d2f = 3.0 * (qlo + qhi - 2.0 * face);
  double lim = 0.0;
  if (sgn(d2f) == sgn(d2lo) && sgn(d2f) == sgn(d2hi)) {
    lim = sgn(d2f) * dmin(C2 * fabs(d2lo), dmin(C2 * fabs(d2hi), fabs(d2f)));
  }
  const double alt = 0.5 * (qlo + qhi) - lim / 6.0;
  if (below * above < 0.0) {
    ORC_TRACE(trace_bit);
    return alt; /* local extremum at this face */
  }
  return face;
}

/* src/recon/ppm_simple.hpp:39-162 */
void orc_ppm(double qm2, double qm1, double q0, double qp1, double qp2, double *ql_ip1,
             double *qr_i) {
  const double C2 = 1.25;
  /* step 1: CW eq 1.6 interface averages */
  const double da = q0 - qm1;
  const double db = qp1 - q0;
  const double dd_m = 0.5 * da + 0.5 * (qm1 - qm2);
  const double dd_c = 0.5 * db + 0.5 * da;
  const double dd_p = 0.5 * (qp2 - qp1) + 0.5 * db;
  double face_m = 0.5 * (qm1 + q0) + (dd_m - dd_c) / 6.0;
  double face_p = 0.5 * (q0 + qp1) + (dd_c - dd_p) / 6.0;

  /* step 2a: second derivatives (no 1/2), off-centred terms added first */
  const double d2_m = qm2 + q0 - 2.0 * qm1;
  const double d2_c = qm1 + qp1 - 2.0 * q0;
  const double d2_p = q0 + qp2 - 2.0 * qp1;
  face_m = ppm_limit_interface(qm1, q0, face_m, d2_m, d2_c, ORC_TR_PPM_LIM_M);
  face_p = ppm_limit_interface(q0, qp1, face_p, d2_c, d2_p, ORC_TR_PPM_LIM_P);

  const double d2_face = 6.0 * (face_m + face_p - 2.0 * q0);

  double qr = face_m;
  double ql = face_p;

  /* step 3 */
  const double dminus = q0 - qr;
  const double dplus = ql - q0;

  /* step 4: CS limiters on the parabola */
  const double ext_a = dminus * dplus;
  const double ext_b = (qp1 - q0) * (q0 - qm1);

  double d2lim = 0.0;
  if (sgn(d2_m) == sgn(d2_c) && sgn(d2_m) == sgn(d2_p) && sgn(d2_m) == sgn(d2_face)) {
    d2lim = sgn(d2_face) * dmin(dmin(C2 * fabs(d2_m), C2 * fabs(d2_c)),
                                dmin(C2 * fabs(d2_p), fabs(d2_face)));
  }
  const double scale_lo = dmax(fabs(qm1), fabs(qm2));
  const double scale_hi = dmax(dmax(fabs(q0), fabs(qp1)), fabs(qp2));
  double ratio = 0.0;
  if (fabs(d2_face) > (1.0e-12) * dmax(scale_lo, scale_hi)) ratio = d2lim / d2_face;
  else if (ext_a <= 0.0 || ext_b <= 0.0) ORC_TRACE(ORC_TR_PPM_ROUNDOFF);

  const double ext_m = q0 - ratio * dminus;
  const double ext_p = q0 + ratio * dplus;
  const double over_m = q0 - 2.0 * dplus;
  const double over_p = q0 + 2.0 * dminus;

  if (ext_a <= 0.0 || ext_b <= 0.0) {
    ORC_TRACE(ORC_TR_PPM_EXTREMUM);
    if (ratio <= (1.0 - (1.0e-12))) {
      qr = ext_m;
      ql = ext_p;
    } else {
      ORC_TRACE(ORC_TR_PPM_RATIO_BIG);
    }
  } else {
    if (fabs(dminus) >= 2.0 * fabs(dplus)) {
      ORC_TRACE(ORC_TR_PPM_OVER_M);
      qr = over_m;
    }
    if (fabs(dplus) >= 2.0 * fabs(dminus)) {
      ORC_TRACE(ORC_TR_PPM_OVER_P);
      ql = over_p;
    }
  }
  *ql_ip1 = ql;
  *qr_i = qr;
}

/* src/recon/wenoz_simple.hpp:28-81 */
void orc_wenoz(double qm2, double qm1, double q0, double qp1, double qp2, double *ql_ip1,
               double *qr_i) {
  const double c0 = 13. / 12., c1 = 0.25;
  const double b0 = c0 * sq(qm2 + q0 - 2.0 * qm1) + c1 * sq(qm2 + 3.0 * q0 - 4.0 * qm1);
  const double b1 = c0 * sq(qm1 + qp1 - 2.0 * q0) + c1 * sq(qm1 - qp1);
  const double b2 = c0 * sq(qp2 + q0 - 2.0 * qp1) + c1 * sq(qp2 + 3.0 * q0 - 4.0 * qp1);
  const double eps = 1.0e-42;
  const double tau5 = fabs(b0 - b2);
  const double i0 = tau5 / (b0 + eps);
  const double i1 = tau5 / (b1 + eps);
  const double i2 = tau5 / (b2 + eps);

  double f0 = (2.0 * qm2 - 7.0 * qm1 + 11.0 * q0);
  double f1 = (-1.0 * qm1 + 5.0 * q0 + 2.0 * qp1);
  double f2 = (2.0 * q0 + 5.0 * qp1 - qp2);
  double a0 = 0.1 * (1.0 + sq(i0));
  double a1 = 0.6 * (1.0 + sq(i1));
  double a2 = 0.3 * (1.0 + sq(i2));
  double asum = 6.0 * (a0 + a1 + a2);
  *ql_ip1 = (f0 * a0 + f1 * a1 + f2 * a2) / asum;

  f0 = (2.0 * qp2 - 7.0 * qp1 + 11.0 * q0);
  f1 = (-1.0 * qp1 + 5.0 * q0 + 2.0 * qm1);
  f2 = (2.0 * q0 + 5.0 * qm1 - qm2);
  a0 = 0.1 * (1.0 + sq(i2));
  a1 = 0.6 * (1.0 + sq(i1));
  a2 = 0.3 * (1.0 + sq(i0));
  asum = 6.0 * (a0 + a1 + a2);
  *qr_i = (f0 * a0 + f1 * a1 + f2 * a2) / asum;
}

/* src/recon/weno3_simple.hpp:26-63 */
void orc_weno3(double qm1, double q0, double qp1, double dx2, double *ql_ip1,
               double *qr_i) {
  const double bp = sq(qp1 - q0);
  const double bm = sq(q0 - qm1);
  const double tau = sq(qp1 - 2.0 * q0 + qm1);
  const double ip = tau / (bp + dx2);
  const double im = tau / (bm + dx2);

  double f0 = q0 + qp1;
  double f1 = -qm1 + 3.0 * q0;
  double a0 = (1.0 + ip) * 2.0 / 3.0;
  double a1 = (1.0 + im) / 3.0;
  double asum = 2.0 * (a0 + a1);
  *ql_ip1 = (a0 * f0 + a1 * f1) / asum;

  f0 = q0 + qm1;
  f1 = -qp1 + 3.0 * q0;
  a0 = (1.0 + im) * 2.0 / 3.0;
  a1 = (1.0 + ip) / 3.0;
  asum = 2.0 * (a0 + a1);
  *qr_i = (a0 * f0 + a1 * f1) / asum;
}

/* src/hydro/diffusion/diffusion.hpp:37-47 */
static double minmod(double a, double b) {
  if (a * b > 0) {
    if (a > 0) return dmin(a, b);
    return dmax(a, b);
  }
  return 0.0;
}

/* src/recon/limo3_simple.hpp:27-58 */
static double limo3_limiter(double dvp, double dvm, double dx) {
  const double r = 0.1;
  const double eps = 10.0 * DBL_EPSILON;
  const double theta = dvm / (dvp + ORC_TINY_NUMBER);
  const double q = (2.0 + theta) / 3.0;
  const double phi =
      dmax(0.0, dmin(q, dmax(-0.5 * theta, dmin(2.0 * theta, dmin(q, 1.6)))));
  double eta = r * dx;
  eta = (dvm * dvm + dvp * dvp) / (eta * eta);
  if (eta <= 1.0 - eps) {
    return q;
  } else if (eta >= 1.0 + eps) {
    return phi;
  }
  return 0.5 * ((1.0 - (eta - 1.0) / eps) * q + (1.0 + (eta - 1.0) / eps) * phi);
}

/* src/recon/limo3_simple.hpp:65-78 */
void orc_limo3(double qm1, double q0, double qp1, double dx, int ensure_positivity,
               double *ql_ip1, double *qr_i) {
  const double dqp = qp1 - q0;
  const double dqm = q0 - qm1;
  double ql = q0 + 0.5 * dqp * limo3_limiter(dqp, dqm, dx);
  double qr = q0 - 0.5 * dqm * limo3_limiter(dqm, dqp, dx);
  if (ensure_positivity && (ql <= 0.0 || qr <= 0.0)) {
    const double dmm = minmod(dqp, dqm);
    ql = q0 + 0.5 * dmm;
    qr = q0 - 0.5 * dmm;
  }
  *ql_ip1 = ql;
  *qr_i = qr;
}

/* Dispatcher mirroring the Reconstruct<recon,DIR> wrappers
 * (dc_simple.hpp:25-47, plm_simple.hpp:48-70, ppm_simple.hpp:173-198,
 *  wenoz_simple.hpp:92-117, weno3_simple.hpp:74-101, limo3_simple.hpp:89-118).
 * q[2] is the cell itself. */
void orc_recon_point(int recon, const double q[5], double dx, int n, double *ql_ip1,
                     double *qr_i) {
  switch (recon) {
  case ORC_RC_DC:
    *ql_ip1 = q[2];
    *qr_i = q[2];
    break;
  case ORC_RC_PLM:
    orc_plm(q[1], q[2], q[3], ql_ip1, qr_i);
    break;
  case ORC_RC_PPM:
    orc_ppm(q[0], q[1], q[2], q[3], q[4], ql_ip1, qr_i);
    break;
  case ORC_RC_WENOZ:
    orc_wenoz(q[0], q[1], q[2], q[3], q[4], ql_ip1, qr_i);
    break;
  case ORC_RC_WENO3: {
    double dx2 = dx;
    dx2 = dx2 * dx2;
    orc_weno3(q[1], q[2], q[3], dx2, ql_ip1, qr_i);
    break;
  }
  case ORC_RC_LIMO3:
    /* positivity fallback only for density and pressure: limo3_simple.hpp:98 */
    orc_limo3(q[1], q[2], q[3], dx, (n == ORC_IDN || n == ORC_IPR), ql_ip1, qr_i);
    break;
  default:
    *ql_ip1 = NAN;
    *qr_i = NAN;
  }
}

void orc_recon_many(int recon, long m, const double *q, double dx, int n, double *ql_ip1,
                    double *qr_i) {
  for (long s = 0; s < m; ++s) orc_recon_point(recon, q + 5 * s, dx, n, ql_ip1 + s, qr_i + s);
}


/* ---- branch tracing (apk_oracle.h) */
__thread unsigned *orc_trace_sink = 0;
void orc_recon_many_traced(int recon, long m, const double *q5, double dx, int n, double *ql, double *qr, unsigned *masks) {
  for (long s = 0; s < m; ++s) {
    masks[s] = 0u;
    orc_trace_sink = masks + s;
    orc_recon_many(recon, 1, q5 + 5 * s, dx, n, ql + s, qr + s);
  }
  orc_trace_sink = 0;
}
