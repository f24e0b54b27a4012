// sim_pgen.cpp -- problem generators of the standalone host driver (the reference's src/pgen/*.cpp for
// the problems on this path) and the host half of the turbulence driver's set-up.
#include "sim_internal.hpp"

using namespace apk;

namespace apk {
namespace host {

// ---- problem generators ---------------------------------------------------------------------
// cell centre: Xc = xmin + (global_index + 1/2) dx.  x0[d] carries the
// global index of the block's first interior cell, so centres do not depend on the decomposition.
double xc(const apk_sim *s, const double x0[3], int d, int idx) {
  const int ng = s->mesh.Active(d) ? s->mesh.ng : 0;
  return s->xmin[d] + ((x0[d] + (double)(idx - ng)) + 0.5) * s->dx[d];
}
void block_origin(const apk_sim *s, int lb, double x0[3]) {
  if (s->amr) {  // in cells of the block's own level (s->dx is set to that level's widths meanwhile)
    for (int d = 0; d < 3; ++d) x0[d] = (double)amr_leaf(s, lb).lx[d] * s->mesh.mb[d];
    return;
  }
  int bc[3];
  s->mesh.Loc(s->mesh.local_gids[lb], bc);
  for (int d = 0; d < 3; ++d) x0[d] = (double)(bc[d] * s->mesh.mb[d]);
}

// hydro eigensystem, src/pgen/linear_wave.cpp:421-500 (eigenvalues + right eigenvectors)
void lw_eigensystem(double gm1, double v1, double v2, double v3, double h, double ev[5], double rem[5][5]) {
  const double vsq = v1 * v1 + v2 * v2 + v3 * v3;
  const double asq = gm1 * std::max((h - 0.5 * vsq), 1.0e-20);
  const double a = std::sqrt(asq);
  ev[0] = v1 - a;
  ev[1] = ev[2] = ev[3] = v1;
  ev[4] = v1 + a;
  const double col[5][5] = {{1.0, v1 - a, v2, v3, h - v1 * a},
                            {0.0, 0.0, 1.0, 0.0, v2},
                            {0.0, 0.0, 0.0, 1.0, v3},
                            {1.0, v1, v2, v3, 0.5 * vsq},
                            {1.0, v1 + a, v2, v3, h + v1 * a}};
  for (int c = 0; c < 5; ++c)
    for (int r = 0; r < 5; ++r) rem[r][c] = col[c][r];
}

// src/pgen/linear_wave.cpp:72-176 (InitUserMeshData)
void lw_setup(apk_sim *s) {
  ParameterInput &pin = s->pin;
  LinearWaveState &lw = s->lw;
  lw.wave_flag = pin.GetInteger("problem/linear_wave", "wave_flag");
  const int nwave = (s->problem_id == "linear_wave_mhd") ? 7 : 5;
  if (lw.wave_flag < 0 || lw.wave_flag >= nwave)
    throw std::runtime_error("problem/linear_wave/wave_flag must be 0.." + std::to_string(nwave - 1));
  lw.amp = pin.GetReal("problem/linear_wave", "amp");
  lw.vflow = pin.GetOrAddReal("problem/linear_wave", "vflow", 0.0);
  double ang_2 = pin.GetOrAddReal("problem/linear_wave", "ang_2", -999.9);
  double ang_3 = pin.GetOrAddReal("problem/linear_wave", "ang_3", -999.9);
  const bool ang_2_vert = pin.GetOrAddBoolean("problem/linear_wave", "ang_2_vert", false);
  const bool ang_3_vert = pin.GetOrAddBoolean("problem/linear_wave", "ang_3_vert", false);
  lw.compute_error = pin.GetOrAddBoolean("problem/linear_wave", "compute_error", false);
  lw.gam = s->pkg.eos.gamma;
  lw.gm1 = lw.gam - 1.0;
  const double x1size = s->xmax[0] - s->xmin[0], x2size = s->xmax[1] - s->xmin[1],
               x3size = s->xmax[2] - s->xmin[2];
  if (ang_3 == -999.9) ang_3 = std::atan(x1size / x2size);
  lw.sin_a3 = std::sin(ang_3);
  lw.cos_a3 = std::cos(ang_3);
  if (ang_3_vert) {
    lw.sin_a3 = 1.0;
    lw.cos_a3 = 0.0;
    ang_3 = 0.5 * M_PI;
  }
  if (ang_2 == -999.9) ang_2 = std::atan(0.5 * (x1size * lw.cos_a3 + x2size * lw.sin_a3) / x3size);
  lw.sin_a2 = std::sin(ang_2);
  lw.cos_a2 = std::cos(ang_2);
  if (ang_2_vert) {
    lw.sin_a2 = 1.0;
    lw.cos_a2 = 0.0;
    ang_2 = 0.5 * M_PI;
  }
  const double x1 = x1size * lw.cos_a2 * lw.cos_a3;
  const double x2 = x2size * lw.cos_a2 * lw.sin_a3;
  const double x3 = x3size * lw.sin_a2;
  const int f2 = (s->mesh.nx[1] > 1) ? 1 : 0, f3 = (s->mesh.nx[2] > 1) ? 1 : 0;
  lw.lambda = x1;
  if (f2 && ang_3 != 0.0) lw.lambda = std::min(lw.lambda, x2);
  if (f3 && ang_2 != 0.0) lw.lambda = std::min(lw.lambda, x3);
  if (ang_3_vert) lw.lambda = x2;
  if (ang_2_vert) lw.lambda = x3;
  lw.k_par = 2.0 * (M_PI) / lw.lambda;
  lw.d0 = 1.0;
  lw.u0 = lw.vflow;
  lw.p0 = 1.0 / lw.gam;
  const double v0 = 0.0, w0 = 0.0;
  const double h0 = ((lw.p0 / lw.gm1 + 0.5 * lw.d0 * (lw.u0 * lw.u0 + v0 * v0 + w0 * w0)) + lw.p0) / lw.d0;
  lw_eigensystem(lw.gm1, lw.u0, v0, w0, h0, lw.ev, lw.rem);
  if (s->problem_id == "linear_wave" && pin.GetOrAddBoolean("problem/linear_wave", "test", false)) {
    // reinterpret tlim as the number of wave periods (linear_wave.cpp:169-175)
    s->tlim = lw.lambda / std::abs(lw.ev[lw.wave_flag]) * s->tlim;
  }
}

// analytic conserved state at a cell centre (linear_wave.cpp:355-373 == :206-226)
void lw_state(const LinearWaveState &lw, double x1, double x2, double x3, double u[5]) {
  const double x = lw.cos_a2 * (x1 * lw.cos_a3 + x2 * lw.sin_a3) + x3 * lw.sin_a2;
  const double sn = std::sin(lw.k_par * x);
  const int wf = lw.wave_flag;
  u[0] = lw.d0 + lw.amp * sn * lw.rem[0][wf];
  const double mx = lw.d0 * lw.vflow + lw.amp * sn * lw.rem[1][wf];
  const double my = lw.amp * sn * lw.rem[2][wf];
  const double mz = lw.amp * sn * lw.rem[3][wf];
  u[1] = mx * lw.cos_a2 * lw.cos_a3 - my * lw.sin_a3 - mz * lw.sin_a2 * lw.cos_a3;
  u[2] = mx * lw.cos_a2 * lw.sin_a3 + my * lw.cos_a3 - mz * lw.sin_a2 * lw.sin_a3;
  u[3] = mx * lw.sin_a2 + mz * lw.cos_a2;
  u[4] = lw.p0 / lw.gm1 + 0.5 * lw.d0 * lw.u0 * lw.u0 + lw.amp * sn * lw.rem[4][wf];
}

// ---- MHD linear wave (src/pgen/linear_wave_mhd.cpp) ---------------------------------------------
// Adiabatic MHD eigensystem in the conserved variables (d, mx, my, mz, E, by, bz): eigenvalues and RIGHT
// eigenvectors (linear_wave_mhd.cpp:486-625; the problem generator never reads the left ones).  One row of
// `fam` per wave family -- fast-, Alfven-, slow-, entropy, slow+, Alfven+, fast+ -- is one column of rem.
void lwm_eigensystem(double gm1, double d, double v1, double v2, double v3, double h, double b1, double b2, double b3,
                     double x, double y, double ev[7], double rem[7][7]) {
  const double vsq = v1 * v1 + v2 * v2 + v3 * v3;
  const double btsq = b2 * b2 + b3 * b3;
  const double bt_starsq = (gm1 - (gm1 - 1.0) * y) * btsq;
  const double vaxsq = b1 * b1 / d;
  const double hp = h - (vaxsq + btsq / d);
  const double twid_asq = std::max((gm1 * (hp - 0.5 * vsq) - (gm1 - 1.0) * x), 1.0e-20);
  const double ct2 = bt_starsq / d;
  const double tsum = vaxsq + ct2 + twid_asq;
  const double tdif = vaxsq + ct2 - twid_asq;
  const double cf2_cs2 = std::sqrt(tdif * tdif + 4.0 * twid_asq * ct2);
  const double cfsq = 0.5 * (tsum + cf2_cs2);
  const double cf = std::sqrt(cfsq);
  const double cssq = twid_asq * vaxsq / cfsq;
  const double cs = std::sqrt(cssq);
  const double bt = std::sqrt(btsq);
  const double bt_star = std::sqrt(bt_starsq);
  double bet2 = 1.0, bet3 = 0.0;
  if (bt != 0.0) {
    bet2 = b2 / bt;
    bet3 = b3 / bt;
  }
  const double bet2_star = bet2 / std::sqrt(gm1 - (gm1 - 1.0) * y);
  const double bet3_star = bet3 / std::sqrt(gm1 - (gm1 - 1.0) * y);
  const double bet_starsq = bet2_star * bet2_star + bet3_star * bet3_star;
  const double vbet = v2 * bet2_star + v3 * bet3_star;
  double alpha_f, alpha_s;
  if ((cfsq - cssq) == 0.0) {
    alpha_f = 1.0;
    alpha_s = 0.0;
  } else if ((twid_asq - cssq) <= 0.0) {
    alpha_f = 0.0;
    alpha_s = 1.0;
  } else if ((cfsq - twid_asq) <= 0.0) {
    alpha_f = 1.0;
    alpha_s = 0.0;
  } else {
    alpha_f = std::sqrt((twid_asq - cssq) / (cfsq - cssq));
    alpha_s = std::sqrt((cfsq - twid_asq) / (cfsq - cssq));
  }
  const double sqrtd = std::sqrt(d);
  const double isqrtd = 1.0 / sqrtd;
  const double sgn = (b1 < 0.0) ? -1.0 : 1.0;  // Parthenon SIGN
  const double twid_a = std::sqrt(twid_asq);
  const double qf = cf * alpha_f * sgn;
  const double qs = cs * alpha_s * sgn;
  const double af_prime = twid_a * alpha_f * isqrtd;
  const double as_prime = twid_a * alpha_s * isqrtd;
  const double afpbb = af_prime * bt_star * bet_starsq;
  const double aspbb = as_prime * bt_star * bet_starsq;
  const double vax = std::sqrt(vaxsq);
  const double lam[7] = {v1 - cf, v1 - vax, v1 - cs, v1, v1 + cs, v1 + vax, v1 + cf};
  for (int w = 0; w < 7; ++w) ev[w] = lam[w];
  const double af_v2 = alpha_f * v2, as_v2 = alpha_s * v2, qs_b2 = qs * bet2_star, qf_b2 = qf * bet2_star;
  const double af_v3 = alpha_f * v3, as_v3 = alpha_s * v3, qs_b3 = qs * bet3_star, qf_b3 = qf * bet3_star;
  const double e_alf = -(v2 * bet3 - v3 * bet2);
  const double fam[7][7] = {
      {alpha_f, alpha_f * lam[0], af_v2 + qs_b2, af_v3 + qs_b3, alpha_f * (hp - v1 * cf) + qs * vbet + aspbb,
       as_prime * bet2_star, as_prime * bet3_star},
      {0.0, 0.0, -bet3, bet2, e_alf, -bet3 * sgn * isqrtd, bet2 * sgn * isqrtd},
      {alpha_s, alpha_s * lam[2], as_v2 - qf_b2, as_v3 - qf_b3, alpha_s * (hp - v1 * cs) - qf * vbet - afpbb,
       -af_prime * bet2_star, -af_prime * bet3_star},
      {1.0, v1, v2, v3, 0.5 * vsq + (gm1 - 1.0) * x / gm1, 0.0, 0.0},
      {alpha_s, alpha_s * lam[4], as_v2 + qf_b2, as_v3 + qf_b3, alpha_s * (hp + v1 * cs) + qf * vbet - afpbb,
       -af_prime * bet2_star, -af_prime * bet3_star},
      {0.0, 0.0, bet3, -bet2, -e_alf, -bet3 * sgn * isqrtd, bet2 * sgn * isqrtd},
      {alpha_f, alpha_f * lam[6], af_v2 - qs_b2, af_v3 - qs_b3, alpha_f * (hp + v1 * cf) - qs * vbet + aspbb,
       as_prime * bet2_star, as_prime * bet3_star}};
  for (int w = 0; w < 7; ++w)
    for (int r = 0; r < 7; ++r) rem[r][w] = fam[w][r];
}

// InitUserMeshData (linear_wave_mhd.cpp:68-172): the hydro wave's geometry and background (the same
// expressions: :91-143 == linear_wave.cpp:86-140) plus B = (1, sqrt 2, 1/2) in the wave frame
void lwm_setup(apk_sim *s) {
  if (s->pkg.fluid != APK_FLUID_GLMMHD) throw std::runtime_error("linear_wave_mhd requires hydro/fluid = glmmhd");
  lw_setup(s);
  LinearWaveState &lw = s->lw;
  LinearWaveMhdState &m = s->lwm;
  m.bx0 = 1.0;
  m.by0 = std::sqrt(2.0);
  m.bz0 = 0.5;
  const double v0 = 0.0, w0 = 0.0;
  double h0 = ((lw.p0 / lw.gm1 + 0.5 * lw.d0 * (lw.u0 * lw.u0 + v0 * v0 + w0 * w0)) + lw.p0) / lw.d0;
  h0 += (m.bx0 * m.bx0 + m.by0 * m.by0 + m.bz0 * m.bz0) / lw.d0;
  lwm_eigensystem(lw.gm1, lw.d0, lw.u0, v0, w0, h0, m.bx0, m.by0, m.bz0, 0.0, 1.0, m.ev, m.rem);
  m.dby = lw.amp * m.rem[5][lw.wave_flag];  // (ProblemGenerator :392-393)
  m.dbz = lw.amp * m.rem[6][lw.wave_flag];
  if (s->pin.GetOrAddBoolean("problem/linear_wave", "test", false)) {
    // tlim counts wave periods (:165-171); the entropy wave of a fluid at rest has none
    s->tlim = lw.lambda / std::abs(m.ev[lw.wave_flag]) * s->tlim;
  }
}

// vector potential in the gauge Ax = 0 (linear_wave_mhd.cpp:443-479)
static void lwm_potential(const apk_sim *s, double x1, double x2, double x3, double A[3]) {
  const LinearWaveState &lw = s->lw;
  const LinearWaveMhdState &m = s->lwm;
  const double x = x1 * lw.cos_a2 * lw.cos_a3 + x2 * lw.cos_a2 * lw.sin_a3 + x3 * lw.sin_a2;
  const double y = -x1 * lw.sin_a3 + x2 * lw.cos_a3;
  const double Ay = m.bz0 * x - (m.dbz / lw.k_par) * std::cos(lw.k_par * (x));
  const double Az = -m.by0 * x + (m.dby / lw.k_par) * std::cos(lw.k_par * (x)) + m.bx0 * y;
  A[0] = -Ay * lw.sin_a3 - Az * lw.sin_a2 * lw.cos_a3;
  A[1] = Ay * lw.cos_a3 - Az * lw.sin_a2 * lw.sin_a3;
  A[2] = Az * lw.cos_a2;
}

// the analytic wave at a cell centre: d, M1, M2, M3, E (linear_wave_mhd.cpp:407-436) and the ANALYTIC field
// B1, B2, B3 of the error norm (:213-224); the problem generator differences the potential instead
void lwm_state(const apk_sim *s, double x1, double x2, double x3, double u[8]) {
  const LinearWaveState &lw = s->lw;
  const LinearWaveMhdState &m = s->lwm;
  const int wf = lw.wave_flag;
  const double x = lw.cos_a2 * (x1 * lw.cos_a3 + x2 * lw.sin_a3) + x3 * lw.sin_a2;
  const double sn = std::sin(lw.k_par * x);
  u[0] = lw.d0 + lw.amp * sn * m.rem[0][wf];
  const double mx = lw.d0 * lw.vflow + lw.amp * sn * m.rem[1][wf];
  const double my = lw.amp * sn * m.rem[2][wf];
  const double mz = lw.amp * sn * m.rem[3][wf];
  u[1] = mx * lw.cos_a2 * lw.cos_a3 - my * lw.sin_a3 - mz * lw.sin_a2 * lw.cos_a3;
  u[2] = mx * lw.cos_a2 * lw.sin_a3 + my * lw.cos_a3 - mz * lw.sin_a2 * lw.sin_a3;
  u[3] = mx * lw.sin_a2 + mz * lw.cos_a2;
  double e0 = lw.p0 / lw.gm1 + 0.5 * lw.d0 * lw.u0 * lw.u0 + lw.amp * sn * m.rem[4][wf];
  e0 += 0.5 * (m.bx0 * m.bx0 + m.by0 * m.by0 + m.bz0 * m.bz0);
  u[4] = e0;
  const double bx = m.bx0;
  const double by = m.by0 + lw.amp * sn * m.rem[5][wf];
  const double bz = m.bz0 + lw.amp * sn * m.rem[6][wf];
  u[5] = bx * lw.cos_a2 * lw.cos_a3 - by * lw.sin_a3 - bz * lw.sin_a2 * lw.cos_a3;
  u[6] = bx * lw.cos_a2 * lw.sin_a3 + by * lw.cos_a3 - bz * lw.sin_a2 * lw.sin_a3;
  u[7] = bx * lw.sin_a2 + bz * lw.cos_a2;
}

// fills the interior of one block's host image [nvar][Nk][Nj][Ni]
// ---- circularly polarised Alfven wave (src/pgen/cpaw.cpp) -----------------------------------------
// InitUserMeshData (cpaw.cpp:58-125)
void cpaw_setup(apk_sim *s) {
  ParameterInput &pin = s->pin;
  CpawState &c = s->cpaw;
  if (s->pkg.fluid != APK_FLUID_GLMMHD) throw std::runtime_error("cpaw requires hydro/fluid = glmmhd");
  if (s->mesh.ndim != 3) throw std::runtime_error("cpaw is set up for 3-D meshes here");
  c.b_par = pin.GetReal("problem/cpaw", "b_par");
  c.b_perp = pin.GetReal("problem/cpaw", "b_perp");
  c.v_par = pin.GetReal("problem/cpaw", "v_par");
  double ang_2 = pin.GetOrAddReal("problem/cpaw", "ang_2", -999.9);
  double ang_3 = pin.GetOrAddReal("problem/cpaw", "ang_3", -999.9);
  const double dir = pin.GetOrAddReal("problem/cpaw", "dir", 1);  // right (1) / left (2) polarisation
  c.gm1 = pin.GetReal("hydro", "gamma") - 1.0;
  c.pres = pin.GetReal("problem/cpaw", "pres");
  c.den = 1.0;
  c.compute_error = pin.GetOrAddBoolean("problem/cpaw", "compute_error", false);
  const double x1size = s->xmax[0] - s->xmin[0], x2size = s->xmax[1] - s->xmin[1], x3size = s->xmax[2] - s->xmin[2];
  if (ang_3 == -999.9) ang_3 = std::atan(x1size / x2size);
  c.sin_a3 = std::sin(ang_3);
  c.cos_a3 = std::cos(ang_3);
  if (ang_2 == -999.9) ang_2 = std::atan(0.5 * (x1size * c.cos_a3 + x2size * c.sin_a3) / x3size);
  c.sin_a2 = std::sin(ang_2);
  c.cos_a2 = std::cos(ang_2);
  const double x1 = x1size * c.cos_a2 * c.cos_a3, x2 = x2size * c.cos_a2 * c.sin_a3, x3 = x3size * c.sin_a2;
  c.lambda = x1;  // the smallest of the three
  if (s->mesh.nx[1] > 1 && ang_3 != 0.0) c.lambda = std::min(c.lambda, x2);
  if (s->mesh.nx[2] > 1 && ang_2 != 0.0) c.lambda = std::min(c.lambda, x3);
  c.k_par = 2.0 * (M_PI) / c.lambda;
  c.v_perp = c.b_perp / std::sqrt(c.den);
  c.fac = (dir == 1) ? 1.0 : -1.0;
}

// vector potential, gauge Ax = 0 (cpaw.cpp:310-344)
void cpaw_potential(const CpawState &c, double x1, double x2, double x3, double A[3]) {
  const double x = x1 * c.cos_a2 * c.cos_a3 + x2 * c.cos_a2 * c.sin_a3 + x3 * c.sin_a2;
  const double y = -x1 * c.sin_a3 + x2 * c.cos_a3;
  const double Ay = c.fac * (c.b_perp / c.k_par) * std::sin(c.k_par * (x));
  const double Az = (c.b_perp / c.k_par) * std::cos(c.k_par * (x)) + c.b_par * y;
  A[0] = -Ay * c.sin_a3 - Az * c.sin_a2 * c.cos_a3;
  A[1] = Ay * c.cos_a3 - Az * c.sin_a2 * c.sin_a3;
  A[2] = Az * c.cos_a2;
}

// analytic momenta / fields of the wave at one point (cpaw.cpp:147-175, 262-275)
void cpaw_state(const CpawState &c, double X1, double X2, double X3, double m[3], double b[3]) {
  const double x = c.cos_a2 * (X1 * c.cos_a3 + X2 * c.sin_a3) + X3 * c.sin_a2;
  const double sn = std::sin(c.k_par * x);
  const double cs = c.fac * std::cos(c.k_par * x);
  const double mx = c.den * c.v_par, my = -c.fac * c.den * c.v_perp * sn, mz = -c.fac * c.den * c.v_perp * cs;
  m[0] = mx * c.cos_a2 * c.cos_a3 - my * c.sin_a3 - mz * c.sin_a2 * c.cos_a3;
  m[1] = mx * c.cos_a2 * c.sin_a3 + my * c.cos_a3 - mz * c.sin_a2 * c.sin_a3;
  m[2] = mx * c.sin_a2 + mz * c.cos_a2;
  const double bx = c.b_par, by = c.b_perp * sn, bz = c.b_perp * cs;
  b[0] = bx * c.cos_a2 * c.cos_a3 - by * c.sin_a3 - bz * c.sin_a2 * c.cos_a3;
  b[1] = bx * c.cos_a2 * c.sin_a3 + by * c.cos_a3 - bz * c.sin_a2 * c.sin_a3;
  b[2] = bx * c.sin_a2 + bz * c.cos_a2;
}

// ---- advected field loop (src/pgen/field_loop.cpp) ------------------------------------------------
void field_loop_setup(apk_sim *s) {  // parameter block :129-170
  ParameterInput &pin = s->pin;
  FieldLoopState &f = s->floop;
  if (s->pkg.fluid != APK_FLUID_GLMMHD) throw std::runtime_error("field_loop requires hydro/fluid = glmmhd");
  if (s->mesh.ndim < 2) throw std::runtime_error("field_loop needs a 2-D or 3-D mesh");
  f.rad = pin.GetReal("problem/field_loop", "rad");
  f.amp = pin.GetReal("problem/field_loop", "amp");
  f.vflow = pin.GetReal("problem/field_loop", "vflow");
  f.drat = pin.GetOrAddReal("problem/field_loop", "drat", 1.0);
  f.iprob = pin.GetInteger("problem/field_loop", "iprob");
  if (f.iprob == 4) {  // rotated cylinder: one wavelength along each of x1 and x3
    const double L1 = s->xmax[0] - s->xmin[0], L3 = s->mesh.ndim < 3 ? 0.0 : s->xmax[2] - s->xmin[2];
    if (L1 == L3) {
      f.cos_a2 = f.sin_a2 = std::sqrt(0.5);
    } else {
      const double ang_2 = std::atan(L1 / L3);
      f.sin_a2 = std::sin(ang_2);
      f.cos_a2 = std::cos(ang_2);
    }
    f.lambda = f.cos_a2 >= f.sin_a2 ? L1 * f.cos_a2 : L3 * f.sin_a2;
  }
}

// cell-centred vector potential of the loop, one branch per iprob (:196-290)
void field_loop_potential(const FieldLoopState &f, double x1, double x2, double x3, double A[3]) {
  A[0] = A[1] = A[2] = 0.0;
  auto cone = [&](double rsq) { return rsq < f.rad * f.rad ? f.amp * (f.rad - std::sqrt(rsq)) : 0.0; };
  switch (f.iprob) {
  case 1: A[2] = cone(x1 * x1 + x2 * x2); break;
  case 2: A[0] = cone(x2 * x2 + x3 * x3); break;
  case 3: A[1] = cone(x1 * x1 + x3 * x3); break;
  case 4: {
    double x = x1 * f.cos_a2 + x3 * f.sin_a2;
    while (x > 0.5 * f.lambda) x -= f.lambda;
    while (x < -0.5 * f.lambda) x += f.lambda;
    if ((x * x + x2 * x2) < f.rad * f.rad) {  // keeps +0 outside the loop
      A[0] = cone(x * x + x2 * x2) * (-f.sin_a2);
      A[2] = cone(x * x + x2 * x2) * (f.cos_a2);
    }
    break;
  }
  case 5: A[1] = A[2] = cone(x1 * x1 + x2 * x2 + x3 * x3); break;
  default: break;
  }
}

// Kelvin-Helmholtz (src/pgen/kh.cpp): validate the deck when the sim is created
void kh_setup(apk_sim *s) {
  ParameterInput &pin = s->pin;
  if (s->pkg.fluid == APK_FLUID_GLMMHD) throw std::runtime_error("the kh problem generator is hydro only");
  if (s->mesh.ndim < 2) throw std::runtime_error("kh needs a 2-D or 3-D mesh");
  (void)pin.GetReal("problem/kh", "vflow");
  const int iprob = pin.GetInteger("problem/kh", "iprob");
  if (iprob < 2 || iprob > 5) throw std::runtime_error("Unknow iprob for KHI pgen.");
  (void)pin.GetReal("problem/kh", "amp");
  if (iprob == 5) {
    (void)pin.GetReal("problem/kh", "a");
    (void)pin.GetReal("problem/kh", "sigma");
    (void)pin.GetReal("problem/kh", "drat");
  }
}

void pgen_block(apk_sim *s, int lb, std::vector<double> &u) {
  LevelDxScope level_dx_scope(s, lb);
  const Mesh &m = s->mesh;
  const HydroPackage &pkg = s->pkg;
  std::fill(u.begin(), u.end(), 0.0);
  double x0[3];
  block_origin(s, lb, x0);
  auto at = [&](int n, int k, int j, int i) -> double & { return u[n * m.sn + k * m.sk + j * m.sj + i]; };
  const bool mhd = pkg.fluid == APK_FLUID_GLMMHD;
  const double gm1 = pkg.eos.gamma - 1.0;
  ParameterInput &pin = s->pin;
  double sod[7] = {0};
  if (s->problem_id == "sod") {  // src/pgen/sod.cpp:24-30
    sod[0] = pin.GetOrAddReal("problem/sod", "rho_l", 1.0);
    sod[1] = pin.GetOrAddReal("problem/sod", "pres_l", 1.0);
    sod[2] = pin.GetOrAddReal("problem/sod", "u_l", 0.0);
    sod[3] = pin.GetOrAddReal("problem/sod", "rho_r", 0.125);
    sod[4] = pin.GetOrAddReal("problem/sod", "pres_r", 0.1);
    sod[5] = pin.GetOrAddReal("problem/sod", "u_r", 0.0);
    sod[6] = pin.GetOrAddReal("problem/sod", "x_discont", 0.5);
  }
  double bl[9] = {0};
  if (s->problem_id == "blast") {  // src/pgen/blast.cpp:125-138
    if (pin.GetOrAddString("problem/blast", "input_image", "none") != "none")
      throw std::runtime_error("problem/blast/input_image is not supported");
    bl[0] = pin.GetReal("problem/blast", "radius_outer");
    bl[1] = pin.GetOrAddReal("problem/blast", "radius_inner", bl[0]);
    bl[2] = pin.GetOrAddReal("problem/blast", "pressure_ambient", 1.0);
    bl[3] = pin.GetOrAddReal("problem/blast", "density_ambient", 1.0);
    bl[4] = pin.GetReal("problem/blast", "pressure_ratio");
    bl[5] = pin.GetOrAddReal("problem/blast", "density_ratio", 1.0);
    bl[6] = pin.GetOrAddReal("problem/blast", "x1_0", 0.0);
    bl[7] = pin.GetOrAddReal("problem/blast", "x2_0", 0.0);
    bl[8] = pin.GetOrAddReal("problem/blast", "x3_0", 0.0);
  }
  double adv[9] = {0};
  if (s->problem_id == "advection") {  // src/pgen/advection.cpp:68-79
    adv[0] = pin.GetOrAddReal("problem/advection", "vx", 0.0);
    adv[1] = pin.GetOrAddReal("problem/advection", "vy", 0.0);
    adv[2] = pin.GetOrAddReal("problem/advection", "vz", 0.0);
    adv[3] = pin.GetOrAddReal("problem/advection", "rho_ratio", 1.0);
    adv[4] = pin.GetOrAddReal("problem/advection", "rho_radius", 0.0);
    adv[5] = pin.GetOrAddReal("problem/advection", "rho_fraction_edge", 0.01);
    adv[6] = pin.GetOrAddReal("problem/advection", "rho0", 1.0);
    adv[7] = pin.GetOrAddReal("problem/advection", "p0", 1.0);
    adv[8] = -adv[4] * adv[4] / 2 / std::log(adv[5]);  // sigmasq
  }
  double kh[8] = {0};
  int kh_iprob = 0;
  if (s->problem_id == "kh") {  // src/pgen/kh.cpp:44-45 and the per-iprob parameter reads
    if (mhd) throw std::runtime_error("the kh problem generator is hydro only");
    if (m.ndim < 2) throw std::runtime_error("kh needs a 2-D or 3-D mesh");
    kh[0] = pin.GetReal("problem/kh", "vflow");
    kh_iprob = pin.GetInteger("problem/kh", "iprob");
    if (kh_iprob < 2 || kh_iprob > 5) throw std::runtime_error("Unknow iprob for KHI pgen.");
    kh[1] = pin.GetReal("problem/kh", "amp");
    if (kh_iprob == 4) {
      kh[2] = pin.GetOrAddReal("problem/kh", "drho_rho0", 0.0);
      kh[3] = pin.GetOrAddReal("problem/kh", "vboost", 0.0);
    } else if (kh_iprob == 5) {
      kh[4] = pin.GetReal("problem/kh", "a");
      kh[5] = pin.GetReal("problem/kh", "sigma");
      kh[6] = pin.GetReal("problem/kh", "drat");
    }
  }
  double lwi[5] = {0};
  if (s->problem_id == "lw_implode") {  // src/pgen/lw_implode.cpp:24-57
    if (mhd) throw std::runtime_error("Only hydro runs are supported for LW implosion problem generator.");
    lwi[0] = pin.GetReal("problem/lw_implode", "d_in");
    lwi[1] = pin.GetReal("problem/lw_implode", "p_in");
    lwi[2] = pin.GetReal("problem/lw_implode", "d_out");
    lwi[3] = pin.GetReal("problem/lw_implode", "p_out");
    // to keep the ICs symmetric y0 sits between cell centres; the reference adjusts it with a loop
    // over the rows of the meshblock being initialised
    double y0 = 0.5 * (s->xmax[1] + s->xmin[1]);
    for (int j = m.js; j <= m.je; ++j)
      if (xc(s, x0, 1, j) > y0) {
        const int ngj = m.Active(1) ? m.ng : 0;
        const double xf = s->xmin[1] + (x0[1] + (double)(j - ngj)) * s->dx[1];  // lower x2 face of cell j
        y0 = xf + 0.5 * s->dx[1];
        break;
      }
    lwi[4] = y0;
  }
  for (int k = m.ks; k <= m.ke; ++k)
    for (int j = m.js; j <= m.je; ++j)
      for (int i = m.is; i <= m.ie; ++i) {
        const double x1 = xc(s, x0, 0, i), x2 = xc(s, x0, 1, j), x3 = xc(s, x0, 2, k);
        if (s->problem_id == "advection") {  // src/pgen/advection.cpp:91-110
          double rho = adv[6];
          const double rsq = x1 * x1 + x2 * x2 + x3 * x3;
          if (rsq < adv[4] * adv[4]) rho += adv[6] * adv[3] * std::exp(-rsq / 2 / adv[8]);
          const double mx = rho * adv[0], my = rho * adv[1], mz = rho * adv[2];
          at(0, k, j, i) = rho;
          at(1, k, j, i) = mx;
          at(2, k, j, i) = my;
          at(3, k, j, i) = mz;
          at(4, k, j, i) = adv[7] / gm1 + 0.5 * (mx * mx + my * my + mz * mz) / rho;
        } else if (s->problem_id == "cpaw") {  // src/pgen/cpaw.cpp:255-300
          const CpawState &c = s->cpaw;
          double mom[3], bana[3];
          cpaw_state(c, x1, x2, x3, mom, bana);
          at(0, k, j, i) = c.den;
          at(1, k, j, i) = mom[0];
          at(2, k, j, i) = mom[1];
          at(3, k, j, i) = mom[2];
          // B = curl A by centred differences of the cell-centred potential
          double Ajp[3], Ajm[3], Akp[3], Akm[3], Aip[3], Aim[3];
          cpaw_potential(c, x1, xc(s, x0, 1, j + 1), x3, Ajp);
          cpaw_potential(c, x1, xc(s, x0, 1, j - 1), x3, Ajm);
          cpaw_potential(c, x1, x2, xc(s, x0, 2, k + 1), Akp);
          cpaw_potential(c, x1, x2, xc(s, x0, 2, k - 1), Akm);
          cpaw_potential(c, xc(s, x0, 0, i + 1), x2, x3, Aip);
          cpaw_potential(c, xc(s, x0, 0, i - 1), x2, x3, Aim);
          const double b1 = (Ajp[2] - Ajm[2]) / s->dx[1] / 2.0 - (Akp[1] - Akm[1]) / s->dx[2] / 2.0;
          const double b2 = (Akp[0] - Akm[0]) / s->dx[2] / 2.0 - (Aip[2] - Aim[2]) / s->dx[0] / 2.0;
          const double b3 = (Aip[1] - Aim[1]) / s->dx[0] / 2.0 - (Ajp[0] - Ajm[0]) / s->dx[1] / 2.0;
          at(5, k, j, i) = b1;
          at(6, k, j, i) = b2;
          at(7, k, j, i) = b3;
          at(4, k, j, i) = c.pres / c.gm1 + 0.5 * (b1 * b1 + b2 * b2 + b3 * b3) +
                           (0.5 / c.den) * (mom[0] * mom[0] + mom[1] * mom[1] + mom[2] * mom[2]);
        } else if (s->problem_id == "kh") {  // src/pgen/kh.cpp:61-234
          const double vflow = kh[0], amp = kh[1];
          double d = 1.0, m1 = 0.0, m2 = 0.0, pr = 1.0;
          if (kh_iprob == 2) {  // one tanh shear layer (Frank et al. 1996)
            const double a = 0.02, sigma = 0.2;
            m1 = vflow * std::tanh(x2 / a);
            m2 = amp * std::cos(2.0 * M_PI * x1) * std::exp(-(x2 * x2) / (sigma * sigma));
          } else if (kh_iprob == 3) {  // two resolved layers at |y| = 0.5 (Beckwith & Stone 2011)
            const double a = 0.01, sigma = 0.1, s2 = std::abs(x2) - 0.5;
            d = 0.505 + 0.495 * std::tanh(s2 / a);
            m1 = vflow * std::tanh(s2 / a);
            m2 = amp * vflow * std::sin(2.0 * M_PI * x1) * std::exp(-(s2 * s2) / (sigma * sigma));
            if (x2 < 0.0) m2 *= -1.0;
            m1 *= d;
            m2 *= d;
          } else if (kh_iprob == 4) {  // Lecoanet et al. 2016, domain centred on the origin
            const double a = 0.05, sigma = 0.2, z1 = -0.5, z2 = 0.5;
            const double t1 = std::tanh((x2 - z1) / a), t2 = std::tanh((x2 - z2) / a);
            pr = 10.0;
            d = 1.0 + 0.5 * kh[2] * (t1 - t2);
            m1 = (vflow * (t1 - t2 - 1.0) + kh[3]) * d;
            // the sine averaged with minus its half-period shift, for shift symmetry in floating point
            double ave_sine = std::sin(2.0 * M_PI * x1);
            ave_sine -= std::sin(2.0 * M_PI * ((x1 > 0.0 ? -0.5 : 0.5) + x1));
            ave_sine /= 2.0;
            const double v2 = -amp * ave_sine *
                              (std::exp(-((x2 - z1) * (x2 - z1)) / (sigma * sigma)) + std::exp(-((x2 - z2) * (x2 - z2)) / (sigma * sigma)));
            m2 = v2 * d;
          } else {  // iprob 5: dense stream in |y| < 1/4, m = 2 perturbation (the AMR test)
            const double s2 = std::abs(x2) - 0.25;
            const double w = (std::tanh(s2 / kh[4]) + 1.0) * 0.5;
            pr = 2.5;
            d = w + (1.0 - w) * kh[6];
            m1 = d * vflow * (w - 0.5);
            m2 = d * amp * std::cos(2.0 * 2.0 * M_PI * x1) * std::exp(-(s2 * s2) / (kh[5] * kh[5]));
          }
          at(0, k, j, i) = d;
          at(1, k, j, i) = m1;
          at(2, k, j, i) = m2;
          at(3, k, j, i) = 0.0;
          at(4, k, j, i) = pr / gm1 + 0.5 * (m1 * m1 + m2 * m2) / d;
        } else if (s->problem_id == "field_loop") {  // src/pgen/field_loop.cpp:292-322
          const FieldLoopState &f = s->floop;
          const bool two_d = m.ndim < 3;
          const double L[3] = {s->xmax[0] - s->xmin[0], s->xmax[1] - s->xmin[1], two_d ? 0.0 : s->xmax[2] - s->xmin[2]};
          const double den = (x1 * x1 + x2 * x2 + x3 * x3) < f.rad * f.rad ? f.drat : 1.0;
          double Ajp[3], Ajm[3], Aip[3], Aim[3], Akp[3] = {0, 0, 0}, Akm[3] = {0, 0, 0};
          field_loop_potential(f, x1, xc(s, x0, 1, j + 1), x3, Ajp);
          field_loop_potential(f, x1, xc(s, x0, 1, j - 1), x3, Ajm);
          field_loop_potential(f, xc(s, x0, 0, i + 1), x2, x3, Aip);
          field_loop_potential(f, xc(s, x0, 0, i - 1), x2, x3, Aim);
          if (!two_d) {
            field_loop_potential(f, x1, x2, xc(s, x0, 2, k + 1), Akp);
            field_loop_potential(f, x1, x2, xc(s, x0, 2, k - 1), Akm);
          }
          const double aydz = two_d ? 0.0 : (Akp[1] - Akm[1]) / s->dx[2] / 2.0;
          const double axdz = two_d ? 0.0 : (Akp[0] - Akm[0]) / s->dx[2] / 2.0;
          const double b1 = (Ajp[2] - Ajm[2]) / s->dx[1] / 2.0 - aydz;
          const double b2 = axdz - (Aip[2] - Aim[2]) / s->dx[0] / 2.0;
          const double b3 = (Aip[1] - Aim[1]) / s->dx[0] / 2.0 - (Ajp[0] - Ajm[0]) / s->dx[1] / 2.0;
          const double m1 = den * f.vflow * L[0], m2 = den * f.vflow * L[1], m3 = den * f.vflow * L[2];
          at(0, k, j, i) = den;
          at(1, k, j, i) = m1;
          at(2, k, j, i) = m2;
          at(3, k, j, i) = m3;
          at(5, k, j, i) = b1;
          at(6, k, j, i) = b2;
          at(7, k, j, i) = b3;
          at(4, k, j, i) = 1.0 / gm1 + 0.5 * (b1 * b1 + b2 * b2 + b3 * b3) + 0.5 * (m1 * m1 + m2 * m2 + m3 * m3) / den;
        } else if (s->problem_id == "lw_implode") {  // src/pgen/lw_implode.cpp:59-73
          const bool outside = x2 > (lwi[4] - x1);
          at(0, k, j, i) = outside ? lwi[2] : lwi[0];
          at(4, k, j, i) = (outside ? lwi[3] : lwi[1]) / gm1;
        } else if (s->problem_id == "blast") {  // src/pgen/blast.cpp:150-201
          const double rout = bl[0], rin = bl[1], pa = bl[2], da = bl[3], prat = bl[4], drat = bl[5];
          double den = da, pres = pa;
          const double rad = std::sqrt((x1 - bl[6]) * (x1 - bl[6]) + (x2 - bl[7]) * (x2 - bl[7]) + (x3 - bl[8]) * (x3 - bl[8]));
          if (rad < rout) {
            if (rad < rin) {
              den = drat * da;
            } else {  // smooth ramp in density
              const double f = (rad - rin) / (rout - rin);
              const double log_den = (1.0 - f) * std::log(drat * da) + f * std::log(da);
              den = std::exp(log_den);
            }
          }
          if (rad < rout) {
            if (rad < rin) {
              pres = prat * pa;
            } else {  // smooth ramp in pressure
              const double f = (rad - rin) / (rout - rin);
              const double log_pres = (1.0 - f) * std::log(prat * pa) + f * std::log(pa);
              pres = std::exp(log_pres);
            }
          }
          at(0, k, j, i) = den;
          at(4, k, j, i) = pres / gm1;
        } else if (s->problem_id == "linear_wave") {
          double w[5];
          lw_state(s->lw, x1, x2, x3, w);
          for (int n = 0; n < 5; ++n) at(n, k, j, i) = w[n];
        } else if (s->problem_id == "linear_wave_mhd") {  // src/pgen/linear_wave_mhd.cpp:395-437
          double w[8];
          lwm_state(s, x1, x2, x3, w);
          for (int n = 0; n < 5; ++n) at(n, k, j, i) = w[n];
          // B = curl A by centred differences of the cell-centred potential
          double Ajp[3], Ajm[3], Akp[3], Akm[3], Aip[3], Aim[3];
          lwm_potential(s, x1, xc(s, x0, 1, j + 1), x3, Ajp);
          lwm_potential(s, x1, xc(s, x0, 1, j - 1), x3, Ajm);
          lwm_potential(s, x1, x2, xc(s, x0, 2, k + 1), Akp);
          lwm_potential(s, x1, x2, xc(s, x0, 2, k - 1), Akm);
          lwm_potential(s, xc(s, x0, 0, i + 1), x2, x3, Aip);
          lwm_potential(s, xc(s, x0, 0, i - 1), x2, x3, Aim);
          at(5, k, j, i) = (Ajp[2] - Ajm[2]) / s->dx[1] / 2.0 - (Akp[1] - Akm[1]) / s->dx[2] / 2.0;
          at(6, k, j, i) = (Akp[0] - Akm[0]) / s->dx[2] / 2.0 - (Aip[2] - Aim[2]) / s->dx[0] / 2.0;
          at(7, k, j, i) = (Aip[1] - Aim[1]) / s->dx[0] / 2.0 - (Ajp[0] - Ajm[0]) / s->dx[1] / 2.0;
        } else if (s->problem_id == "sod") {  // src/pgen/sod.cpp:37-50
          const bool left = x1 < sod[6];
          const double rho = left ? sod[0] : sod[3], pr = left ? sod[1] : sod[4], ux = left ? sod[2] : sod[5];
          at(0, k, j, i) = rho;
          at(1, k, j, i) = rho * ux;
          at(4, k, j, i) = 0.5 * rho * ux * ux + pr / (pkg.eos.gamma - 1.0);
        } else if (s->problem_id == "orszag_tang") {  // src/pgen/orszag_tang.cpp:33-61
          if (!mhd) throw std::runtime_error("orszag_tang requires hydro/fluid = glmmhd");
          const double B0 = 1.0 / std::sqrt(4.0 * M_PI), d0 = 25.0 / (36.0 * M_PI), v0 = 1.0,
                       p0 = 5.0 / (12.0 * M_PI);
          at(0, k, j, i) = d0;
          at(1, k, j, i) = d0 * v0 * std::sin(2.0 * M_PI * x2);
          at(2, k, j, i) = -d0 * v0 * std::sin(2.0 * M_PI * x1);
          at(3, k, j, i) = 0.0;
          at(5, k, j, i) = B0 * std::sin(2.0 * M_PI * x2);
          at(6, k, j, i) = B0 * std::sin(4.0 * M_PI * x1);
          at(7, k, j, i) = 0.0;
          const double b1 = at(5, k, j, i), b2 = at(6, k, j, i), b3 = at(7, k, j, i);
          const double m1 = at(1, k, j, i), m2 = at(2, k, j, i), m3 = at(3, k, j, i);
          at(4, k, j, i) = p0 / gm1 + 0.5 * (b1 * b1 + b2 * b2 + b3 * b3 + (m1 * m1 + m2 * m2 + m3 * m3) / at(0, k, j, i));
        } else if (s->problem_id == "synthetic") {
          // analytic, seedless smooth state (SURVEY.md 8(d) synthetic kernel benchmark)
          const double tp = 2.0 * M_PI;
          const double fx = (x1 - s->xmin[0]) / (s->xmax[0] - s->xmin[0]);
          const double fy = (x2 - s->xmin[1]) / (s->xmax[1] - s->xmin[1]);
          const double fz = (x3 - s->xmin[2]) / (s->xmax[2] - s->xmin[2]);
          const double rho = 1.0 + 0.2 * std::sin(tp * (fx + fy + fz));
          const double p = 1.0 + 0.1 * std::cos(tp * (fx - fy + 2.0 * fz));
          const double v1 = 0.17 * std::sin(tp * (fy + fz));
          const double v2 = 0.17 * std::cos(tp * (fx - fz));
          const double v3 = 0.17 * std::sin(tp * (2.0 * fx + fy));
          double b1 = 0, b2 = 0, b3 = 0, psi = 0;
          if (mhd) {
            b1 = 0.28 * std::cos(tp * (fy - fz));
            b2 = 0.28 * std::sin(tp * (fx + 2.0 * fz));
            b3 = 0.28 * std::cos(tp * (fx + fy));
            psi = 0.01 * std::sin(tp * (fx + fy - fz));
          }
          at(0, k, j, i) = rho;
          at(1, k, j, i) = rho * v1;
          at(2, k, j, i) = rho * v2;
          at(3, k, j, i) = rho * v3;
          double e = p / gm1 + 0.5 * rho * (v1 * v1 + v2 * v2 + v3 * v3);
          if (mhd) {
            e += 0.5 * (b1 * b1 + b2 * b2 + b3 * b3);
            at(5, k, j, i) = b1;
            at(6, k, j, i) = b2;
            at(7, k, j, i) = b3;
            at(8, k, j, i) = psi;
          }
          at(4, k, j, i) = e;
          for (int n = pkg.nhydro; n < m.nvar; ++n)
            at(n, k, j, i) = rho * (0.5 + 0.25 * std::sin(tp * (fx + (n - pkg.nhydro + 1) * fy)));
        } else {
          throw std::runtime_error("unknown job/problem_id: " + s->problem_id);
        }
      }
}


// ---- few-modes turbulence driver -----------------------------------------------------------
// turbulence::ProblemInitPackageData (src/pgen/turbulence.cpp:103-200) + the checks of
// FewModesFT::SetPhases (src/utils/few_modes_ft.cpp:113-140)
void turbulence_setup(apk_sim *s) {
  ParameterInput &pin = s->pin;
  const int num_modes = pin.GetInteger("problem/turbulence", "num_modes");
  const uint32_t rseed = static_cast<uint32_t>(pin.GetOrAddInteger("problem/turbulence", "rseed", -1));
  const double k_peak = pin.GetOrAddReal("problem/turbulence", "kpeak", 0.0);
  s->accel_rms = pin.GetReal("problem/turbulence", "accel_rms");
  const double t_corr = pin.GetReal("problem/turbulence", "corr_time");
  const double sol_weight = pin.GetReal("problem/turbulence", "sol_weight");
  if (num_modes <= 0) throw std::runtime_error("problem/turbulence/num_modes must be positive");
  std::vector<double> k_vec(3 * (size_t)num_modes);
  for (int d = 0; d < 3; ++d)
    for (int m = 1; m <= num_modes; ++m)
      k_vec[(size_t)d * num_modes + (m - 1)] = pin.GetInteger("modes", "k_" + std::to_string(m) + "_" + std::to_string(d));
  if (pin.GetOrAddInteger("parthenon/mesh", "pack_size", -1) != -1)
    throw std::runtime_error("Few modes FT currently needs parthenon/mesh/pack_size=-1 to work because of global reductions.");
  const Mesh &m = s->mesh;
  const double L[3] = {s->xmax[0] - s->xmin[0], s->xmax[1] - s->xmin[1], s->xmax[2] - s->xmin[2]};
  if (!(m.nx[0] == m.nx[1] && m.nx[1] == m.nx[2] && L[0] == L[1] && L[1] == L[2]))
    throw std::runtime_error("FMFT has only been tested with cubic meshes and constant dx/dy/dz. "
                             "Remove this warning at your own risk.");
  if (pin.DoesParameterExist("problem/turbulence", "accel_hat_0_0_r"))
    throw std::runtime_error("restarting the turbulence driver state is not supported");
  if (s->pkg.fluid == APK_FLUID_GLMMHD) {
    const int b_config = pin.GetInteger("problem/turbulence", "b_config");
    if (b_config == 3) throw std::runtime_error("Random B fields not implemented yet.");
    if (b_config < 0 || b_config > 2)
      throw std::runtime_error("problem/turbulence/b_config = " + std::to_string(b_config) + " is not supported (0, 1, 2 are)");
  }
  const int gnx[3] = {m.nx[0], m.nx[1], m.nx[2]};
  s->fmft = std::make_unique<FewModesFT>(num_modes, std::move(k_vec), k_peak, sol_weight, t_corr, rseed, gnx);
}

// turbulence::ProblemGenerator (src/pgen/turbulence.cpp:217-370): a MeshData-wide generator --
// the magnetic field is normalised with a global reduction -- so it fills all local blocks at once
int pgen_turbulence(apk_sim *s, std::vector<std::vector<double>> &blocks) {
  const Mesh &m = s->mesh;
  ParameterInput &pin = s->pin;
  const bool mhd = s->pkg.fluid == APK_FLUID_GLMMHD;
  const double gm1 = pin.GetReal("hydro", "gamma") - 1.0;
  const double p0 = pin.GetReal("problem/turbulence", "p0");
  const double rho0 = pin.GetReal("problem/turbulence", "rho0");
  const double Lx = s->xmax[0] - s->xmin[0], Ly = s->xmax[1] - s->xmin[1], Lz = s->xmax[2] - s->xmin[2];
  const double x3min = s->xmin[2];
  const double kz = 2.0 * M_PI / Lz;
  const double vol = s->dx[0] * s->dx[1] * s->dx[2];
  const int nlb = (int)m.local_gids.size();
  blocks.assign(nlb, std::vector<double>((size_t)s->nper, 0.0));
  double b_norm = 0.0;
  if (mhd) {
    const double b0 = pin.GetReal("problem/turbulence", "b0");
    const int b_config = pin.GetInteger("problem/turbulence", "b_config");
    double mag_en_sum = 0.0;
    for (int lb = 0; lb < nlb; ++lb) {
      double x0[3];
      block_origin(s, lb, x0);
      std::vector<double> &u = blocks[lb];
      for (int k = m.ks; k <= m.ke; ++k)
        for (int j = m.js; j <= m.je; ++j)
          for (int i = m.is; i <= m.ie; ++i) {
            double b1 = 0.0;
            if (b_config == 0) b1 = b0;                                                   // uniform
            if (b_config == 1) b1 = (xc(s, x0, 2, k) < x3min + Lz / 2.0) ? b0 : -b0;      // no net flux
            if (b_config == 2) b1 = b0 / std::sqrt(0.5) * std::sin(kz * xc(s, x0, 2, k));  // sin(z)
            u[5 * m.sn + k * m.sk + j * m.sj + i] = b1;
            mag_en_sum += 0.5 * (b1 * b1 + 0.0 + 0.0) * vol;
          }
    }
    if (s->have_comm && s->nranks > 1) {
      if (s->comm.allreduce_sum(s->comm.user, &mag_en_sum, 1) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_sum failed");
    }
    b_norm = std::sqrt(mag_en_sum / (Lx * Ly * Lz) / (0.5 * b0 * b0));
  }
  double v0[3] = {0., 0., 0.};
  if (pin.DoesParameterExist("problem/turbulence", "v0")) {
    std::string txt = pin.GetString("problem/turbulence", "v0");
    for (char &c : txt)
      if (c == ',') c = ' ';
    std::istringstream iss(txt);
    int n = 0;
    double v;
    while (iss >> v) {
      if (n < 3) v0[n] = v;
      ++n;
    }
    if (n != 3) throw std::runtime_error("Initial velocity vector should have three components.");
  }
  for (int lb = 0; lb < nlb; ++lb) {
    std::vector<double> &u = blocks[lb];
    for (int k = m.ks; k <= m.ke; ++k)
      for (int j = m.js; j <= m.je; ++j)
        for (int i = m.is; i <= m.ie; ++i) {
          const int64_t c = k * m.sk + j * m.sj + i;
          u[0 * m.sn + c] = rho0;
          u[1 * m.sn + c] = rho0 * v0[0];
          u[2 * m.sn + c] = rho0 * v0[1];
          u[3 * m.sn + c] = rho0 * v0[2];
          u[4 * m.sn + c] = p0 / gm1 + 0.5 * rho0 * (v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]);
          if (mhd) {
            u[5 * m.sn + c] /= b_norm;
            u[6 * m.sn + c] /= b_norm;
            u[7 * m.sn + c] /= b_norm;
            const double b1 = u[5 * m.sn + c], b2 = u[6 * m.sn + c], b3 = u[7 * m.sn + c];
            u[4 * m.sn + c] += 0.5 * (b1 * b1 + b2 * b2 + b3 * b3);
          }
        }
  }
  return APK_OK;
}


}  // namespace host
}  // namespace apk
