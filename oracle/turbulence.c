/*
 * turbulence.c -- ORACLE (test infrastructure only; see apk_oracle.h).
 * Restatement of the few-modes turbulence driver that BASELINE config 4 enrols as a
 * first-order operator-split source after the last stage (src/hydro/hydro_driver.cpp:559-560):
 *   - FewModesFT (src/utils/few_modes_ft.cpp:30-348): host RNG draws, parabolic injection
 *     spectrum, complex-to-real symmetry, Helmholtz projection, Ornstein-Uhlenbeck update,
 *     per-axis phase tables and the explicit inverse transform;
 *   - turbulence::Perturb (src/pgen/turbulence.cpp:384-470): mean-momentum removal, RMS
 *     normalisation, momentum/energy kick;
 *   - turbulence::ProblemGenerator (src/pgen/turbulence.cpp:217-370), b_config 0/1/2;
 *   - TurbulenceHst Ms / Ma / plasma beta (src/pgen/turbulence.cpp:47-101).
 * The reference draws its random numbers on the host with std::mt19937 and
 * std::uniform_real_distribution<>(-1,1) so that GPU runs are deterministic
 * (few_modes_ft.hpp:39-40, .cpp:205-219); both are restated here (MT19937 per Matsumoto &
 * Nishimura; libstdc++'s generate_canonical<double,53>) and cross-checked against the C++
 * standard library in the tests.
 */
#include "apk_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- std::mt19937 ------------------------------------------------------------------------ */
void orc_mt_seed(orc_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

uint32_t orc_mt_next(orc_mt19937 *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* std::uniform_real_distribution<double>(-1,1)(rng) as libstdc++ implements it:
 * generate_canonical<double,53> consumes two 32-bit draws, sum = x1 + x2 * 2^32, / 2^64 */
double orc_uniform_m1_p1(orc_mt19937 *g) {
  const double r = 4294967296.0;
  double sum = 0.0, tmp = 1.0;
  for (int k = 0; k < 2; ++k) {
    sum += (double)orc_mt_next(g) * tmp;
    tmp *= r;
  }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret * (1.0 - (-1.0)) + (-1.0);
}

/* ---- FewModesFT ------------------------------------------------------------------------------ */
orc_fmft *orc_fmft_create(int num_modes, const double *k_vec /*[3][M]*/, double k_peak,
                          double sol_weight, double t_corr, uint32_t rseed) {
  orc_fmft *f = (orc_fmft *)calloc(1, sizeof(orc_fmft));
  f->num_modes = num_modes;
  f->k_peak = k_peak;
  f->sol_weight = sol_weight;
  f->t_corr = t_corr;
  f->k_vec = (double *)malloc(sizeof(double) * 3 * num_modes);
  memcpy(f->k_vec, k_vec, sizeof(double) * 3 * num_modes);
  f->var_hat = (double *)calloc((size_t)3 * num_modes * 2, sizeof(double));
  f->var_hat_new = (double *)calloc((size_t)3 * num_modes * 2, sizeof(double));
  orc_mt_seed(&f->rng, rseed);
  return f;
}

void orc_fmft_destroy(orc_fmft *f) {
  if (!f) return;
  free(f->k_vec);
  free(f->var_hat);
  free(f->var_hat_new);
  free(f);
}

#define KV(d, m) f->k_vec[(d)*M + (m)]
#define VH(arr, n, m, c) (arr)[(((n)*M + (m)) * 2) + (c)]

/* the spectral part of FewModesFT::Generate (few_modes_ft.cpp:205-320) */
void orc_fmft_evolve(orc_fmft *f, double dt) {
  const int M = f->num_modes;
  double *rnd = (double *)malloc(sizeof(double) * 3 * M * 2);
  for (int n = 0; n < 3; ++n)
    for (int m = 0; m < M; ++m) {
      double v1, v2, v_sqr;
      do {
        v1 = orc_uniform_m1_p1(&f->rng);
        v2 = orc_uniform_m1_p1(&f->rng);
        v_sqr = v1 * v1 + v2 * v2;
      } while (v_sqr >= 1.0 || v_sqr == 0.0);
      rnd[(n * M + m) * 2 + 0] = v1;
      rnd[(n * M + m) * 2 + 1] = v2;
    }
  /* new power spectrum (injection): polar Box-Muller on the drawn pair */
  for (int n = 0; n < 3; ++n)
    for (int m = 0; m < M; ++m) {
      const double kx = KV(0, m), ky = KV(1, m), kz = KV(2, m);
      const double kmag = sqrt(kx * kx + ky * ky + kz * kz);
      double tmp = pow(kmag / f->k_peak, 2.) * (2. - pow(kmag / f->k_peak, 2.));
      if (tmp < 0.) tmp = 0.;
      const double r0 = rnd[(n * M + m) * 2 + 0], r1 = rnd[(n * M + m) * 2 + 1];
      const double v_sqr = r0 * r0 + r1 * r1;
      const double norm = sqrt(-2.0 * log(v_sqr) / v_sqr);
      VH(f->var_hat_new, n, m, 0) = tmp * norm * r0;
      VH(f->var_hat_new, n, m, 1) = tmp * norm * r1;
    }
  /* enforce symmetry of the complex-to-real transform (:250-262) */
  for (int n = 0; n < 3; ++n)
    for (int m = 0; m < M; ++m)
      if (KV(0, m) == 0.) {
        for (int m2 = 0; m2 < m; ++m2)
          if (KV(1, m) == -KV(1, m2) && KV(2, m) == -KV(2, m2)) {
            VH(f->var_hat_new, n, m, 0) = VH(f->var_hat_new, n, m2, 0);
            VH(f->var_hat_new, n, m, 1) = -VH(f->var_hat_new, n, m2, 1);
          }
      }
  /* Helmholtz projection (:264-309) */
  if (f->sol_weight >= 0.0) {
    const double sw = f->sol_weight;
    for (int m = 0; m < M; ++m) {
      double kx = KV(0, m), ky = KV(1, m), kz = KV(2, m);
      double kmag = sqrt(kx * kx + ky * ky + kz * kz);
      if (kmag == 0.) kmag = 1.;
      kx /= kmag;
      ky /= kmag;
      kz /= kmag;
      const double dot_r = VH(f->var_hat_new, 0, m, 0) * kx + VH(f->var_hat_new, 1, m, 0) * ky +
                           VH(f->var_hat_new, 2, m, 0) * kz;
      const double dot_i = VH(f->var_hat_new, 0, m, 1) * kx + VH(f->var_hat_new, 1, m, 1) * ky +
                           VH(f->var_hat_new, 2, m, 1) * kz;
      const double kk[3] = {kx, ky, kz};
      for (int n = 0; n < 3; ++n) {
        VH(f->var_hat_new, n, m, 0) =
            VH(f->var_hat_new, n, m, 0) * sw + (1. - 2. * sw) * dot_r * kk[n];
        VH(f->var_hat_new, n, m, 1) =
            VH(f->var_hat_new, n, m, 1) * sw + (1. - 2. * sw) * dot_i * kk[n];
      }
    }
  }
  /* Ornstein-Uhlenbeck evolution (:311-320) */
  const double c_drift = exp(-dt / f->t_corr);
  const double c_diff = sqrt(1.0 - c_drift * c_drift);
  for (int n = 0; n < 3; ++n)
    for (int m = 0; m < M; ++m)
      for (int c = 0; c < 2; ++c)
        VH(f->var_hat, n, m, c) = VH(f->var_hat, n, m, c) * c_drift + VH(f->var_hat_new, n, m, c) * c_diff;
  free(rnd);
}

/* phase table of one axis (few_modes_ft.cpp:143-190): out[idx][m][2], idx = 0..n-1 are the
 * block's interior cells, g0 = global index of the first one, gn = mesh size along the axis */
void orc_fmft_phases(const orc_fmft *f, int axis, int n, int g0, int gn, double *out) {
  const int M = f->num_modes;
  for (int i = 0; i < n; ++i) {
    const double gi = (double)((i + g0) % gn);
    for (int m = 0; m < M; ++m) {
      const double w = KV(axis, m) * 2. * M_PI / (double)gn;
      /* Kokkos::exp(I*w*gi): I*w = (0*w - 1*0, 0*0 + 1*w) -> times gi; exp(x+iy) = e^x (cos y, sin y) */
      const double arg = w * gi;
      double re = cos(arg), im = sin(arg);
      if (axis == 0 && KV(0, m) == 0.0) { /* u_hat*(k) = u_hat(-k): halve the k_x = 0 modes */
        re = 0.5 * re;
        im = 0.5 * im;
      }
      out[(i * M + m) * 2 + 0] = re;
      out[(i * M + m) * 2 + 1] = im;
    }
  }
}

/* inverse transform on the interior of one block (few_modes_ft.cpp:330-347);
 * acc is [3][Nk][Nj][Ni] */
void orc_fmft_inverse(const orc_fmft *f, const orc_geom *g, const double *ph_i, const double *ph_j,
                      const double *ph_k, double *acc) {
  const int M = f->num_modes;
  const int ni = orc_ni(g), nj = orc_nj(g), nk = orc_nk(g);
  const int is = g->ng, js = (g->nx[1] > 1) ? g->ng : 0, ks = (g->nx[2] > 1) ? g->ng : 0;
  const long sn = (long)ni * nj * nk;
  for (int n = 0; n < 3; ++n)
    for (int k = 0; k < g->nx[2]; ++k)
      for (int j = 0; j < g->nx[1]; ++j)
        for (int i = 0; i < g->nx[0]; ++i) {
          double sum = 0.0;
          for (int m = 0; m < M; ++m) {
            const double ir = ph_i[(i * M + m) * 2], ii = ph_i[(i * M + m) * 2 + 1];
            const double jr = ph_j[(j * M + m) * 2], ji = ph_j[(j * M + m) * 2 + 1];
            const double kr = ph_k[(k * M + m) * 2], ki = ph_k[(k * M + m) * 2 + 1];
            /* phase = phase_i * phase_j * phase_k, complex products left to right */
            const double pr = ir * jr - ii * ji, pi = ir * ji + ii * jr;
            const double qr = pr * kr - pi * ki, qi = pr * ki + pi * kr;
            sum += 2. * (VH(f->var_hat, n, m, 0) * qr - VH(f->var_hat, n, m, 1) * qi);
          }
          acc[n * sn + ((long)(k + ks) * nj + (j + js)) * ni + (i + is)] = sum;
        }
}

/* turbulence::Perturb (src/pgen/turbulence.cpp:384-470) over all blocks */
void orc_turb_perturb(int nblocks, const orc_geom *g, double **cons, double **acc, double dt,
                      double accel_rms, double box_volume) {
  const int ni = orc_ni(g), nj = orc_nj(g), nk = orc_nk(g);
  const int is = g->ng, js = (g->nx[1] > 1) ? g->ng : 0, ks = (g->nx[2] > 1) ? g->ng : 0;
  const long sn = (long)ni * nj * nk;
  const double vol = g->dx[0] * g->dx[1] * g->dx[2];
  double sums[4] = {0, 0, 0, 0};
#define CELL(k, j, i) (((long)((k) + ks) * nj + ((j) + js)) * ni + ((i) + is))
  for (int b = 0; b < nblocks; ++b)
    for (int k = 0; k < g->nx[2]; ++k)
      for (int j = 0; j < g->nx[1]; ++j)
        for (int i = 0; i < g->nx[0]; ++i) {
          const long c = CELL(k, j, i);
          const double den = cons[b][ORC_IDN * sn + c];
          sums[0] += den * vol;
          sums[1] += den * acc[b][0 * sn + c] * vol;
          sums[2] += den * acc[b][1 * sn + c] * vol;
          sums[3] += den * acc[b][2 * sn + c] * vol;
        }
  double ampl = 0.0;
  for (int b = 0; b < nblocks; ++b)
    for (int n = 0; n < 3; ++n)
      for (int k = 0; k < g->nx[2]; ++k)
        for (int j = 0; j < g->nx[1]; ++j)
          for (int i = 0; i < g->nx[0]; ++i) {
            const long c = CELL(k, j, i);
            acc[b][n * sn + c] -= sums[n + 1] / sums[0];
            ampl += acc[b][n * sn + c] * acc[b][n * sn + c] * vol;
          }
  const double norm = accel_rms / sqrt(ampl / box_volume);
  for (int b = 0; b < nblocks; ++b)
    for (int k = 0; k < g->nx[2]; ++k)
      for (int j = 0; j < g->nx[1]; ++j)
        for (int i = 0; i < g->nx[0]; ++i) {
          const long c = CELL(k, j, i);
          double *u = cons[b];
          double a0 = acc[b][0 * sn + c] * norm, a1 = acc[b][1 * sn + c] * norm,
                 a2 = acc[b][2 * sn + c] * norm;
          acc[b][0 * sn + c] = a0;
          acc[b][1 * sn + c] = a1;
          acc[b][2 * sn + c] = a2;
          const double qa = dt * u[ORC_IDN * sn + c];
          u[ORC_IEN * sn + c] +=
              (u[ORC_IM1 * sn + c] * dt * a0 + u[ORC_IM2 * sn + c] * dt * a1 +
               u[ORC_IM3 * sn + c] * dt * a2 +
               (a0 * a0 + a1 * a1 + a2 * a2) * qa * qa / (2 * u[ORC_IDN * sn + c]));
          u[ORC_IM1 * sn + c] += qa * a0;
          u[ORC_IM2 * sn + c] += qa * a1;
          u[ORC_IM3 * sn + c] += qa * a2;
        }
#undef CELL
}

/* TurbulenceHst (src/pgen/turbulence.cpp:47-101): out[3] += Ms, Ma, plasma beta sums */
void orc_turb_history(const orc_geom *g, int fluid, double gamma, const double *prim, double *out3) {
  const int ni = orc_ni(g), nj = orc_nj(g), nk = orc_nk(g);
  const int is = g->ng, js = (g->nx[1] > 1) ? g->ng : 0, ks = (g->nx[2] > 1) ? g->ng : 0;
  const long sn = (long)ni * nj * nk;
  const double vol = g->dx[0] * g->dx[1] * g->dx[2];
  for (int k = 0; k < g->nx[2]; ++k)
    for (int j = 0; j < g->nx[1]; ++j)
      for (int i = 0; i < g->nx[0]; ++i) {
        const long c = ((long)(k + ks) * nj + (j + js)) * ni + (i + is);
        const double d = prim[ORC_IDN * sn + c], p = prim[ORC_IPR * sn + c];
        const double v1 = prim[ORC_IV1 * sn + c], v2 = prim[ORC_IV2 * sn + c], v3 = prim[ORC_IV3 * sn + c];
        const double vel2 = (v1 * v1 + v2 * v2 + v3 * v3);
        const double c_s = sqrt(gamma * p / d);
        const double e_kin = 0.5 * d * vel2;
        out3[0] += sqrt(vel2) / c_s * vol;
        if (fluid == ORC_FLUID_GLMMHD) {
          const double b1 = prim[ORC_IB1 * sn + c], b2 = prim[ORC_IB2 * sn + c], b3 = prim[ORC_IB3 * sn + c];
          const double e_mag = 0.5 * (b1 * b1 + b2 * b2 + b3 * b3);
          out3[1] += sqrt(e_kin / e_mag) * vol;
          out3[2] += p / e_mag * vol;
        }
      }
}
