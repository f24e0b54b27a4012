/*
 * sim.c -- ORACLE (test infrastructure only; see apk_oracle.h).
 * Mini-driver reproducing the per-stage ordering of HydroDriver::MakeTaskCollection
 * (src/hydro/hydro_driver.cpp:347-673) on a uniform mesh of equal meshblocks, plus the
 * Parthenon-side semantics it depends on (un-vendored => unpinned):
 * low-storage integrator coefficients, dt control, ghost fill order, cell centres.
 */
#include "apk_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DBL_HUGE 1.7976931348623157e308

struct orc_sim {
  orc_sim_params p;
  orc_geom g;
  int nb[3], nblocks;
  long nper; /* doubles per field per block */
  double **cons, **prim, **u1, **flux[3];
  double time, dt, c_h, mindx, dt_hyp;
  int ncycle;
  long fofc_total;
  /* turbulence driver (NULL unless orc_pgen_turbulence was called) */
  orc_fmft *fmft;
  /* circularly polarised Alfven wave (orc_pgen_cpaw) */
  struct {
    double den, pres, gm1, b_par, b_perp, v_perp, v_par, fac;
    double sin_a2, cos_a2, sin_a3, cos_a3, lambda, k_par;
  } cpaw;
  double accel_rms;
  double **acc, **ph_i, **ph_j, **ph_k;
  /* linear wave state (src/pgen/linear_wave.cpp globals) */
  double lw_sin_a2, lw_cos_a2, lw_sin_a3, lw_cos_a3, lw_kpar, lw_lambda;
  double lw_d0, lw_p0, lw_u0, lw_gam, lw_gm1, lw_ev[5], lw_rem[5][5];
  /* MHD linear wave (src/pgen/linear_wave_mhd.cpp globals): background field in the wave frame,
   * transverse field amplitudes of the chosen family, 7-wave eigensystem */
  double lwm_bx0, lwm_by0, lwm_bz0, lwm_dby, lwm_dbz, lwm_ev[7], lwm_rem[7][7];
};

/* SURVEY.md App. A.2 */
int orc_integrator_coeffs(int integrator, double *beta, double *gam0, double *gam1) {
  switch (integrator) {
  case ORC_INT_RK1:
    beta[0] = 1.0; gam0[0] = 0.0; gam1[0] = 1.0;
    return 1;
  case ORC_INT_RK2:
    beta[0] = 1.0; gam0[0] = 0.0; gam1[0] = 1.0;
    beta[1] = 0.5; gam0[1] = 0.5; gam1[1] = 0.5;
    return 2;
  case ORC_INT_VL2:
    beta[0] = 0.5; gam0[0] = 0.0; gam1[0] = 1.0;
    beta[1] = 1.0; gam0[1] = 0.0; gam1[1] = 1.0;
    return 2;
  case ORC_INT_RK3:
    beta[0] = 1.0;       gam0[0] = 0.0;       gam1[0] = 1.0;
    beta[1] = 0.25;      gam0[1] = 0.25;      gam1[1] = 0.75;
    beta[2] = 2.0 / 3.0; gam0[2] = 2.0 / 3.0; gam1[2] = 1.0 / 3.0;
    return 3;
  default:
    return 0;
  }
}

orc_sim *orc_sim_create(const orc_sim_params *p) {
  orc_sim *s = (orc_sim *)calloc(1, sizeof(orc_sim));
  s->p = *p;
  for (int d = 0; d < 3; ++d) {
    if (p->nx[d] % p->mb[d] != 0) {
      free(s);
      return NULL;
    }
    s->nb[d] = p->nx[d] / p->mb[d];
    s->g.nx[d] = p->mb[d];
    s->g.dx[d] = (p->xmax[d] - p->xmin[d]) / (double)p->nx[d];
  }
  s->g.ng = p->ng;
  s->g.nhydro = (p->fluid == ORC_FLUID_EULER) ? ORC_NHYDRO : ORC_NGLMMHD;
  s->g.nvar = s->g.nhydro + p->nscalars;
  s->nblocks = s->nb[0] * s->nb[1] * s->nb[2];
  s->nper = orc_ncell(&s->g) * s->g.nvar;
  s->cons = (double **)calloc(s->nblocks, sizeof(double *));
  s->prim = (double **)calloc(s->nblocks, sizeof(double *));
  s->u1 = (double **)calloc(s->nblocks, sizeof(double *));
  for (int d = 0; d < 3; ++d) s->flux[d] = (double **)calloc(s->nblocks, sizeof(double *));
  for (int b = 0; b < s->nblocks; ++b) {
    s->cons[b] = (double *)calloc(s->nper, sizeof(double));
    s->prim[b] = (double *)calloc(s->nper, sizeof(double));
    s->u1[b] = (double *)calloc(s->nper, sizeof(double));
    for (int d = 0; d < 3; ++d) s->flux[d][b] = (double *)calloc(s->nper, sizeof(double));
  }
  s->mindx = DBL_HUGE;
  s->dt_hyp = DBL_HUGE;
  s->dt = DBL_HUGE;
#ifdef _OPENMP
  if (p->nthreads > 0) omp_set_num_threads(p->nthreads);
#endif
  return s;
}

void orc_sim_destroy(orc_sim *s) {
  if (!s) return;
  if (s->fmft) {
    for (int b = 0; b < s->nblocks; ++b) {
      free(s->acc[b]);
      free(s->ph_i[b]);
      free(s->ph_j[b]);
      free(s->ph_k[b]);
    }
    free(s->acc);
    free(s->ph_i);
    free(s->ph_j);
    free(s->ph_k);
    orc_fmft_destroy(s->fmft);
  }
  for (int b = 0; b < s->nblocks; ++b) {
    free(s->cons[b]);
    free(s->prim[b]);
    free(s->u1[b]);
    for (int d = 0; d < 3; ++d) free(s->flux[d][b]);
  }
  free(s->cons);
  free(s->prim);
  free(s->u1);
  for (int d = 0; d < 3; ++d) free(s->flux[d]);
  free(s);
}

int orc_sim_nblocks(const orc_sim *s) { return s->nblocks; }
void orc_sim_block_geom(const orc_sim *s, orc_geom *g) { *g = s->g; }
double *orc_sim_cons(orc_sim *s, int b) { return s->cons[b]; }
double *orc_sim_prim(orc_sim *s, int b) { return s->prim[b]; }
double orc_sim_time(const orc_sim *s) { return s->time; }
double orc_sim_dt(const orc_sim *s) { return s->dt; }
double orc_sim_c_h(const orc_sim *s) { return s->c_h; }
long orc_sim_fofc_count(const orc_sim *s) { return s->fofc_total; }

static void block_coords(const orc_sim *s, int b, int bc[3]) {
  bc[0] = b % s->nb[0];
  bc[1] = (b / s->nb[0]) % s->nb[1];
  bc[2] = b / (s->nb[0] * s->nb[1]);
}

/* x0[d] carries the GLOBAL index of the block's first interior cell (as a double) so that
 * cell centres depend on the global cell index only, not on the block decomposition. */
void orc_sim_block_origin(const orc_sim *s, int b, double x0[3]) {
  int bc[3];
  block_coords(s, b, bc);
  for (int d = 0; d < 3; ++d) x0[d] = (double)(bc[d] * s->p.mb[d]);
}

/* cell centre: Xc = xmin + (global_index + 1/2) dx */
static inline double xc(const orc_sim *s, const double x0[3], int d, int idx) {
  const int ng = (s->g.nx[d] > 1) ? s->g.ng : 0;
  return s->p.xmin[d] + ((x0[d] + (double)(idx - ng)) + 0.5) * s->g.dx[d];
}

typedef struct {
  int is, ie, js, je, ks, ke, ni, nj, nk;
  long sj, sk, sn;
} sb_t;

static sb_t sim_bounds(const orc_geom *g) {
  sb_t b;
  b.ni = orc_ni(g);
  b.nj = orc_nj(g);
  b.nk = orc_nk(g);
  b.is = g->ng;
  b.ie = g->ng + g->nx[0] - 1;
  b.js = (g->nx[1] > 1) ? g->ng : 0;
  b.je = (g->nx[1] > 1) ? g->ng + g->nx[1] - 1 : 0;
  b.ks = (g->nx[2] > 1) ? g->ng : 0;
  b.ke = (g->nx[2] > 1) ? g->ng + g->nx[2] - 1 : 0;
  b.sj = b.ni;
  b.sk = (long)b.ni * b.nj;
  b.sn = b.sk * b.nk;
  return b;
}

#define SAT(arr, n, k, j, i) (arr)[(n)*bb.sn + (k)*bb.sk + (j)*bb.sj + (i)]

/* ---------------------------------------------------------------------------------------
 * Ghost fill (Parthenon bvals, un-vendored):
 * (1) every ghost cell whose global index lies inside the mesh (after periodic wrap) is
 *     copied from the owning block's interior;
 * (2) physical (non-periodic) boundaries are then applied in the order inner_x1, outer_x1,
 *     inner_x2, outer_x2, inner_x3, outer_x3, each over the ENTIRE transverse extent. */
void orc_sim_exchange_ghosts(orc_sim *s) {
  const sb_t bb = sim_bounds(&s->g);
  const int ng = s->g.ng;
  const int nvar = s->g.nvar;
  const int act[3] = {1, s->g.nx[1] > 1, s->g.nx[2] > 1};
#pragma omp parallel for schedule(static)
  for (int b = 0; b < s->nblocks; ++b) {
    int bc[3];
    block_coords(s, b, bc);
    double *u = s->cons[b];
    for (int k = 0; k < bb.nk; ++k)
      for (int j = 0; j < bb.nj; ++j)
        for (int i = 0; i < bb.ni; ++i) {
          const int loc[3] = {i, j, k};
          int interior = 1, ok = 1;
          int src_b[3], src_l[3];
          for (int d = 0; d < 3; ++d) {
            if (!act[d]) {
              src_b[d] = 0;
              src_l[d] = 0;
              continue;
            }
            const int li = loc[d] - ng; /* block-local interior index */
            if (li < 0 || li >= s->p.mb[d]) interior = 0;
            int gi = bc[d] * s->p.mb[d] + li;
            if (gi < 0 || gi >= s->p.nx[d]) {
              const int periodic = (gi < 0) ? (s->p.bc_inner[d] == ORC_BC_PERIODIC)
                                            : (s->p.bc_outer[d] == ORC_BC_PERIODIC);
              if (!periodic) {
                ok = 0;
                break;
              }
              gi = ((gi % s->p.nx[d]) + s->p.nx[d]) % s->p.nx[d];
            }
            src_b[d] = gi / s->p.mb[d];
            src_l[d] = gi % s->p.mb[d] + ng;
          }
          if (interior || !ok) continue;
          const int sb = src_b[0] + s->nb[0] * (src_b[1] + s->nb[1] * src_b[2]);
          const double *v = s->cons[sb];
          for (int n = 0; n < nvar; ++n)
            SAT(u, n, k, j, i) = SAT(v, n, src_l[2], src_l[1], src_l[0]);
        }
  }
  /* physical boundaries: outflow = copy last active cell (docs/input.md:416-419) */
#pragma omp parallel for schedule(static)
  for (int b = 0; b < s->nblocks; ++b) {
    int bc[3];
    block_coords(s, b, bc);
    double *u = s->cons[b];
    const int lo[3] = {bb.is, bb.js, bb.ks};
    const int hi[3] = {bb.ie, bb.je, bb.ke};
    const int ext[3] = {bb.ni, bb.nj, bb.nk};
    for (int d = 0; d < 3; ++d) {
      if (!act[d]) continue;
      for (int side = 0; side < 2; ++side) {
        const int at_edge = side ? (bc[d] == s->nb[d] - 1) : (bc[d] == 0);
        const int kind = side ? s->p.bc_outer[d] : s->p.bc_inner[d];
        if (!at_edge || kind == ORC_BC_PERIODIC) continue;
        const int g0 = side ? hi[d] + 1 : 0;
        const int g1 = side ? ext[d] - 1 : lo[d] - 1;
        for (int n = 0; n < nvar; ++n)
          for (int k = 0; k < bb.nk; ++k)
            for (int j = 0; j < bb.nj; ++j)
              for (int i = 0; i < bb.ni; ++i) {
                const int loc[3] = {i, j, k};
                if (loc[d] < g0 || loc[d] > g1) continue;
                int src[3] = {i, j, k};
                double sign = 1.0;
                if (kind == ORC_BC_OUTFLOW) {
                  src[d] = side ? hi[d] : lo[d];
                } else { /* reflecting: src/bvals/boundary_conditions_apk.hpp:38-85 */
                  src[d] = side ? (2 * hi[d] + 1 - loc[d]) : (2 * lo[d] - 1 - loc[d]);
                  if (n == ORC_IM1 + d) sign = -1.0;
                }
                SAT(u, n, k, j, i) = sign * SAT(u, n, src[2], src[1], src[0]);
              }
      }
    }
  }
}

void orc_sim_fill_derived(orc_sim *s) {
#pragma omp parallel for schedule(static)
  for (int b = 0; b < s->nblocks; ++b)
    orc_cons_to_prim(&s->g, s->p.fluid, &s->p.eos, s->cons[b], s->prim[b]);
}

/* Hydro::EstimateTimestep over all blocks (hydro.cpp:913-977, hyperbolic part only) */
static double estimate_timestep(orc_sim *s) {
  double m = DBL_HUGE;
#pragma omp parallel for schedule(static) reduction(min : m)
  for (int b = 0; b < s->nblocks; ++b) {
    const double v = orc_estimate_dt_hyp(&s->g, s->p.fluid, &s->p.eos, s->prim[b]);
    if (v < m) m = v;
  }
  const double dt = s->p.cfl * m;
  if (s->p.fluid == ORC_FLUID_GLMMHD) {
    if (dt < s->dt_hyp) s->dt_hyp = dt; /* hydro.cpp:903-908 */
  }
  return dt;
}

/* EvolutionDriver::SetGlobalTimeStep */
static void set_global_dt(orc_sim *s, double dt_est, double tlim) {
  double dt = s->dt;
  if (dt < 0.1 * DBL_HUGE) dt *= 2.0;
  if (dt_est < dt) dt = dt_est;
  if (s->time < tlim && (tlim - s->time) < dt) dt = tlim - s->time;
  s->dt = dt;
}

void orc_sim_initialize(orc_sim *s) {
  orc_sim_exchange_ghosts(s);
  orc_sim_fill_derived(s);
  const double est = estimate_timestep(s);
  s->time = 0.0;
  s->ncycle = 0;
  s->dt = DBL_HUGE;
  set_global_dt(s, est, DBL_HUGE);
}

/* Hydro::PreStepMeshUserWorkInLoop (hydro.cpp:102-143) */
static void pre_step(orc_sim *s) {
  if (s->p.fluid != ORC_FLUID_GLMMHD) return;
  double mindx = s->g.dx[0];
  if (s->g.nx[1] > 1) mindx = fmin(mindx, s->g.dx[1]);
  if (s->g.nx[2] > 1) mindx = fmin(mindx, s->g.dx[2]);
  s->mindx = mindx;
  s->c_h = s->p.cfl * s->mindx / s->dt_hyp;
}

double orc_sim_step(orc_sim *s, double tlim) {
  double beta[4], gam0[4], gam1[4];
  const int nstages = orc_integrator_coeffs(s->p.integrator, beta, gam0, gam1);
  if (s->time < tlim && (tlim - s->time) < s->dt) s->dt = tlim - s->time;
  pre_step(s);
  const double dt = s->dt;
  for (int stage = 1; stage <= nstages; ++stage) {
    const double g0 = gam0[stage - 1], g1 = gam1[stage - 1];
    const double beta_dt = beta[stage - 1] * dt;
    /* vl2 predictor uses donor cell with the same Riemann solver (hydro.cpp:457-463) */
    const int recon =
        (stage == 1 && s->p.integrator == ORC_INT_VL2) ? ORC_RC_DC : s->p.recon;
    long fofc = 0;
#pragma omp parallel for schedule(static) reduction(+ : fofc)
    for (int b = 0; b < s->nblocks; ++b) {
      if (stage == 1) memcpy(s->u1[b], s->cons[b], sizeof(double) * s->nper);
      if (s->p.riemann == ORC_RS_LLF) {
        orc_calculate_fluxes_tight(&s->g, s->p.fluid, &s->p.eos, s->c_h, s->prim[b],
                                   s->flux[0][b], s->flux[1][b], s->flux[2][b]);
      } else {
        orc_calculate_fluxes(&s->g, s->p.fluid, recon, s->p.riemann, &s->p.eos, s->c_h,
                             s->prim[b], s->flux[0][b], s->flux[1][b], s->flux[2][b]);
      }
      if (s->p.first_order_flux_correct) {
        fofc += orc_first_order_flux_correct(&s->g, s->p.fluid, &s->p.eos, s->c_h, s->cons[b],
                                             s->prim[b], s->u1[b], s->flux[0][b],
                                             s->flux[1][b], s->flux[2][b], g0, g1, beta_dt);
      }
      orc_update_flux_div(&s->g, s->cons[b], s->u1[b], s->flux[0][b], s->flux[1][b],
                          s->flux[2][b], g0, g1, beta_dt);
      if (s->p.fluid == ORC_FLUID_GLMMHD) {
        orc_dedner_source(&s->g, s->p.dedner_extended, s->p.glmmhd_alpha, s->c_h, s->mindx,
                          beta_dt, s->cons[b], s->prim[b]);
      }
    }
    s->fofc_total += fofc;
    if (stage == nstages && s->fmft) {
      /* AddSplitSourcesFirstOrder -> turbulence::Driving(md, tm, tm.dt) (hydro_driver.cpp:559-560,
       * src/pgen/turbulence.cpp:476-482): Generate then Perturb */
      orc_fmft_evolve(s->fmft, dt);
      for (int b = 0; b < s->nblocks; ++b)
        orc_fmft_inverse(s->fmft, &s->g, s->ph_i[b], s->ph_j[b], s->ph_k[b], s->acc[b]);
      const double box = (s->p.xmax[0] - s->p.xmin[0]) * (s->p.xmax[1] - s->p.xmin[1]) *
                         (s->p.xmax[2] - s->p.xmin[2]);
      orc_turb_perturb(s->nblocks, &s->g, s->cons, s->acc, dt, s->accel_rms, box);
    }
    orc_sim_exchange_ghosts(s);
    orc_sim_fill_derived(s);
    if (stage == nstages) {
      /* reset reduction params (hydro_driver.cpp:589-603) then EstimateTimestep */
      if (s->p.fluid == ORC_FLUID_GLMMHD) {
        s->mindx = DBL_HUGE;
        s->dt_hyp = DBL_HUGE;
      }
    }
  }
  s->time += dt;
  s->ncycle += 1;
  const double est = estimate_timestep(s);
  set_global_dt(s, est, tlim);
  return dt;
}

int orc_sim_run(orc_sim *s, double tlim, int nlim) {
  int n = 0;
  while (s->time < tlim && (nlim < 0 || n < nlim)) {
    orc_sim_step(s, tlim);
    ++n;
  }
  return n;
}

void orc_sim_history(orc_sim *s, double *out8) {
  for (int q = 0; q < 8; ++q) out8[q] = 0.0;
  for (int b = 0; b < s->nblocks; ++b) {
    double o[8];
    orc_history(&s->g, s->p.fluid, s->cons[b], o);
    for (int q = 0; q < 8; ++q) out8[q] += o[q];
  }
}

void orc_sim_gather_cons(orc_sim *s, double *out) {
  const sb_t bb = sim_bounds(&s->g);
  const long NX = s->p.nx[0], NY = s->p.nx[1], NZ = s->p.nx[2];
  for (int b = 0; b < s->nblocks; ++b) {
    int bc[3];
    block_coords(s, b, bc);
    for (int n = 0; n < s->g.nvar; ++n)
      for (int k = bb.ks; k <= bb.ke; ++k)
        for (int j = bb.js; j <= bb.je; ++j)
          for (int i = bb.is; i <= bb.ie; ++i) {
            const long gi = bc[0] * s->p.mb[0] + (i - bb.is);
            const long gj = bc[1] * s->p.mb[1] + (j - bb.js);
            const long gk = bc[2] * s->p.mb[2] + (k - bb.ks);
            out[((n * NZ + gk) * NY + gj) * NX + gi] = SAT(s->cons[b], n, k, j, i);
          }
  }
}

/* ------------------------- problem generators ------------------------------------------ */

/* hydro eigensystem, src/pgen/linear_wave.cpp:421-500 (only ev and rem are needed) */
static void lw_eigensystem(double gm1, double v1, double v2, double v3, double h,
                           double ev[5], double rem[5][5]) {
  const double vsq = v1 * v1 + v2 * v2 + v3 * v3;
  const double asq = gm1 * fmax((h - 0.5 * vsq), ORC_TINY_NUMBER);
  const double a = sqrt(asq);
  ev[0] = v1 - a;
  ev[1] = v1;
  ev[2] = v1;
  ev[3] = v1;
  ev[4] = v1 + a;
  const double col[5][5] = {{1.0, v1 - a, v2, v3, h - v1 * a},
                            {0.0, 0.0, 1.0, 0.0, v2},
                            {0.0, 0.0, 0.0, 1.0, v3},
                            {1.0, v1, v2, v3, 0.5 * vsq},
                            {1.0, v1 + a, v2, v3, h + v1 * a}};
  for (int c = 0; c < 5; ++c)
    for (int r = 0; r < 5; ++r) rem[r][c] = col[c][r];
}

/* src/pgen/linear_wave.cpp:72-176 (InitUserMeshData) */
static void lw_setup(orc_sim *s, double vflow) {
  const double x1size = s->p.xmax[0] - s->p.xmin[0];
  const double x2size = s->p.xmax[1] - s->p.xmin[1];
  const double x3size = s->p.xmax[2] - s->p.xmin[2];
  s->lw_gam = s->p.eos.gamma;
  s->lw_gm1 = s->lw_gam - 1.0;
  double ang_3 = atan(x1size / x2size);
  s->lw_sin_a3 = sin(ang_3);
  s->lw_cos_a3 = cos(ang_3);
  double ang_2 = atan(0.5 * (x1size * s->lw_cos_a3 + x2size * s->lw_sin_a3) / x3size);
  s->lw_sin_a2 = sin(ang_2);
  s->lw_cos_a2 = cos(ang_2);
  const double x1 = x1size * s->lw_cos_a2 * s->lw_cos_a3;
  const double x2 = x2size * s->lw_cos_a2 * s->lw_sin_a3;
  const double x3 = x3size * s->lw_sin_a2;
  const int f2 = (s->p.nx[1] > 1) ? 1 : 0;
  const int f3 = (s->p.nx[2] > 1) ? 1 : 0;
  double lambda = x1;
  if (f2 && ang_3 != 0.0) lambda = fmin(lambda, x2);
  if (f3 && ang_2 != 0.0) lambda = fmin(lambda, x3);
  s->lw_lambda = lambda;
  s->lw_kpar = 2.0 * (M_PI) / lambda;
  s->lw_d0 = 1.0;
  s->lw_u0 = vflow;
  s->lw_p0 = 1.0 / s->lw_gam;
  const double v0 = 0.0, w0 = 0.0;
  const double h0 = ((s->lw_p0 / s->lw_gm1 +
                      0.5 * s->lw_d0 * (s->lw_u0 * s->lw_u0 + v0 * v0 + w0 * w0)) +
                     s->lw_p0) /
                    s->lw_d0;
  lw_eigensystem(s->lw_gm1, s->lw_u0, v0, w0, h0, s->lw_ev, s->lw_rem);
}

/* analytic conserved state at a cell centre (linear_wave.cpp:355-373 == :206-226) */
static void lw_state(const orc_sim *s, int wave_flag, double amp, double vflow, double x1,
                     double x2, double x3, double u[5]) {
  const double x = s->lw_cos_a2 * (x1 * s->lw_cos_a3 + x2 * s->lw_sin_a3) + x3 * s->lw_sin_a2;
  const double sn = sin(s->lw_kpar * x);
  u[ORC_IDN] = s->lw_d0 + amp * sn * s->lw_rem[0][wave_flag];
  const double mx = s->lw_d0 * vflow + amp * sn * s->lw_rem[1][wave_flag];
  const double my = amp * sn * s->lw_rem[2][wave_flag];
  const double mz = amp * sn * s->lw_rem[3][wave_flag];
  u[ORC_IM1] = mx * s->lw_cos_a2 * s->lw_cos_a3 - my * s->lw_sin_a3 -
               mz * s->lw_sin_a2 * s->lw_cos_a3;
  u[ORC_IM2] = mx * s->lw_cos_a2 * s->lw_sin_a3 + my * s->lw_cos_a3 -
               mz * s->lw_sin_a2 * s->lw_sin_a3;
  u[ORC_IM3] = mx * s->lw_sin_a2 + mz * s->lw_cos_a2;
  u[ORC_IEN] = s->lw_p0 / s->lw_gm1 + 0.5 * s->lw_d0 * s->lw_u0 * s->lw_u0 +
               amp * sn * s->lw_rem[4][wave_flag];
}

double orc_pgen_linear_wave(orc_sim *s, int wave_flag, double amp, double vflow) {
  lw_setup(s, vflow);
  const sb_t bb = sim_bounds(&s->g);
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          double u[5];
          lw_state(s, wave_flag, amp, vflow, xc(s, x0, 0, i), xc(s, x0, 1, j),
                   xc(s, x0, 2, k), u);
          for (int n = 0; n < 5; ++n) SAT(s->cons[b], n, k, j, i) = u[n];
        }
  }
  /* "test = true": tlim is reinterpreted as number of periods (linear_wave.cpp:169-175) */
  return s->lw_lambda / fabs(s->lw_ev[wave_flag]);
}

/* src/pgen/linear_wave.cpp:183-335 */
double orc_linear_wave_errors(orc_sim *s, int wave_flag, double amp, double vflow, double *l1,
                              double *maxerr) {
  const sb_t bb = sim_bounds(&s->g);
  const double cellvol = s->g.dx[0] * s->g.dx[1] * s->g.dx[2];
  for (int n = 0; n < 5; ++n) l1[n] = maxerr[n] = 0.0;
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          double u[5];
          lw_state(s, wave_flag, amp, vflow, xc(s, x0, 0, i), xc(s, x0, 1, j),
                   xc(s, x0, 2, k), u);
          for (int n = 0; n < 5; ++n) {
            const double e = fabs(u[n] - SAT(s->cons[b], n, k, j, i));
            l1[n] += e * cellvol;
            if (e > maxerr[n]) maxerr[n] = e;
          }
        }
  }
  const double vol = (s->p.xmax[0] - s->p.xmin[0]) * (s->p.xmax[1] - s->p.xmin[1]) *
                     (s->p.xmax[2] - s->p.xmin[2]);
  double rms = 0.0;
  for (int n = 0; n < 5; ++n) {
    l1[n] = l1[n] / vol;
    rms += l1[n] * l1[n];
  }
  return sqrt(rms);
}

/* ---- MHD linear wave (src/pgen/linear_wave_mhd.cpp) ------------------------------------------- */
/* adiabatic MHD eigensystem in the conserved variables (d, mx, my, mz, E, by, bz), eigenvalues and RIGHT
 * eigenvectors only (linear_wave_mhd.cpp:486-625; the left eigenvectors :627-710 are never read by the
 * problem generator).  x, y are the reference's xfact / yfact. */
static void lwm_eigensystem(double gm1, double d, double v1, double v2, double v3, double h, double b1,
                            double b2, double b3, double x, double y, double ev[7], double rem[7][7]) {
  const double vsq = v1 * v1 + v2 * v2 + v3 * v3;
  const double btsq = b2 * b2 + b3 * b3;
  const double bt_starsq = (gm1 - (gm1 - 1.0) * y) * btsq;
  const double vaxsq = b1 * b1 / d;
  const double hp = h - (vaxsq + btsq / d);
  const double twid_asq = fmax((gm1 * (hp - 0.5 * vsq) - (gm1 - 1.0) * x), ORC_TINY_NUMBER);
  /* fast and slow speeds (eq. B18) */
  const double ct2 = bt_starsq / d;
  const double tsum = vaxsq + ct2 + twid_asq;
  const double tdif = vaxsq + ct2 - twid_asq;
  const double cf2_cs2 = sqrt(tdif * tdif + 4.0 * twid_asq * ct2);
  const double cfsq = 0.5 * (tsum + cf2_cs2);
  const double cf = sqrt(cfsq);
  const double cssq = twid_asq * vaxsq / cfsq;
  const double cs = sqrt(cssq);
  /* betas (eqs. A17, B20, B28) */
  const double bt = sqrt(btsq);
  const double bt_star = sqrt(bt_starsq);
  double bet2 = 1.0, bet3 = 0.0;
  if (bt != 0.0) {
    bet2 = b2 / bt;
    bet3 = b3 / bt;
  }
  const double bet2_star = bet2 / sqrt(gm1 - (gm1 - 1.0) * y);
  const double bet3_star = bet3 / sqrt(gm1 - (gm1 - 1.0) * y);
  const double bet_starsq = bet2_star * bet2_star + bet3_star * bet3_star;
  const double vbet = v2 * bet2_star + v3 * bet3_star;
  /* alphas (eq. A16) */
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
    alpha_f = sqrt((twid_asq - cssq) / (cfsq - cssq));
    alpha_s = sqrt((cfsq - twid_asq) / (cfsq - cssq));
  }
  /* Q and A (eqs. A14-15) */
  const double sqrtd = sqrt(d);
  const double isqrtd = 1.0 / sqrtd;
  const double sgn = (b1 < 0.0) ? -1.0 : 1.0; /* Parthenon SIGN */
  const double twid_a = sqrt(twid_asq);
  const double qf = cf * alpha_f * sgn;
  const double qs = cs * alpha_s * sgn;
  const double af_prime = twid_a * alpha_f * isqrtd;
  const double as_prime = twid_a * alpha_s * isqrtd;
  const double afpbb = af_prime * bt_star * bet_starsq;
  const double aspbb = as_prime * bt_star * bet_starsq;
  const double vax = sqrt(vaxsq);
  /* eigenvalues (eq. B17): fast-, Alfven-, slow-, entropy, slow+, Alfven+, fast+ */
  const double lam[7] = {v1 - cf, v1 - vax, v1 - cs, v1, v1 + cs, v1 + vax, v1 + cf};
  for (int w = 0; w < 7; ++w) ev[w] = lam[w];
  /* right eigenvectors (eq. B21), one ROW of this table per wave family = one COLUMN of rem */
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

/* InitUserMeshData (linear_wave_mhd.cpp:68-172): the hydro wave's geometry (lw_setup) and the magnetised
 * background d0 = 1, p0 = 1/gamma, B = (1, sqrt 2, 1/2) in the wave frame */
static void lwm_setup(orc_sim *s, double vflow) {
  lw_setup(s, vflow); /* angles, lambda, k_par, d0, u0, p0: the same expressions (:91-143 == linear_wave.cpp:86-140) */
  s->lwm_bx0 = 1.0;
  s->lwm_by0 = sqrt(2.0);
  s->lwm_bz0 = 0.5;
  const double d0 = s->lw_d0, u0 = s->lw_u0, p0 = s->lw_p0, v0 = 0.0, w0 = 0.0;
  double h0 = ((p0 / s->lw_gm1 + 0.5 * d0 * (u0 * u0 + v0 * v0 + w0 * w0)) + p0) / d0;
  h0 += (s->lwm_bx0 * s->lwm_bx0 + s->lwm_by0 * s->lwm_by0 + s->lwm_bz0 * s->lwm_bz0) / d0;
  lwm_eigensystem(s->lw_gm1, d0, u0, v0, w0, h0, s->lwm_bx0, s->lwm_by0, s->lwm_bz0, 0.0, 1.0, s->lwm_ev, s->lwm_rem);
}

/* vector potential in the gauge Ax = 0 (linear_wave_mhd.cpp:443-479) */
static void lwm_A(const orc_sim *s, double x1, double x2, double x3, double A[3]) {
  const double x = x1 * s->lw_cos_a2 * s->lw_cos_a3 + x2 * s->lw_cos_a2 * s->lw_sin_a3 + x3 * s->lw_sin_a2;
  const double y = -x1 * s->lw_sin_a3 + x2 * s->lw_cos_a3;
  const double Ay = s->lwm_bz0 * x - (s->lwm_dbz / s->lw_kpar) * cos(s->lw_kpar * (x));
  const double Az = -s->lwm_by0 * x + (s->lwm_dby / s->lw_kpar) * cos(s->lw_kpar * (x)) + s->lwm_bx0 * y;
  A[0] = -Ay * s->lw_sin_a3 - Az * s->lw_sin_a2 * s->lw_cos_a3;
  A[1] = Ay * s->lw_cos_a3 - Az * s->lw_sin_a2 * s->lw_sin_a3;
  A[2] = Az * s->lw_cos_a2;
}

/* analytic d, m1, m2, m3, E at a cell centre (linear_wave_mhd.cpp:407-436 == :199-224) */
static void lwm_hydro_part(const orc_sim *s, int wf, double amp, double vflow, double x1, double x2, double x3, double u[5],
                           double *sn_out) {
  const double x = s->lw_cos_a2 * (x1 * s->lw_cos_a3 + x2 * s->lw_sin_a3) + x3 * s->lw_sin_a2;
  const double sn = sin(s->lw_kpar * x);
  u[ORC_IDN] = s->lw_d0 + amp * sn * s->lwm_rem[0][wf];
  const double mx = s->lw_d0 * vflow + amp * sn * s->lwm_rem[1][wf];
  const double my = amp * sn * s->lwm_rem[2][wf];
  const double mz = amp * sn * s->lwm_rem[3][wf];
  u[ORC_IM1] = mx * s->lw_cos_a2 * s->lw_cos_a3 - my * s->lw_sin_a3 - mz * s->lw_sin_a2 * s->lw_cos_a3;
  u[ORC_IM2] = mx * s->lw_cos_a2 * s->lw_sin_a3 + my * s->lw_cos_a3 - mz * s->lw_sin_a2 * s->lw_sin_a3;
  u[ORC_IM3] = mx * s->lw_sin_a2 + mz * s->lw_cos_a2;
  double e0 = s->lw_p0 / s->lw_gm1 + 0.5 * s->lw_d0 * s->lw_u0 * s->lw_u0 + amp * sn * s->lwm_rem[4][wf];
  e0 += 0.5 * (s->lwm_bx0 * s->lwm_bx0 + s->lwm_by0 * s->lwm_by0 + s->lwm_bz0 * s->lwm_bz0);
  u[ORC_IEN] = e0;
  *sn_out = sn;
}

/* ProblemGenerator (linear_wave_mhd.cpp:370-441): B = curl A by centred differences of the cell-centred
 * potential; returns the wave period lambda / |ev[wave_flag]| ("test = true" reinterprets tlim in periods) */
double orc_pgen_linear_wave_mhd(orc_sim *s, int wave_flag, double amp, double vflow) {
  lwm_setup(s, vflow);
  s->lwm_dby = amp * s->lwm_rem[5][wave_flag];
  s->lwm_dbz = amp * s->lwm_rem[6][wave_flag];
  const sb_t bb = sim_bounds(&s->g);
  const double dx1 = s->g.dx[0], dx2 = s->g.dx[1], dx3 = s->g.dx[2];
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          const double X1 = xc(s, x0, 0, i), X2 = xc(s, x0, 1, j), X3 = xc(s, x0, 2, k);
          double w[5], sn;
          lwm_hydro_part(s, wave_flag, amp, vflow, X1, X2, X3, w, &sn);
          for (int n = 0; n < 5; ++n) SAT(u, n, k, j, i) = w[n];
          double Ajp[3], Ajm[3], Akp[3], Akm[3], Aip[3], Aim[3];
          lwm_A(s, X1, xc(s, x0, 1, j + 1), X3, Ajp);
          lwm_A(s, X1, xc(s, x0, 1, j - 1), X3, Ajm);
          lwm_A(s, X1, X2, xc(s, x0, 2, k + 1), Akp);
          lwm_A(s, X1, X2, xc(s, x0, 2, k - 1), Akm);
          lwm_A(s, xc(s, x0, 0, i + 1), X2, X3, Aip);
          lwm_A(s, xc(s, x0, 0, i - 1), X2, X3, Aim);
          SAT(u, ORC_IB1, k, j, i) = (Ajp[2] - Ajm[2]) / dx2 / 2.0 - (Akp[1] - Akm[1]) / dx3 / 2.0;
          SAT(u, ORC_IB2, k, j, i) = (Akp[0] - Akm[0]) / dx3 / 2.0 - (Aip[2] - Aim[2]) / dx1 / 2.0;
          SAT(u, ORC_IB3, k, j, i) = (Aip[1] - Aim[1]) / dx1 / 2.0 - (Ajp[0] - Ajm[0]) / dx2 / 2.0;
        }
  }
  return s->lw_lambda / fabs(s->lwm_ev[wave_flag]);
}

/* the eigensystem the last orc_pgen_linear_wave_mhd used: ev[7], rem[7][7] row-major (tests) */
void orc_linear_wave_mhd_eigen(const orc_sim *s, double *ev7, double *rem49) {
  for (int w = 0; w < 7; ++w) ev7[w] = s->lwm_ev[w];
  for (int r = 0; r < 7; ++r)
    for (int w = 0; w < 7; ++w) rem49[r * 7 + w] = s->lwm_rem[r][w];
}

/* UserWorkAfterLoop (linear_wave_mhd.cpp:177-276): volume-weighted L1 and max errors of d, M1, M2, M3, E, B1, B2, B3
 * (psi excluded) against the analytic wave with the ANALYTIC field (not curl A); returns the RMS of the L1 errors */
double orc_linear_wave_mhd_errors(orc_sim *s, int wave_flag, double amp, double vflow, double *l1, double *maxerr) {
  const sb_t bb = sim_bounds(&s->g);
  const double cellvol = s->g.dx[0] * s->g.dx[1] * s->g.dx[2];
  const double ca2 = s->lw_cos_a2, sa2 = s->lw_sin_a2, ca3 = s->lw_cos_a3, sa3 = s->lw_sin_a3;
  for (int n = 0; n < 8; ++n) l1[n] = maxerr[n] = 0.0;
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          double a[8], sn;
          lwm_hydro_part(s, wave_flag, amp, vflow, xc(s, x0, 0, i), xc(s, x0, 1, j), xc(s, x0, 2, k), a, &sn);
          const double bx = s->lwm_bx0;
          const double by = s->lwm_by0 + amp * sn * s->lwm_rem[5][wave_flag];
          const double bz = s->lwm_bz0 + amp * sn * s->lwm_rem[6][wave_flag];
          a[ORC_IB1] = bx * ca2 * ca3 - by * sa3 - bz * sa2 * ca3;
          a[ORC_IB2] = bx * ca2 * sa3 + by * ca3 - bz * sa2 * sa3;
          a[ORC_IB3] = bx * sa2 + bz * ca2;
          for (int n = 0; n < 8; ++n) {
            const double e = fabs(a[n] - SAT(s->cons[b], n, k, j, i));
            l1[n] += e * cellvol;
            if (e > maxerr[n]) maxerr[n] = e;
          }
        }
  }
  const double vol = (s->p.xmax[0] - s->p.xmin[0]) * (s->p.xmax[1] - s->p.xmin[1]) * (s->p.xmax[2] - s->p.xmin[2]);
  double rms = 0.0;
  for (int n = 0; n < 8; ++n) {
    l1[n] = l1[n] / vol;
    rms += l1[n] * l1[n];
  }
  return sqrt(rms);
}

/* src/pgen/sod.cpp:17-51 */
void orc_pgen_sod(orc_sim *s, double rho_l, double pres_l, double u_l, double rho_r,
                  double pres_r, double u_r, double x_discont) {
  const sb_t bb = sim_bounds(&s->g);
  const double gamma = s->p.eos.gamma;
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          double *u = s->cons[b];
          if (xc(s, x0, 0, i) < x_discont) {
            SAT(u, ORC_IDN, k, j, i) = rho_l;
            SAT(u, ORC_IM1, k, j, i) = rho_l * u_l;
            SAT(u, ORC_IEN, k, j, i) = 0.5 * rho_l * u_l * u_l + pres_l / (gamma - 1.0);
          } else {
            SAT(u, ORC_IDN, k, j, i) = rho_r;
            SAT(u, ORC_IM1, k, j, i) = rho_r * u_r;
            SAT(u, ORC_IEN, k, j, i) = 0.5 * rho_r * u_r * u_r + pres_r / (gamma - 1.0);
          }
        }
  }
}

/* src/pgen/blast.cpp:124-207 (analytic sphere; the input-image variant is not restated) */
void orc_pgen_blast(orc_sim *s, double rout, double rin, double pa, double da, double prat, double drat,
                    double x0c, double y0c, double z0c) {
  const sb_t bb = sim_bounds(&s->g);
  const double gm1 = s->p.eos.gamma - 1.0;
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          double den = da, pres = pa;
          const double x = xc(s, x0, 0, i), y = xc(s, x0, 1, j), z = xc(s, x0, 2, k);
          const double rad = sqrt((x - x0c) * (x - x0c) + (y - y0c) * (y - y0c) + (z - z0c) * (z - z0c));
          if (rad < rout) {
            if (rad < rin) {
              den = drat * da;
            } else {
              const double f = (rad - rin) / (rout - rin);
              const double log_den = (1.0 - f) * log(drat * da) + f * log(da);
              den = exp(log_den);
            }
          }
          if (rad < rout) {
            if (rad < rin) {
              pres = prat * pa;
            } else {
              const double f = (rad - rin) / (rout - rin);
              const double log_pres = (1.0 - f) * log(prat * pa) + f * log(pa);
              pres = exp(log_pres);
            }
          }
          SAT(u, ORC_IDN, k, j, i) = den;
          SAT(u, ORC_IM1, k, j, i) = 0.0;
          SAT(u, ORC_IM2, k, j, i) = 0.0;
          SAT(u, ORC_IM3, k, j, i) = 0.0;
          SAT(u, ORC_IEN, k, j, i) = pres / gm1;
        }
  }
}

/* src/pgen/advection.cpp:64-115 (smooth density blob advected with a uniform flow); the caller
 * reinterprets tlim as box diagonals / |v| like InitUserMeshData (:34-59) */
void orc_pgen_advection(orc_sim *s, double vx, double vy, double vz, double rho_ratio, double rho_radius,
                        double rho_fraction_edge, double rho0, double p0) {
  const sb_t bb = sim_bounds(&s->g);
  const double gm1 = s->p.eos.gamma - 1.0;
  const double sigmasq = -rho_radius * rho_radius / 2 / log(rho_fraction_edge);
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          const double x = xc(s, x0, 0, i), y = xc(s, x0, 1, j), z = xc(s, x0, 2, k);
          double rho = rho0;
          const double rsq = x * x + y * y + z * z;
          if (rsq < rho_radius * rho_radius) rho += rho0 * rho_ratio * exp(-rsq / 2 / sigmasq);
          const double mx = rho * vx, my = rho * vy, mz = rho * vz;
          SAT(u, ORC_IDN, k, j, i) = rho;
          SAT(u, ORC_IM1, k, j, i) = mx;
          SAT(u, ORC_IM2, k, j, i) = my;
          SAT(u, ORC_IM3, k, j, i) = mz;
          SAT(u, ORC_IEN, k, j, i) = p0 / gm1 + 0.5 * (mx * mx + my * my + mz * mz) / rho;
        }
  }
}

/* src/pgen/lw_implode.cpp:24-75 (Liska-Wendroff implosion; hydro only).  The diagonal offset y0 is
 * adjusted per meshblock exactly as the reference's loop over the block's own rows does. */
void orc_pgen_lw_implode(orc_sim *s, double d_in, double p_in, double d_out, double p_out) {
  const sb_t bb = sim_bounds(&s->g);
  const double gm1 = s->p.eos.gamma - 1.0;
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    double *u = s->cons[b];
    double y0 = 0.5 * (s->p.xmax[1] + s->p.xmin[1]);
    for (int j = bb.js; j <= bb.je; ++j) {
      if (xc(s, x0, 1, j) > y0) {
        const int ng = (s->g.nx[1] > 1) ? s->g.ng : 0;
        const double xf = s->p.xmin[1] + (x0[1] + (double)(j - ng)) * s->g.dx[1]; /* lower x2 face of cell j */
        y0 = xf + 0.5 * s->g.dx[1];
        break;
      }
    }
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          SAT(u, ORC_IM1, k, j, i) = 0.0;
          SAT(u, ORC_IM2, k, j, i) = 0.0;
          SAT(u, ORC_IM3, k, j, i) = 0.0;
          if (xc(s, x0, 1, j) > (y0 - xc(s, x0, 0, i))) {
            SAT(u, ORC_IDN, k, j, i) = d_out;
            SAT(u, ORC_IEN, k, j, i) = p_out / gm1;
          } else {
            SAT(u, ORC_IDN, k, j, i) = d_in;
            SAT(u, ORC_IEN, k, j, i) = p_in / gm1;
          }
        }
  }
}

/* ---- Kelvin-Helmholtz shear layers (src/pgen/kh.cpp:43-239), iprob 2..5 ----------------------------- */
void orc_pgen_kh(orc_sim *s, int iprob, double vflow, double amp, double drho_rho0, double vboost, double a5,
                 double sigma5, double drat5) {
  const sb_t bb = sim_bounds(&s->g);
  const double gm1 = s->p.eos.gamma - 1.0;
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          const double x = xc(s, x0, 0, i), y = xc(s, x0, 1, j);
          double d = 1.0, m1 = 0.0, m2 = 0.0, pr = 1.0;
          if (iprob == 2) { /* single tanh shear layer, Frank et al. 1996 */
            const double a = 0.02, sigma = 0.2;
            m1 = vflow * tanh(y / a);
            m2 = amp * cos(2.0 * M_PI * x) * exp(-(y * y) / (sigma * sigma));
          } else if (iprob == 3) { /* two resolved layers at y = +-0.5, Beckwith & Stone 2011 */
            const double a = 0.01, sigma = 0.1;
            d = 0.505 + 0.495 * tanh((fabs(y) - 0.5) / a);
            m1 = vflow * tanh((fabs(y) - 0.5) / a);
            m2 = amp * vflow * sin(2.0 * M_PI * x) * exp(-((fabs(y) - 0.5) * (fabs(y) - 0.5)) / (sigma * sigma));
            if (y < 0.0) m2 *= -1.0;
            m1 *= d;
            m2 *= d;
          } else if (iprob == 4) { /* Lecoanet et al. 2016 in coordinates centred on the origin */
            const double a = 0.05, sigma = 0.2, z1 = -0.5, z2 = 0.5;
            pr = 10.0;
            d = 1.0 + 0.5 * drho_rho0 * (tanh((y - z1) / a) - tanh((y - z2) / a));
            const double v1 = vflow * (tanh((y - z1) / a) - tanh((y - z2) / a) - 1.0) + vboost;
            m1 = v1 * d;
            /* sine averaged with minus its half-period shift: exact shift symmetry in floating point */
            double ave_sine = sin(2.0 * M_PI * x);
            if (x > 0.0) ave_sine -= sin(2.0 * M_PI * (-0.5 + x));
            else ave_sine -= sin(2.0 * M_PI * (0.5 + x));
            ave_sine /= 2.0;
            const double v2 = -amp * ave_sine *
                              (exp(-((y - z1) * (y - z1)) / (sigma * sigma)) + exp(-((y - z2) * (y - z2)) / (sigma * sigma)));
            m2 = v2 * d;
          } else { /* iprob 5: stream of density drat in |y| < 1/4, m = 2 perturbation (the AMR test) */
            const double w = (tanh((fabs(y) - 0.25) / a5) + 1.0) * 0.5;
            pr = 2.5;
            d = w + (1.0 - w) * drat5;
            m1 = d * vflow * (w - 0.5);
            m2 = d * amp * cos(2.0 * 2.0 * M_PI * x) * exp(-((fabs(y) - 0.25) * (fabs(y) - 0.25)) / (sigma5 * sigma5));
          }
          SAT(u, ORC_IDN, k, j, i) = d;
          SAT(u, ORC_IM1, k, j, i) = m1;
          SAT(u, ORC_IM2, k, j, i) = m2;
          SAT(u, ORC_IM3, k, j, i) = 0.0;
          SAT(u, ORC_IEN, k, j, i) = pr / gm1 + 0.5 * (m1 * m1 + m2 * m2) / d;
        }
  }
}

/* ---- advected field loop (src/pgen/field_loop.cpp:105-316) ----------------------------------------- */
typedef struct {
  int iprob;
  double rad, amp, cos_a2, sin_a2, lambda;
} floop_t;

static void floop_A(const floop_t *f, double x1, double x2, double x3, double A[3]) {
  A[0] = A[1] = A[2] = 0.0;
  const double rad = f->rad, amp = f->amp;
  if (f->iprob == 1) {
    if ((x1 * x1 + x2 * x2) < rad * rad) A[2] = amp * (rad - sqrt(x1 * x1 + x2 * x2));
  } else if (f->iprob == 2) {
    if ((x2 * x2 + x3 * x3) < rad * rad) A[0] = amp * (rad - sqrt(x2 * x2 + x3 * x3));
  } else if (f->iprob == 3) {
    if ((x1 * x1 + x3 * x3) < rad * rad) A[1] = amp * (rad - sqrt(x1 * x1 + x3 * x3));
  } else if (f->iprob == 4) {
    double x = x1 * f->cos_a2 + x3 * f->sin_a2;
    const double y = x2;
    while (x > 0.5 * f->lambda) x -= f->lambda;
    while (x < -0.5 * f->lambda) x += f->lambda;
    if ((x * x + y * y) < rad * rad) {
      A[0] = amp * (rad - sqrt(x * x + y * y)) * (-f->sin_a2);
      A[2] = amp * (rad - sqrt(x * x + y * y)) * (f->cos_a2);
    }
  } else if (f->iprob == 5) {
    if ((x1 * x1 + x2 * x2 + x3 * x3) < rad * rad) {
      A[1] = amp * (rad - sqrt(x1 * x1 + x2 * x2 + x3 * x3));
      A[2] = amp * (rad - sqrt(x1 * x1 + x2 * x2 + x3 * x3));
    }
  }
}

void orc_pgen_field_loop(orc_sim *s, double rad, double amp, double vflow, double drat, int iprob) {
  const sb_t bb = sim_bounds(&s->g);
  const double gm1 = s->p.eos.gamma - 1.0;
  const double x1size = s->p.xmax[0] - s->p.xmin[0], x2size = s->p.xmax[1] - s->p.xmin[1];
  const int two_d = s->p.nx[2] <= 1;
  const double x3size = two_d ? 0 : s->p.xmax[2] - s->p.xmin[2];
  floop_t f = {iprob, rad, amp, 0.0, 0.0, 0.0};
  if (iprob == 4) {
    if (x1size == x3size) {
      f.cos_a2 = f.sin_a2 = sqrt(0.5);
    } else {
      const double ang_2 = atan(x1size / x3size);
      f.sin_a2 = sin(ang_2);
      f.cos_a2 = cos(ang_2);
    }
    f.lambda = (f.cos_a2 >= f.sin_a2) ? x1size * f.cos_a2 : x3size * f.sin_a2;
  }
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          const double X1 = xc(s, x0, 0, i), X2 = xc(s, x0, 1, j), X3 = xc(s, x0, 2, k);
          double Ajp[3], Ajm[3], Aip[3], Aim[3], Akp[3] = {0, 0, 0}, Akm[3] = {0, 0, 0};
          floop_A(&f, X1, xc(s, x0, 1, j + 1), X3, Ajp);
          floop_A(&f, X1, xc(s, x0, 1, j - 1), X3, Ajm);
          floop_A(&f, xc(s, x0, 0, i + 1), X2, X3, Aip);
          floop_A(&f, xc(s, x0, 0, i - 1), X2, X3, Aim);
          if (!two_d) {
            floop_A(&f, X1, X2, xc(s, x0, 2, k + 1), Akp);
            floop_A(&f, X1, X2, xc(s, x0, 2, k - 1), Akm);
          }
          const double dx1 = s->g.dx[0], dx2 = s->g.dx[1], dx3 = s->g.dx[2];
          double den = 1.0;
          if ((X1 * X1 + X2 * X2 + X3 * X3) < rad * rad) den = drat;
          SAT(u, ORC_IDN, k, j, i) = den;
          SAT(u, ORC_IM1, k, j, i) = den * vflow * x1size;
          SAT(u, ORC_IM2, k, j, i) = den * vflow * x2size;
          SAT(u, ORC_IM3, k, j, i) = den * vflow * x3size;
          const double aydz = two_d ? 0.0 : (Akp[1] - Akm[1]) / dx3 / 2.0;
          const double axdz = two_d ? 0.0 : (Akp[0] - Akm[0]) / dx3 / 2.0;
          const double b1 = (Ajp[2] - Ajm[2]) / dx2 / 2.0 - aydz;
          const double b2 = axdz - (Aip[2] - Aim[2]) / dx1 / 2.0;
          const double b3 = (Aip[1] - Aim[1]) / dx1 / 2.0 - (Ajp[0] - Ajm[0]) / dx2 / 2.0;
          SAT(u, ORC_IB1, k, j, i) = b1;
          SAT(u, ORC_IB2, k, j, i) = b2;
          SAT(u, ORC_IB3, k, j, i) = b3;
          const double m1 = SAT(u, ORC_IM1, k, j, i), m2 = SAT(u, ORC_IM2, k, j, i), m3 = SAT(u, ORC_IM3, k, j, i);
          SAT(u, ORC_IEN, k, j, i) = 1.0 / gm1 + 0.5 * (b1 * b1 + b2 * b2 + b3 * b3) + 0.5 * (m1 * m1 + m2 * m2 + m3 * m3) / den;
        }
  }
}

/* field_loop::RelDivBHst (:60-95) summed over all blocks */
double orc_user_reldivb(orc_sim *s, double B0) {
  const sb_t bb = sim_bounds(&s->g);
  const double dx1 = s->g.dx[0], dx2 = s->g.dx[1], dx3 = s->g.dx[2];
  const double vol = dx1 * dx2 * dx3;
  const int three_d = s->p.nx[2] > 1;
  double sum = 0.0;
  for (int b = 0; b < s->nblocks; ++b) {
    const double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          double divb = (SAT(u, ORC_IB1, k, j, i + 1) - SAT(u, ORC_IB1, k, j, i - 1)) / dx1 +
                        (SAT(u, ORC_IB2, k, j + 1, i) - SAT(u, ORC_IB2, k, j - 1, i)) / dx2;
          if (three_d) divb += (SAT(u, ORC_IB3, k + 1, j, i) - SAT(u, ORC_IB3, k - 1, j, i)) / dx3;
          sum += 0.5 * (sqrt(dx1 * dx1 + dx2 * dx2 + dx3 * dx3)) * fabs(divb) / B0 * vol;
        }
  }
  return sum;
}

/* ---- circularly polarised Alfven wave (src/pgen/cpaw.cpp) -------------------------------------- */
/* vector potential in a gauge with Ax = 0 (cpaw.cpp:310-344) */
static void cpaw_A(const orc_sim *s, double x1, double x2, double x3, double A[3]) {
  const double x = x1 * s->cpaw.cos_a2 * s->cpaw.cos_a3 + x2 * s->cpaw.cos_a2 * s->cpaw.sin_a3 + x3 * s->cpaw.sin_a2;
  const double y = -x1 * s->cpaw.sin_a3 + x2 * s->cpaw.cos_a3;
  const double Ay = s->cpaw.fac * (s->cpaw.b_perp / s->cpaw.k_par) * sin(s->cpaw.k_par * (x));
  const double Az = (s->cpaw.b_perp / s->cpaw.k_par) * cos(s->cpaw.k_par * (x)) + s->cpaw.b_par * y;
  A[0] = -Ay * s->cpaw.sin_a3 - Az * s->cpaw.sin_a2 * s->cpaw.cos_a3;
  A[1] = Ay * s->cpaw.cos_a3 - Az * s->cpaw.sin_a2 * s->cpaw.sin_a3;
  A[2] = Az * s->cpaw.cos_a2;
}

/* InitUserMeshData + ProblemGenerator (cpaw.cpp:58-125, 227-303); 3-D only; returns lambda.
 * ang_2 / ang_3 = -999.9 select the grid-diagonal wavevector like the reference. */
double orc_pgen_cpaw(orc_sim *s, double b_par, double b_perp, double pres, double v_par, int dir, double ang_2,
                     double ang_3) {
  const sb_t bb = sim_bounds(&s->g);
  s->cpaw.b_par = b_par;
  s->cpaw.b_perp = b_perp;
  s->cpaw.v_par = v_par;
  s->cpaw.gm1 = s->p.eos.gamma - 1.0;
  s->cpaw.pres = pres;
  s->cpaw.den = 1.0;
  const double x1size = s->p.xmax[0] - s->p.xmin[0], x2size = s->p.xmax[1] - s->p.xmin[1],
               x3size = s->p.xmax[2] - s->p.xmin[2];
  if (ang_3 == -999.9) ang_3 = atan(x1size / x2size);
  s->cpaw.sin_a3 = sin(ang_3);
  s->cpaw.cos_a3 = cos(ang_3);
  if (ang_2 == -999.9) ang_2 = atan(0.5 * (x1size * s->cpaw.cos_a3 + x2size * s->cpaw.sin_a3) / x3size);
  s->cpaw.sin_a2 = sin(ang_2);
  s->cpaw.cos_a2 = cos(ang_2);
  const double x1 = x1size * s->cpaw.cos_a2 * s->cpaw.cos_a3, x2 = x2size * s->cpaw.cos_a2 * s->cpaw.sin_a3,
               x3 = x3size * s->cpaw.sin_a2;
  double lambda = x1;
  if (s->p.nx[1] > 1 && ang_3 != 0.0) lambda = fmin(lambda, x2);
  if (s->p.nx[2] > 1 && ang_2 != 0.0) lambda = fmin(lambda, x3);
  s->cpaw.lambda = lambda;
  s->cpaw.k_par = 2.0 * (M_PI) / lambda;
  s->cpaw.v_perp = b_perp / sqrt(s->cpaw.den);
  s->cpaw.fac = (dir == 1) ? 1.0 : -1.0;
  const double den = s->cpaw.den, fac = s->cpaw.fac, v_perp = s->cpaw.v_perp;
  const double sa2 = s->cpaw.sin_a2, ca2 = s->cpaw.cos_a2, sa3 = s->cpaw.sin_a3, ca3 = s->cpaw.cos_a3;
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          const double X1 = xc(s, x0, 0, i), X2 = xc(s, x0, 1, j), X3 = xc(s, x0, 2, k);
          const double x = ca2 * (X1 * ca3 + X2 * sa3) + X3 * sa2;
          const double sn = sin(s->cpaw.k_par * x);
          const double cs = fac * cos(s->cpaw.k_par * x);
          SAT(u, ORC_IDN, k, j, i) = den;
          const double mx = den * v_par, my = -fac * den * v_perp * sn, mz = -fac * den * v_perp * cs;
          SAT(u, ORC_IM1, k, j, i) = mx * ca2 * ca3 - my * sa3 - mz * sa2 * ca3;
          SAT(u, ORC_IM2, k, j, i) = mx * ca2 * sa3 + my * ca3 - mz * sa2 * sa3;
          SAT(u, ORC_IM3, k, j, i) = mx * sa2 + mz * ca2;
          /* B = curl A by centred differences of the cell-centred potential */
          double Ajp[3], Ajm[3], Akp[3], Akm[3], Aip[3], Aim[3];
          cpaw_A(s, X1, xc(s, x0, 1, j + 1), X3, Ajp);
          cpaw_A(s, X1, xc(s, x0, 1, j - 1), X3, Ajm);
          cpaw_A(s, X1, X2, xc(s, x0, 2, k + 1), Akp);
          cpaw_A(s, X1, X2, xc(s, x0, 2, k - 1), Akm);
          cpaw_A(s, xc(s, x0, 0, i + 1), X2, X3, Aip);
          cpaw_A(s, xc(s, x0, 0, i - 1), X2, X3, Aim);
          const double dx1 = s->g.dx[0], dx2 = s->g.dx[1], dx3 = s->g.dx[2];
          const double b1 = (Ajp[2] - Ajm[2]) / dx2 / 2.0 - (Akp[1] - Akm[1]) / dx3 / 2.0;
          const double b2 = (Akp[0] - Akm[0]) / dx3 / 2.0 - (Aip[2] - Aim[2]) / dx1 / 2.0;
          const double b3 = (Aip[1] - Aim[1]) / dx1 / 2.0 - (Ajp[0] - Ajm[0]) / dx2 / 2.0;
          SAT(u, ORC_IB1, k, j, i) = b1;
          SAT(u, ORC_IB2, k, j, i) = b2;
          SAT(u, ORC_IB3, k, j, i) = b3;
          const double m1 = SAT(u, ORC_IM1, k, j, i), m2 = SAT(u, ORC_IM2, k, j, i), m3 = SAT(u, ORC_IM3, k, j, i);
          SAT(u, ORC_IEN, k, j, i) = s->cpaw.pres / s->cpaw.gm1 + 0.5 * (b1 * b1 + b2 * b2 + b3 * b3) +
                                     (0.5 / den) * (m1 * m1 + m2 * m2 + m3 * m3);
        }
  }
  return lambda;
}

/* UserWorkAfterLoop (cpaw.cpp:127-221): L1 errors against the initial state, err8 in the order
 * d, M1, M2, M3, E, B1, B2, B3; returns the RMS */
double orc_cpaw_errors(orc_sim *s, double *err8) {
  const sb_t bb = sim_bounds(&s->g);
  const double den = s->cpaw.den, fac = s->cpaw.fac, v_perp = s->cpaw.v_perp, v_par = s->cpaw.v_par;
  const double sa2 = s->cpaw.sin_a2, ca2 = s->cpaw.cos_a2, sa3 = s->cpaw.sin_a3, ca3 = s->cpaw.cos_a3;
  for (int n = 0; n < 8; ++n) err8[n] = 0.0;
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    const double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          const double x = ca2 * (xc(s, x0, 0, i) * ca3 + xc(s, x0, 1, j) * sa3) + xc(s, x0, 2, k) * sa2;
          const double sn = sin(s->cpaw.k_par * x);
          const double cs = fac * cos(s->cpaw.k_par * x);
          err8[ORC_IDN] += fabs(den - SAT(u, ORC_IDN, k, j, i));
          const double mx = den * v_par, my = -fac * den * v_perp * sn, mz = -fac * den * v_perp * cs;
          const double m1 = mx * ca2 * ca3 - my * sa3 - mz * sa2 * ca3;
          const double m2 = mx * ca2 * sa3 + my * ca3 - mz * sa2 * sa3;
          const double m3 = mx * sa2 + mz * ca2;
          err8[ORC_IM1] += fabs(m1 - SAT(u, ORC_IM1, k, j, i));
          err8[ORC_IM2] += fabs(m2 - SAT(u, ORC_IM2, k, j, i));
          err8[ORC_IM3] += fabs(m3 - SAT(u, ORC_IM3, k, j, i));
          const double bx = s->cpaw.b_par, by = s->cpaw.b_perp * sn, bz = s->cpaw.b_perp * cs;
          const double b1 = bx * ca2 * ca3 - by * sa3 - bz * sa2 * ca3;
          const double b2 = bx * ca2 * sa3 + by * ca3 - bz * sa2 * sa3;
          const double b3 = bx * sa2 + bz * ca2;
          err8[ORC_IB1] += fabs(b1 - SAT(u, ORC_IB1, k, j, i));
          err8[ORC_IB2] += fabs(b2 - SAT(u, ORC_IB2, k, j, i));
          err8[ORC_IB3] += fabs(b3 - SAT(u, ORC_IB3, k, j, i));
          const double e0 = s->cpaw.pres / s->cpaw.gm1 + 0.5 * (m1 * m1 + m2 * m2 + m3 * m3) / den +
                            0.5 * (b1 * b1 + b2 * b2 + b3 * b3);
          err8[ORC_IEN] += fabs(e0 - SAT(u, ORC_IEN, k, j, i));
        }
  }
  const double ncells = (double)s->p.nx[0] * s->p.nx[1] * s->p.nx[2];
  double rms = 0.0;
  for (int n = 0; n < 8; ++n) {
    err8[n] = err8[n] / ncells;
    rms += err8[n] * err8[n];
  }
  return sqrt(rms);
}

/* src/pgen/orszag_tang.cpp:25-63 */
void orc_pgen_orszag_tang(orc_sim *s) {
  const sb_t bb = sim_bounds(&s->g);
  const double gm1 = s->p.eos.gamma - 1.0;
  const double B0 = 1.0 / sqrt(4.0 * M_PI);
  const double d0 = 25.0 / (36.0 * M_PI);
  const double v0 = 1.0;
  const double p0 = 5.0 / (12.0 * M_PI);
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          const double x1 = xc(s, x0, 0, i), x2 = xc(s, x0, 1, j);
          SAT(u, ORC_IDN, k, j, i) = d0;
          SAT(u, ORC_IM1, k, j, i) = d0 * v0 * sin(2.0 * M_PI * x2);
          SAT(u, ORC_IM2, k, j, i) = -d0 * v0 * sin(2.0 * M_PI * x1);
          SAT(u, ORC_IM3, k, j, i) = 0.0;
          SAT(u, ORC_IB1, k, j, i) = B0 * sin(2.0 * M_PI * x2);
          SAT(u, ORC_IB2, k, j, i) = B0 * sin(4.0 * M_PI * x1);
          SAT(u, ORC_IB3, k, j, i) = 0.0;
          const double b1 = SAT(u, ORC_IB1, k, j, i), b2 = SAT(u, ORC_IB2, k, j, i),
                       b3 = SAT(u, ORC_IB3, k, j, i);
          const double m1 = SAT(u, ORC_IM1, k, j, i), m2 = SAT(u, ORC_IM2, k, j, i),
                       m3 = SAT(u, ORC_IM3, k, j, i);
          SAT(u, ORC_IEN, k, j, i) =
              p0 / gm1 + 0.5 * (b1 * b1 + b2 * b2 + b3 * b3 +
                                (m1 * m1 + m2 * m2 + m3 * m3) / SAT(u, ORC_IDN, k, j, i));
        }
  }
}

/* Analytic, seedless smooth state for the synthetic kernel benchmark:
 * rho = 1+0.2 sin, p = 1+0.1 cos, |v| <= 0.3, |B| <= 0.5, psi = 0.01 sin; periodic in the
 * mesh.  Arguments are fractions of the domain so that the field is exactly periodic. */
void orc_pgen_synthetic(orc_sim *s) {
  const sb_t bb = sim_bounds(&s->g);
  const double gm1 = s->p.eos.gamma - 1.0;
  const int mhd = (s->p.fluid == ORC_FLUID_GLMMHD);
  const double tp = 2.0 * M_PI;
  for (int b = 0; b < s->nblocks; ++b) {
    double x0[3];
    orc_sim_block_origin(s, b, x0);
    double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          const double fx = (xc(s, x0, 0, i) - s->p.xmin[0]) / (s->p.xmax[0] - s->p.xmin[0]);
          const double fy = (xc(s, x0, 1, j) - s->p.xmin[1]) / (s->p.xmax[1] - s->p.xmin[1]);
          const double fz = (xc(s, x0, 2, k) - s->p.xmin[2]) / (s->p.xmax[2] - s->p.xmin[2]);
          const double rho = 1.0 + 0.2 * sin(tp * (fx + fy + fz));
          const double p = 1.0 + 0.1 * cos(tp * (fx - fy + 2.0 * fz));
          const double v1 = 0.17 * sin(tp * (fy + fz));
          const double v2 = 0.17 * cos(tp * (fx - fz));
          const double v3 = 0.17 * sin(tp * (2.0 * fx + fy));
          double b1 = 0.0, b2 = 0.0, b3 = 0.0, psi = 0.0;
          if (mhd) {
            b1 = 0.28 * cos(tp * (fy - fz));
            b2 = 0.28 * sin(tp * (fx + 2.0 * fz));
            b3 = 0.28 * cos(tp * (fx + fy));
            psi = 0.01 * sin(tp * (fx + fy - fz));
          }
          SAT(u, ORC_IDN, k, j, i) = rho;
          SAT(u, ORC_IM1, k, j, i) = rho * v1;
          SAT(u, ORC_IM2, k, j, i) = rho * v2;
          SAT(u, ORC_IM3, k, j, i) = rho * v3;
          double e = p / gm1 + 0.5 * rho * (v1 * v1 + v2 * v2 + v3 * v3);
          if (mhd) {
            e += 0.5 * (b1 * b1 + b2 * b2 + b3 * b3);
            SAT(u, ORC_IB1, k, j, i) = b1;
            SAT(u, ORC_IB2, k, j, i) = b2;
            SAT(u, ORC_IB3, k, j, i) = b3;
            SAT(u, ORC_IPS, k, j, i) = psi;
          }
          SAT(u, ORC_IEN, k, j, i) = e;
          for (int n = s->g.nhydro; n < s->g.nvar; ++n)
            SAT(u, n, k, j, i) = rho * (0.5 + 0.25 * sin(tp * (fx + (n - s->g.nhydro + 1) * fy)));
        }
  }
}


/* src/pgen/turbulence.cpp:217-370 (uniform / no-net-flux / sin(z) field, b_config 0/1/2) */
void orc_pgen_turbulence(orc_sim *s, double rho0, double p0, double b0, int b_config, int num_modes,
                         const double *k_vec, double k_peak, double sol_weight, double t_corr,
                         double accel_rms, uint32_t rseed) {
  const sb_t bb = sim_bounds(&s->g);
  const int mhd = (s->p.fluid == ORC_FLUID_GLMMHD);
  const double gm1 = s->p.eos.gamma - 1.0;
  const double x3min = s->p.xmin[2];
  const double Lx = s->p.xmax[0] - s->p.xmin[0], Ly = s->p.xmax[1] - s->p.xmin[1],
               Lz = s->p.xmax[2] - s->p.xmin[2];
  const double kz = 2.0 * M_PI / Lz;
  const double vol = s->g.dx[0] * s->g.dx[1] * s->g.dx[2];
  double b_norm = 0.0;
  if (mhd) {
    double mag_en_sum = 0.0;
    for (int b = 0; b < s->nblocks; ++b) {
      double x0[3];
      orc_sim_block_origin(s, b, x0);
      double *u = s->cons[b];
      for (int k = bb.ks; k <= bb.ke; ++k)
        for (int j = bb.js; j <= bb.je; ++j)
          for (int i = bb.is; i <= bb.ie; ++i) {
            double b1 = 0.0;
            if (b_config == 0) b1 = b0;
            if (b_config == 1) b1 = (xc(s, x0, 2, k) < x3min + Lz / 2.0) ? b0 : -b0;
            if (b_config == 2) b1 = b0 / sqrt(0.5) * sin(kz * xc(s, x0, 2, k));
            /* the vector-potential terms vanish for these configurations (a == 0) */
            SAT(u, ORC_IB1, k, j, i) = b1;
            SAT(u, ORC_IB2, k, j, i) = 0.0;
            SAT(u, ORC_IB3, k, j, i) = 0.0;
            mag_en_sum += 0.5 * (b1 * b1 + 0.0 + 0.0) * vol;
          }
    }
    b_norm = sqrt(mag_en_sum / (Lx * Ly * Lz) / (0.5 * b0 * b0));
  }
  for (int b = 0; b < s->nblocks; ++b) {
    double *u = s->cons[b];
    for (int k = bb.ks; k <= bb.ke; ++k)
      for (int j = bb.js; j <= bb.je; ++j)
        for (int i = bb.is; i <= bb.ie; ++i) {
          SAT(u, ORC_IDN, k, j, i) = rho0;
          SAT(u, ORC_IM1, k, j, i) = rho0 * 0.0;
          SAT(u, ORC_IM2, k, j, i) = rho0 * 0.0;
          SAT(u, ORC_IM3, k, j, i) = rho0 * 0.0;
          SAT(u, ORC_IEN, k, j, i) = p0 / gm1 + 0.5 * rho0 * (0.0 + 0.0 + 0.0);
          if (mhd) {
            SAT(u, ORC_IB1, k, j, i) /= b_norm;
            SAT(u, ORC_IB2, k, j, i) /= b_norm;
            SAT(u, ORC_IB3, k, j, i) /= b_norm;
            const double b1 = SAT(u, ORC_IB1, k, j, i), b2 = SAT(u, ORC_IB2, k, j, i),
                         b3 = SAT(u, ORC_IB3, k, j, i);
            SAT(u, ORC_IEN, k, j, i) += 0.5 * (b1 * b1 + b2 * b2 + b3 * b3);
          }
        }
  }
  /* enrol the driver: FewModesFT + per-block phase tables (SetPhases) + acc field */
  s->fmft = orc_fmft_create(num_modes, k_vec, k_peak, sol_weight, t_corr, rseed);
  s->accel_rms = accel_rms;
  s->acc = (double **)calloc(s->nblocks, sizeof(double *));
  s->ph_i = (double **)calloc(s->nblocks, sizeof(double *));
  s->ph_j = (double **)calloc(s->nblocks, sizeof(double *));
  s->ph_k = (double **)calloc(s->nblocks, sizeof(double *));
  for (int b = 0; b < s->nblocks; ++b) {
    int bc[3];
    block_coords(s, b, bc);
    s->acc[b] = (double *)calloc((size_t)3 * bb.sn, sizeof(double));
    s->ph_i[b] = (double *)malloc(sizeof(double) * s->g.nx[0] * num_modes * 2);
    s->ph_j[b] = (double *)malloc(sizeof(double) * s->g.nx[1] * num_modes * 2);
    s->ph_k[b] = (double *)malloc(sizeof(double) * s->g.nx[2] * num_modes * 2);
    orc_fmft_phases(s->fmft, 0, s->g.nx[0], bc[0] * s->p.mb[0], s->p.nx[0], s->ph_i[b]);
    orc_fmft_phases(s->fmft, 1, s->g.nx[1], bc[1] * s->p.mb[1], s->p.nx[1], s->ph_j[b]);
    orc_fmft_phases(s->fmft, 2, s->g.nx[2], bc[2] * s->p.mb[2], s->p.nx[2], s->ph_k[b]);
  }
}

void orc_sim_turb_history(orc_sim *s, double *out3) {
  out3[0] = out3[1] = out3[2] = 0.0;
  for (int b = 0; b < s->nblocks; ++b)
    orc_turb_history(&s->g, s->p.fluid, s->p.eos.gamma, s->prim[b], out3);
}

const double *orc_sim_var_hat(const orc_sim *s) { return s->fmft ? s->fmft->var_hat : NULL; }
double *orc_sim_acc(orc_sim *s, int b) { return s->fmft ? s->acc[b] : NULL; }
