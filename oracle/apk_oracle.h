/*
 * apk_oracle.h -- CPU ORACLE for the AthenaPK flux-divergence hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (athenapk_amd/) never
 * links, imports or falls back to anything in this directory.
 *
 * It is a plain-C restatement (no Kokkos, no Parthenon) of the reference algorithm; every
 * function cites the reference file:line (relative to /root/reference) it follows, and
 * keeps the reference's floating-point operation ORDER so that, built with
 * -ffp-contract=off, it is the canonical IEEE result the HIP kernels are compared to.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *  - pointwise known answers recorded from the reference's own headers in SURVEY.md 8(c)
 *    (PLM/PPM/WENOZ/LimO3 probe stencils, one HLLD flux vector) -> tests/golden/survey_probes.json
 *  - the reference's own regression bounds (hydro linear wave VL2+PLM+HLLE 128x64x64
 *    RMS-L1 <= 1.547584e-08, tst/regression/test_suites/convergence/convergence.py:163;
 *    GLM-MHD RK3+WENOZ+HLLE 256x128x128 <= 6.14e-12, mhd_convergence.py:167)
 *  - Parthenon-owned pieces (flux divergence form, integrator coefficients, dt control,
 *    ghost-fill order) are un-vendored: PARITY UNPINNED except through those bounds.
 */
#ifndef APK_ORACLE_H_
#define APK_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* variable ordering contract: src/main.hpp:19-33 */
enum { ORC_IDN = 0, ORC_IM1 = 1, ORC_IM2 = 2, ORC_IM3 = 3, ORC_IEN = 4, ORC_NHYDRO = 5,
       ORC_IB1 = 5, ORC_IB2 = 6, ORC_IB3 = 7, ORC_IPS = 8, ORC_NGLMMHD = 9 };
enum { ORC_IV1 = 1, ORC_IV2 = 2, ORC_IV3 = 3, ORC_IPR = 4 };

/* option enums, numeric values mirror `enum class` order in src/main.hpp:35-38 */
enum { ORC_RS_UNDEF = 0, ORC_RS_NONE = 1, ORC_RS_HLLE = 2, ORC_RS_LLF = 3, ORC_RS_HLLC = 4,
       ORC_RS_HLLD = 5 };
enum { ORC_RC_UNDEF = 0, ORC_RC_DC = 1, ORC_RC_PLM = 2, ORC_RC_PPM = 3, ORC_RC_WENOZ = 4,
       ORC_RC_WENO3 = 5, ORC_RC_LIMO3 = 6 };
enum { ORC_INT_UNDEF = 0, ORC_INT_RK1 = 1, ORC_INT_RK2 = 2, ORC_INT_VL2 = 3, ORC_INT_RK3 = 4 };
enum { ORC_FLUID_UNDEF = 0, ORC_FLUID_EULER = 1, ORC_FLUID_GLMMHD = 2 };
enum { ORC_BC_PERIODIC = 0, ORC_BC_OUTFLOW = 1, ORC_BC_REFLECT = 2 };

#define ORC_TINY_NUMBER 1.0e-20 /* Parthenon macro (un-vendored), SURVEY.md App. A.7 */

typedef struct {
  double gamma;
  double pfloor, dfloor, efloor; /* <=0 : disabled (src/hydro/hydro.cpp:507-510) */
  double vceil, eceil;           /* +inf : disabled (src/hydro/hydro.cpp:525-529) */
} orc_eos;

/* geometry of one meshblock: interior nx[], ghosts ng in active dims, cell widths */
typedef struct {
  int nx[3];    /* interior cells nx1,nx2,nx3 (1 = collapsed) */
  int ng;       /* ghost layers in active dims */
  int nvar;     /* nhydro + nscalars */
  int nhydro;   /* 5 or 9 */
  double dx[3]; /* uniform Cartesian cell widths */
} orc_geom;

/* ---- derived sizes ---- */
int orc_ndim(const orc_geom *g);
int orc_ni(const orc_geom *g); /* total cells incl. ghosts in i */
int orc_nj(const orc_geom *g);
int orc_nk(const orc_geom *g);
long orc_ncell(const orc_geom *g);

/* ---- pointwise reconstruction (src/recon/...) ---- */
void orc_plm(double qm1, double q0, double qp1, double *ql_ip1, double *qr_i);
void orc_ppm(double qm2, double qm1, double q0, double qp1, double qp2, double *ql_ip1,
             double *qr_i);
void orc_wenoz(double qm2, double qm1, double q0, double qp1, double qp2, double *ql_ip1,
               double *qr_i);
void orc_weno3(double qm1, double q0, double qp1, double dx2, double *ql_ip1, double *qr_i);
void orc_limo3(double qm1, double q0, double qp1, double dx, int ensure_positivity,
               double *ql_ip1, double *qr_i);
/* generic dispatcher: q[5] = (i-2..i+2); unused entries ignored; n = variable index */
void orc_recon_point(int recon, const double q[5], double dx, int n, double *ql_ip1,
                     double *qr_i);
/* vectorised over m stencils, q laid out [m][5] */
void orc_recon_many(int recon, long m, const double *q, double dx, int n, double *ql_ip1,
                    double *qr_i);

/* ---- pointwise Riemann solvers (src/hydro/rsolvers/...) ----
 * wl/wr: primitive L/R states in NATURAL order (rho,v1,v2,v3,p[,B1,B2,B3,psi]);
 * ivx in {1,2,3}; flux returned in NATURAL conserved order (permutation on store as in
 * the reference, e.g. glmmhd_hlld.hpp:385-393). */
void orc_riemann_point(int fluid, int riemann, int ivx, const double *wl, const double *wr,
                       double gamma, double c_h, double *flux);
void orc_riemann_many(int fluid, int riemann, int ivx, long m, const double *wl,
                      const double *wr, double gamma, double c_h, double *flux);

double orc_sound_speed(double gamma, double d, double p);
double orc_fast_speed(double gamma, double d, double p, double bx, double by, double bz);

/* ---- cons->prim for one cell; returns 0 ok, 1 negative density, 2 negative pressure ---- */
int orc_cons_to_prim_cell(int fluid, const orc_eos *eos, int nhydro, int nscalars, double *u,
                          double *w);

/* ---- block level (arrays [nvar][Nk][Nj][Ni], i fastest) ---- */
/* src/hydro/hydro.cpp:1025-1208 ; flux[d] same shape as cons; face i = lower face of cell i */
void orc_calculate_fluxes(const orc_geom *g, int fluid, int recon, int riemann,
                          const orc_eos *eos, double c_h, const double *prim, double *flux1,
                          double *flux2, double *flux3);
/* src/hydro/hydro.cpp:980-1022 (DC + LLF tight loop) */
void orc_calculate_fluxes_tight(const orc_geom *g, int fluid, const orc_eos *eos, double c_h,
                                const double *prim, double *flux1, double *flux2,
                                double *flux3);
/* Parthenon Update::UpdateWithFluxDivergence */
void orc_update_flux_div(const orc_geom *g, double *u0, const double *u1, const double *flux1,
                         const double *flux2, const double *flux3, double gam0, double gam1,
                         double beta_dt);
/* src/hydro/glmmhd/dedner_source.cpp:17-75 */
void orc_dedner_source(const orc_geom *g, int extended, double alpha, double c_h,
                       double mindx, double beta_dt, double *cons, const double *prim);
/* src/eos/adiabatic_hydro.cpp:33-55, adiabatic_glmmhd.cpp:33-56 : entire block incl ghosts.
 * returns number of cells that tripped the negative density/pressure REQUIRE */
long orc_cons_to_prim(const orc_geom *g, int fluid, const orc_eos *eos, double *cons,
                      double *prim);
/* src/hydro/hydro.cpp:828-910 : returns min_d dx_d/(|v_d|+c_d) (WITHOUT cfl) */
double orc_estimate_dt_hyp(const orc_geom *g, int fluid, const orc_eos *eos,
                           const double *prim);
/* src/hydro/hydro.cpp:1223-1342 ; returns total number of corrected cells */
long orc_first_order_flux_correct(const orc_geom *g, int fluid, const orc_eos *eos,
                                  double c_h, const double *u0_cons, const double *u0_prim,
                                  const double *u1_cons, double *flux1, double *flux2,
                                  double *flux3, double gam0, double gam1, double beta_dt);
/* src/hydro/hydro.cpp:145-208 ; out[8] = mass,1-mom,2-mom,3-mom,KE,tot-E,ME,relDivB */
void orc_history(const orc_geom *g, int fluid, const double *cons, double *out);

/* ---- mini-driver (stage ordering of src/hydro/hydro_driver.cpp:347-673) ---- */
typedef struct {
  int fluid, recon, riemann, integrator;
  int nx[3];       /* mesh size */
  int mb[3];       /* meshblock size */
  int ng;
  int nscalars;
  int bc_inner[3], bc_outer[3];
  double xmin[3], xmax[3];
  double cfl;
  double glmmhd_alpha;
  int dedner_extended;
  int first_order_flux_correct;
  orc_eos eos;
  int nthreads; /* OpenMP threads for block loops; 0 = library default */
} orc_sim_params;

typedef struct orc_sim orc_sim;

orc_sim *orc_sim_create(const orc_sim_params *p);
void orc_sim_destroy(orc_sim *s);
int orc_sim_nblocks(const orc_sim *s);
void orc_sim_block_geom(const orc_sim *s, orc_geom *g);
double *orc_sim_cons(orc_sim *s, int b);
double *orc_sim_prim(orc_sim *s, int b);
void orc_sim_block_origin(const orc_sim *s, int b, double x0[3]); /* global index of first interior cell */
/* problem generators: fill interior cons of every block */
/* src/pgen/linear_wave.cpp:72-176,342-376 ; returns period-scaled tlim factor lambda/|ev| */
double orc_pgen_linear_wave(orc_sim *s, int wave_flag, double amp, double vflow);
/* src/pgen/linear_wave_mhd.cpp: wave_flag 0..6 = fast-, Alfven-, slow-, entropy, slow+, Alfven+, fast+; returns the period */
double orc_pgen_linear_wave_mhd(orc_sim *s, int wave_flag, double amp, double vflow);
void orc_linear_wave_mhd_eigen(const orc_sim *s, double *ev7, double *rem49);
double orc_linear_wave_mhd_errors(orc_sim *s, int wave_flag, double amp, double vflow, double *l1_8, double *max_8);
/* src/pgen/sod.cpp:17-51 */
void orc_pgen_sod(orc_sim *s, double rho_l, double pres_l, double u_l, double rho_r,
                  double pres_r, double u_r, double x_discont);
/* src/pgen/orszag_tang.cpp:25-63 */
void orc_pgen_orszag_tang(orc_sim *s);
void orc_pgen_kh(orc_sim *s, int iprob, double vflow, double amp, double drho_rho0, double vboost, double a5,
                 double sigma5, double drat5);
void orc_pgen_field_loop(orc_sim *s, double rad, double amp, double vflow, double drat, int iprob);
double orc_user_reldivb(orc_sim *s, double B0);
void orc_pgen_advection(orc_sim *s, double vx, double vy, double vz, double rho_ratio, double rho_radius,
                        double rho_fraction_edge, double rho0, double p0);
double orc_pgen_cpaw(orc_sim *s, double b_par, double b_perp, double pres, double v_par, int dir, double ang_2,
                     double ang_3);
double orc_cpaw_errors(orc_sim *s, double *err8);
void orc_pgen_lw_implode(orc_sim *s, double d_in, double p_in, double d_out, double p_out);
void orc_pgen_blast(orc_sim *s, double rout, double rin, double pa, double da, double prat, double drat,
                    double x0, double y0, double z0);
/* analytic seedless smooth MHD/hydro state (SURVEY.md 8(d) synthetic kernel benchmark) */
void orc_pgen_synthetic(orc_sim *s);
/* after pgen: ghost exchange -> FillDerived -> first EstimateTimestep (App. A.4) */
void orc_sim_initialize(orc_sim *s);
/* one cycle: PreStep (c_h) -> stages -> time += dt -> new dt.  returns dt used */
double orc_sim_step(orc_sim *s, double tlim);
/* run until tlim or nlim cycles; returns number of cycles */
int orc_sim_run(orc_sim *s, double tlim, int nlim);
double orc_sim_time(const orc_sim *s);
double orc_sim_dt(const orc_sim *s);
double orc_sim_c_h(const orc_sim *s);
long orc_sim_fofc_count(const orc_sim *s);
void orc_sim_history(orc_sim *s, double *out8);
/* src/pgen/linear_wave.cpp:183-335 : l1[5], maxerr[5]; returns RMS-L1 */
double orc_linear_wave_errors(orc_sim *s, int wave_flag, double amp, double vflow, double *l1,
                              double *maxerr);
/* gather interior of all blocks into a global [nvar][nx3][nx2][nx1] array */
void orc_sim_gather_cons(orc_sim *s, double *out);
void orc_sim_exchange_ghosts(orc_sim *s);
void orc_sim_fill_derived(orc_sim *s);

/* ---- few-modes turbulence driver (turbulence.c; BASELINE config 4 forcing) ---------------- */
#include <stdint.h>
typedef struct {
  uint32_t mt[624];
  int idx;
} orc_mt19937; /* std::mt19937 */
void orc_mt_seed(orc_mt19937 *g, uint32_t seed);
uint32_t orc_mt_next(orc_mt19937 *g);
double orc_uniform_m1_p1(orc_mt19937 *g); /* std::uniform_real_distribution<>(-1,1) (libstdc++) */

typedef struct {
  int num_modes;
  double k_peak, sol_weight, t_corr;
  double *k_vec;       /* [3][M] */
  double *var_hat;     /* [3][M][2] (re, im) */
  double *var_hat_new; /* [3][M][2] */
  orc_mt19937 rng;
} orc_fmft;
orc_fmft *orc_fmft_create(int num_modes, const double *k_vec, double k_peak, double sol_weight,
                          double t_corr, uint32_t rseed);
void orc_fmft_destroy(orc_fmft *f);
void orc_fmft_evolve(orc_fmft *f, double dt);
void orc_fmft_phases(const orc_fmft *f, int axis, int n, int g0, int gn, double *out);
void orc_fmft_inverse(const orc_fmft *f, const orc_geom *g, const double *ph_i, const double *ph_j,
                      const double *ph_k, double *acc);
void orc_turb_perturb(int nblocks, const orc_geom *g, double **cons, double **acc, double dt,
                      double accel_rms, double box_volume);
void orc_turb_history(const orc_geom *g, int fluid, double gamma, const double *prim, double *out3);
/* src/pgen/turbulence.cpp:217-370 (b_config 0, 1, 2) + enrols the driver on the sim */
void orc_pgen_turbulence(orc_sim *s, double rho0, double p0, double b0, int b_config, int num_modes,
                         const double *k_vec, double k_peak, double sol_weight, double t_corr,
                         double accel_rms, uint32_t rseed);
void orc_sim_turb_history(orc_sim *s, double *out3);
const double *orc_sim_var_hat(const orc_sim *s);
double *orc_sim_acc(orc_sim *s, int b);

/* integrator coefficient table; returns nstages */
int orc_integrator_coeffs(int integrator, double *beta, double *gam0, double *gam1);

#ifdef __cplusplus
}
#endif

/* ---- mesh-refinement operators and tagging (amr.c; SURVEY 8(f) rank 3 building blocks) ------ */
typedef struct {
  int nx[3];      /* fine interior cells of the meshblock (1 in a collapsed dimension) */
  int ng;         /* fine ghost cells */
  int cng;        /* ghost cells of the coarse buffer */
  double xmin[3]; /* lower interior corner of the block */
  double dx[3];   /* fine cell widths */
} orc_refine_geom;
int orc_refine_ndim(const orc_refine_geom *r);
void orc_refine_dims(const orc_refine_geom *r, int fine[3], int coarse[3]);
void orc_prolongate_minmod(const orc_refine_geom *r, int nvar, const double *coarse, double *fine,
                           const int lo[3], const int hi[3]);
void orc_restrict_average(const orc_refine_geom *r, int nvar, int el, const double *fine, double *coarse,
                          const int lo[3], const int hi[3]);
int orc_tag_pressure_gradient(const orc_geom *g, const double *prim, double threshold, double *crit);
int orc_tag_velocity_gradient(const orc_geom *g, const double *prim, double threshold, double *crit);
int orc_tag_max_density(const orc_geom *g, const double *prim, double refine_above, double deref_below,
                        double *crit);


/* ---- branch tracing (test infrastructure for the crafted edge-case fixture) ---------------------
 * When orc_trace_sink points at a word, the pointwise functions OR the bits of the branches they
 * take into it.  tests/golden/make_edge_cases.py records, per crafted case, which of the
 * reference's special-case branches it exercises. */
enum {
  ORC_TR_PPM_LIM_M = 1 << 0,     /* step 2a replaced the lower interface value (ppm_simple.hpp:66-80) */
  ORC_TR_PPM_LIM_P = 1 << 1,     /* ... the upper one (:82-98) */
  ORC_TR_PPM_EXTREMUM = 1 << 2,  /* local extremum branch of step 4 (:139) */
  ORC_TR_PPM_ROUNDOFF = 1 << 3,  /* second derivative within 1e-12 of round-off: ratio forced to 0 (:129-136) */
  ORC_TR_PPM_RATIO_BIG = 1 << 4, /* extremum kept because the limited ratio is ~1 (:141) */
  ORC_TR_PPM_OVER_M = 1 << 5,    /* overshoot limiter on the lower state (:150) */
  ORC_TR_PPM_OVER_P = 1 << 6,    /* ... on the upper state (:154) */
  ORC_TR_HLLD_FL = 1 << 8,       /* returned F_L (s0 >= 0) */
  ORC_TR_HLLD_FR = 1 << 9,       /* F_R (s4 <= 0) */
  ORC_TR_HLLD_LSTAR = 1 << 10,
  ORC_TR_HLLD_LDSTAR = 1 << 11,
  ORC_TR_HLLD_RDSTAR = 1 << 12,
  ORC_TR_HLLD_RSTAR = 1 << 13,
  ORC_TR_HLLD_DEG_L = 1 << 14,   /* degenerate left star state (glmmhd_hlld.hpp:196-202) */
  ORC_TR_HLLD_DEG_R = 1 << 15,   /* degenerate right star state (:228-234) */
  ORC_TR_HLLD_DEG_DST = 1 << 16, /* Bx ~ 0: double-star states = star states (:252-255) */
  ORC_TR_HLLC_AM_POS = 1 << 18,  /* contact moves right (hydro_hllc.hpp:126) */
  ORC_TR_HLLC_CP_CLIP = 1 << 19, /* contact pressure clipped to 0 (:112) */
  ORC_TR_HLLC_QL = 1 << 20,      /* left shock correction q_l > 1 (:63) */
  ORC_TR_HLLC_QR = 1 << 21,
  ORC_TR_HLLE_BP_EQ_BM = 1 << 22, /* bp == bm: no averaging weight (hydro_hlle.hpp:127, glmmhd_hlle.hpp:180) */
  ORC_TR_C2P_DFLOOR = 1 << 24,
  ORC_TR_C2P_VCEIL = 1 << 25,
  ORC_TR_C2P_PFLOOR = 1 << 26,
  ORC_TR_C2P_EFLOOR = 1 << 27,
  ORC_TR_C2P_ECEIL = 1 << 28
};
extern __thread unsigned *orc_trace_sink;
#define ORC_TRACE(bit) do { if (orc_trace_sink) *orc_trace_sink |= (unsigned)(bit); } while (0)
void orc_recon_many_traced(int recon, long m, const double *q5, double dx, int n, double *ql, double *qr, unsigned *masks);
void orc_c2p_many_traced(int fluid, const orc_eos *eos, long m, double *u, double *w, int *status, unsigned *masks);
void orc_riemann_many_traced(int fluid, int riemann, int ivx, long m, const double *wl, const double *wr, double gamma,
                             double c_h, double *flux, unsigned *masks);

#endif /* APK_ORACLE_H_ */
