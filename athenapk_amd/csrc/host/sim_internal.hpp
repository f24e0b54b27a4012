// sim_internal.hpp -- what the translation units of the standalone host driver share (sim.cpp: set-up,
// stage loop, C API; sim_pgen.cpp: problem generators; sim_amr.cpp: refined meshes).  Not installed.
#pragma once

#include "sim.hpp"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <sstream>
#include <unordered_map>
#include <unordered_set>

namespace apk {
namespace host {

constexpr double kHuge = std::numeric_limits<double>::max();

inline int fail(apk_sim *s, int code, const std::string &msg) {
  if (s) s->err = msg;
  return code;
}

#define SIM_TRY(s, expr)                                                        \
  do {                                                                          \
    int rc__ = (expr);                                                          \
    if (rc__ != APK_OK) {                                                       \
      if ((s)->err.empty() && (s)->ctx) (s)->err = apk_last_error((s)->ctx);    \
      if ((s)->err.empty()) (s)->err = #expr;                                   \
      return rc__;                                                              \
    }                                                                           \
  } while (0)

#define SIM_HIP(s, expr)                                                        \
  do {                                                                          \
    hipError_t e__ = (expr);                                                    \
    if (e__ != hipSuccess) return fail((s), APK_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

inline hipStream_t hs(const apk_sim *s) { return reinterpret_cast<hipStream_t>(s->stream); }

int parse_bc(const std::string &v);
void hydro_initialize(apk_sim *s);
void mesh_initialize(apk_sim *s);
int dev_alloc(apk_sim *s, const char *tag, size_t bytes, double **out);
void dev_free(apk_sim *s, double *p);
int build_packs(apk_sim *s);
int ensure_spare_prim(apk_sim *s);
int ensure_flux_arrays(apk_sim *s);
bool stage_can_fuse(const apk_sim *s);
int build_copy_plans(apk_sim *s);
int build_prim_plans(apk_sim *s);
void set_global_dt(apk_sim *s, double dt_est);
struct DtEstimate {  // what one rank measured: see estimate_timestep_read / _commit
  double dt_hyp_local = kHuge;
  unsigned flags = 0;
};
int estimate_timestep_read(apk_sim *s, DtEstimate *e);
int estimate_timestep_commit(apk_sim *s, const DtEstimate &e, double *dt_out);
int estimate_timestep(apk_sim *s, double *dt_out);  // read + commit
bool ghost_c2p_fusable(const apk_sim *s);
// how a ghost-zone fill treats the primitives: not at all / ConsToPrim of every cell it fills / that, without storing the
// conserved values (ghost zones of a state whose conserved values nothing reads: VL2's half step)
// GHOST_PRIM_COPY: the exchange moves the stored primitives themselves (plans over the primitive buffers)
enum { GHOST_COPY = 0, GHOST_C2P = 1, GHOST_PRIM_ONLY = 2, GHOST_PRIM_COPY = 3 };
int run_ghost_plan(apk_sim *s, int buf, int phase, int c2p, apk_stream_t stream = nullptr);
int exchange_begin(apk_sim *s, bool async, int c2p, bool skip_local = false, bool thin = false);
bool thin_exchange_cycle(const apk_sim *s);
bool x1_direct_cycle(const apk_sim *s);
int x1_direct_kind(const apk_sim *s);
int build_x1_tables(apk_sim *s);
bool rk_prim_free_cycle(const apk_sim *s);
int materialize_remote_ghosts(apk_sim *s);
int exchange_end(apk_sim *s, int c2p);
int exchange_ghosts(apk_sim *s, int c2p = GHOST_COPY, bool skip_local = false, bool thin = false);
int upload_window(apk_sim *s, const char *tag, const std::vector<int> &w, apk_sim::WindowTable &t);
int build_windows(apk_sim *s);
bool can_overlap_next(const apk_sim *s, int next);
int finish_pending(apk_sim *s);
bool direct_neighbors(const apk_sim *s);
bool amr_faces_only(const apk_sim *s);
bool amr_prim_free_cycle(const apk_sim *s);
bool regrid_check_follows(const apk_sim *s);
int materialize_local_ghosts(apk_sim *s, int buf = -1);
int sync_ghosts(apk_sim *s);  // finish_pending + materialize_local_ghosts
int fill_derived(apk_sim *s);
int pre_step(apk_sim *s);
int turbulence_device_setup(apk_sim *s);
int turbulence_driving(apk_sim *s, double dt, bool fill = false, bool no_prim = false);
int do_stage(apk_sim *s, int stage);
double xc(const apk_sim *s, const double x0[3], int d, int idx);
void block_origin(const apk_sim *s, int lb, double x0[3]);
void lw_eigensystem(double gm1, double v1, double v2, double v3, double h, double ev[5], double rem[5][5]);
void lw_setup(apk_sim *s);
void lw_state(const LinearWaveState &lw, double x1, double x2, double x3, double u[5]);
void lwm_setup(apk_sim *s);
void lwm_state(const apk_sim *s, double x1, double x2, double x3, double u[8]);  // analytic d, M, E, B (no psi)
void cpaw_setup(apk_sim *s);
void cpaw_potential(const CpawState &c, double x1, double x2, double x3, double A[3]);
void cpaw_state(const CpawState &c, double X1, double X2, double X3, double m[3], double b[3]);
void field_loop_potential(const FieldLoopState &f, double x1, double x2, double x3, double A[3]);
void kh_setup(apk_sim *s);
void field_loop_setup(apk_sim *s);
void pgen_block(apk_sim *s, int lb, std::vector<double> &u);
void turbulence_setup(apk_sim *s);
int pgen_turbulence(apk_sim *s, std::vector<std::vector<double>> &blocks);
double level_dx(const apk_sim *s, int level, int d);
const AmrLeaf &amr_leaf(const apk_sim *s, int lb);
int block_level(const apk_sim *s, int lb);
void amr_sync_mesh(apk_sim *s);
void amr_localize(apk_sim *s);
void amr_initialize(apk_sim *s, bool adaptive);
double *amr_base(apk_sim *s, int parity, int kind, int block, const apk_sim::MsgSet *msgs);
apk_copy_region to_copy_region(const BoxRegion &r, const double *src, double *dst);
int amr_make_refine_plans(apk_sim *s, int parity, const std::vector<AmrRefOp> &ops, std::vector<apk_refine_plan *> &out);
void amr_destroy_device_plans(apk_sim *s);
int amr_ensure_buffers(apk_sim *s, apk_sim::MsgSet &m, const char *name);
void amr_free_buffers(apk_sim *s, apk_sim::MsgSet &m);
int amr_exchange_messages(apk_sim *s, const apk_sim::MsgSet &m);
int amr_allocate(apk_sim *s, size_t n, double *cons2[2], double **prim, double *flux[3], double **coarse);
int amr_rebuild(apk_sim *s);
enum { AMR_XCHG_FULL = 0, AMR_XCHG_FACES = 1, AMR_XCHG_DIRECT = 2, AMR_XCHG_SHELL = 3, AMR_XCHG_SHELL_DIRECT = 4 };  // which ghost zones the multilevel exchange fills
enum { AMR_GHOSTS_COMPLETE = 0, AMR_GHOSTS_SHELL = 1, AMR_GHOSTS_FACES = 2, AMR_GHOSTS_SHELL_DIRECT = 3 };  // apk_sim::amr_ghost_state
constexpr int AMR_SHELL_DEPTH = 2;  // ghost layers of the shell exchange (refinement/gradient.cpp:33-36 reads [s-2, e+2]^3)
int amr_exchange(apk_sim *s, int buf, int mode = AMR_XCHG_FULL);
bool amr_direct(const apk_sim *s);
bool amr_has_shell(const apk_sim *s);
bool amr_shell_before_check(const apk_sim *s);
int amr_exchange_pre(apk_sim *s, int buf, int mode);
int amr_exchange_post(apk_sim *s, int buf, int mode);
void amr_capture_half(apk_sim *s, int buf, bool pre, int mode, void **out, bool whole = false);
void amr_destroy_graphs(apk_sim *s);
bool amr_has_coarse_fine_faces(const apk_sim *s);
int amr_flux_fix(apk_sim *s, const apk_flux_cfg &cfg, double beta_dt, double psi_factor, bool planes_ahead = false, int cons_input = -1);
bool amr_flux_planes_ahead(apk_sim *s, const apk_flux_cfg &cfg, int cons_input = -1);
int amr_flux_correction(apk_sim *s);
int refinement_criterion(apk_sim *s, int *criterion, double *p0, double *p1);
bool amr_update_tree(apk_sim *s, const std::vector<int> &tags, bool allow_derefine);
int amr_transfer(apk_sim *s, const std::vector<AmrLeaf> &old, const AmrPartition &old_part);
struct AmrTagRequest {  // a tag reduction in flight (amr_tags_begin .. amr_tags_end)
  int criterion = -1, pending = 0;
  double p0 = 0.0, p1 = 0.0;
};
int amr_tags_begin(apk_sim *s, AmrTagRequest *req);
int amr_tags_end(apk_sim *s, const AmrTagRequest &req, std::vector<int> &tags);
int amr_global_tags(apk_sim *s, std::vector<int> &tags);
int amr_reallocate(apk_sim *s);
int amr_regrid(apk_sim *s, bool *changed, const AmrTagRequest *posted = nullptr);

// On refined meshes the problem generators and the error norms see the cell widths of the block
// they work on through s->dx (restored on scope exit)
struct LevelDxScope {
  apk_sim *s;
  double saved[3];
  LevelDxScope(apk_sim *sim, int lb) : s(sim) {
    for (int d = 0; d < 3; ++d) {
      saved[d] = s->dx[d];
      if (s->amr && s->mesh.Active(d)) s->dx[d] = saved[d] / (double)(1 << amr_leaf(s, lb).level);
    }
  }
  ~LevelDxScope() {
    for (int d = 0; d < 3; ++d) s->dx[d] = saved[d];
  }
};

}  // namespace host
}  // namespace apk
