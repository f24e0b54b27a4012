// sim.hpp -- standalone host driver above the hot-path boundary: the "Hydro" package
// options (src/hydro/hydro.cpp:264-826), the per-cycle c_h update (hydro.cpp:102-143), the
// per-stage task order (src/hydro/hydro_driver.cpp:347-673) and Parthenon's dt control
// (SURVEY.md App. A.2-A.4).  All compute goes through the C-ABI of include/apk_amd.h.
#pragma once

#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/apk_amd.h"
#include "../../../include/apk_host.h"
#include "amr.hpp"
#include "mesh.hpp"
#include "params.hpp"
#include "turbulence.hpp"

namespace apk {

// "Hydro" StateDescriptor params (names as in hydro.cpp)
struct HydroPackage {
  int fluid = APK_FLUID_EULER;
  int recon = APK_RC_UNDEFINED;
  int riemann = APK_RS_UNDEFINED;
  int integrator = APK_INT_UNDEFINED;
  int nhydro = 5, nscalars = 0;
  double cfl = 0.3;
  apk_eos eos{};
  bool glmmhd_source_extended = false;  // hydro/glmmhd_source
  double glmmhd_alpha = 0.1;
  bool calc_c_h = false, calc_dt_hyp = true;
  bool first_order_flux_correct = false;
  double max_dt = -1.0;
  // mutable params
  double c_h = 0.0;
  double mindx = std::numeric_limits<double>::max();
  double dt_hyp = std::numeric_limits<double>::max();
  // flux function keys (FluxFunKey_t) of the first and the other stages (hydro.cpp:449-467)
  apk_flux_cfg flux_first_stage{}, flux_other_stage{};
};

struct LinearWaveState {  // globals of src/pgen/linear_wave.cpp
  int wave_flag = 0;
  double amp = 0.0, vflow = 0.0;
  double sin_a2 = 0, cos_a2 = 1, sin_a3 = 0, cos_a3 = 1, k_par = 0, lambda = 1;
  double d0 = 1, p0 = 0, u0 = 0, gam = 0, gm1 = 0, ev[5] = {0}, rem[5][5] = {{0}};
  bool compute_error = false;
};

struct LinearWaveMhdState {  // what src/pgen/linear_wave_mhd.cpp keeps besides the hydro wave's geometry (LinearWaveState)
  double bx0 = 1.0, by0 = 0.0, bz0 = 0.0, dby = 0.0, dbz = 0.0, ev[7] = {0}, rem[7][7] = {{0}};
};

struct CpawState {  // globals of src/pgen/cpaw.cpp
  double den = 1.0, pres = 0, gm1 = 0, b_par = 0, b_perp = 0, v_perp = 0, v_par = 0, fac = 1.0;
  double sin_a2 = 0, cos_a2 = 1, sin_a3 = 0, cos_a3 = 1, lambda = 1, k_par = 0;
  bool compute_error = false;
};

struct FieldLoopState {  // parameters of src/pgen/field_loop.cpp:129-170; B0 normalises UserRelDivB
  int iprob = 1;
  double rad = 0, amp = 0, vflow = 0, drat = 1.0, cos_a2 = 0, sin_a2 = 0, lambda = 0;
};

}  // namespace apk

namespace apk {
namespace host {
struct RcclTransport;  // comm_rccl.cpp
void rccl_transport_destroy(RcclTransport *t);
// one-GPU rehearsal (mesh.hpp "rehearse"): the transport's streams and events without RCCL, every peer's message
// delivered by a device copy on the halo stream from this rank's own send buffer
int comm_loopback_attach(apk_sim *s);
}  // namespace host
}  // namespace apk

struct apk_sim {
  apk::host::RcclTransport *rccl = nullptr;  // native transport (apk_sim_comm_rccl); comm.* point into it
  apk::ParameterInput pin;
  apk::Mesh mesh;
  apk::HydroPackage pkg;
  std::string problem_id;
  apk::LinearWaveState lw;
  apk::LinearWaveMhdState lwm;
  apk::CpawState cpaw;
  apk::FieldLoopState floop;
  bool host_only = false;
  bool fused = true;
  int rank = 0, nranks = 1;
  double xmin[3] = {0, 0, 0}, xmax[3] = {1, 1, 1}, dx[3] = {1, 1, 1};
  // SimTime
  double time = 0.0, dt = std::numeric_limits<double>::max(), tlim = 1.0;
  int nlim = -1, ncycle = 0;
  long long fofc_total = 0;
  long long fofc_fallback_stages = 0;
  bool stage_dt_pending = false;  // the last fused stage already reduced the hyperbolic dt
  bool dt_hyp_is_global = false;  // pkg.dt_hyp holds the minimum over all ranks (no reduction needed in pre_step)
  // integrator (Parthenon LowStorageIntegrator)
  int nstages = 0;
  double beta[4] = {0}, gam0[4] = {0}, gam1[4] = {0};
  // device state
  apk_ctx *ctx = nullptr;
  apk_stream_t stream = nullptr;
  apk_allocator alloc{};
  bool have_alloc = false;
  apk_comm_ops comm{};
  bool have_comm = false;
  // doubles per field per block: nper = the slot a block has in a field's allocation (a multiple of 16 when rows are
  // padded), nblk = what is addressed from the block's pointer (nvar * sn), which sits mesh.lead doubles into the slot
  int64_t nper = 0, nblk = 0;
  double *blk(double *field, int64_t lb) const { return field + lb * nper + mesh.lead; }
  // Two conserved-variable buffers: the stage-1 "u1 <- u0" DeepCopy of the reference
  // (hydro_driver.cpp:474-495) is a buffer-role swap here -- stage 1 has gam0 = 0 for every
  // integrator, so it reads the old state as u1 and writes the new state into the other buffer.
  // (a third buffer of the same layout is allocated on first use as the output of trial stages with
  // gam0 != 0 -- first-order flux correction in the later stages of RK2 / RK3 -- whose input u0 must
  // survive a rejected trial; an accepted trial just makes it the current buffer)
  double *d_cons2[3] = {nullptr, nullptr, nullptr};
  int freebuf() const { return 3 - cur - u1buf; }  // the buffer that is neither u0 nor u1
  int cur = 0;    // buffer holding the current state u0 ("base")
  int u1buf = 1;  // buffer holding the register u1 of the step in flight
  // Two primitive-variable buffers as well (the second one is allocated on first use): a stage
  // whose finishing kernel cannot replace prim in place (3-D donor cell, see fused_dc3_kernel)
  // writes the new primitives into the other buffer and the roles swap.
  double *d_prim2[2] = {nullptr, nullptr};
  int pcur = 0;  // buffer holding the primitives of the current state
  double *d_flux[3] = {nullptr, nullptr, nullptr};
  std::vector<double *> send_buf, recv_buf;
  // per buffer: the pack presenting it as MeshData "base" (with prim/flux), the pack presenting it
  // as "u1" (cons, plus the spare prim buffer when there is one), and the ghost-exchange plans that target it
  // [cons buffer][prim buffer]; mu1 packs carry the OTHER prim buffer's arrays as "u1.prim"
  apk_pack *mu0_of[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}},
           *mu1_of[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  apk_copy_plan *plans_of[3][apk::PH_COUNT] = {};
  // the same plans over the two primitive buffers: exchanges that carry PRIMITIVES (GHOST_PRIM_COPY: VL2's half-step
  // state, whose conserved values nobody reads) -- the messages then hold what the face table shows a same-rank reader,
  // the neighbour's stored primitives, and neither side converts anything
  apk_copy_plan *pplans_of[2][apk::PH_COUNT] = {};
  int xchg_prim = 0;  // the primitive buffer such an exchange packs from and fills
  // refined meshes: the boundary-plane fluxes of a fused stage's flux correction run beside the stage kernels
  // (amr_flux_planes_ahead)
  void *side_stream = nullptr, *ev_fork = nullptr, *ev_join = nullptr;
  bool side_stream_failed = false;  // its creation failed once: not tried again
  apk_pack *mu0() const { return mu0_of[cur][pcur]; }
  apk_pack *mu1() const { return mu1_of[u1buf][pcur]; }
  double *d_prim() const { return d_prim2[pcur]; }
  apk_copy_plan *plan(int ph) const { return plans_of[cur][ph]; }
  double *d_cons() const { return d_cons2[cur]; }
  // few-modes turbulence driver (problem_id = turbulence; src/pgen/turbulence.cpp:103-200)
  std::unique_ptr<apk::FewModesFT> fmft;
  double accel_rms = 0.0;
  double *d_acc = nullptr;     // [nblocks][3][Nk][Nj][Ni]
  double *d_phases = nullptr;  // per block: phases_i | phases_j | phases_k
  apk_fmft *fm_dev = nullptr;
  // Overlap of the halo exchange between two stages with the x1 sweep of the next stage
  // (BASELINE north_star / SURVEY 8(e)): after a stage the messages are posted and left in flight
  // (exchange_pending); the next stage runs its x1 sweep on all cells farther than nghost from a
  // face with a remote neighbour, completes the exchange, then does the thin slabs and the rest.
  bool overlap = true;
  bool exchange_pending = false;
  // One-layer exchanges (mesh.hpp PH_PACK_THIN): the exchange at the end of a cycle whose first stage is the
  // donor-cell predictor.  thin_msgs: the message set apk_sim_peer reports is the one-layer one (transports read it at
  // every exchange); xchg_thin: the exchange begun last; remote_ghosts_thin: the ghost zones filled by messages are up
  // to date one layer deep only -- sync_ghosts completes them (a collective, like the one of refined meshes).
  bool thin_on = true, thin_msgs = false, xchg_thin = false, remote_ghosts_thin = false;
  long long thin_exchanges = 0;
  // x1 strips without pack / unpack kernels (apk_stage_args.x1_halo; x1_direct_cycle): the finishing kernels of a VL2
  // cycle store their x1 boundary columns straight into the send buffers (the predictor: primitives, nghost deep, into
  // the full messages; the corrector: the conserved state, one layer, into the one-layer messages) and read their x1
  // ghost columns straight from the receive buffers.  d_x1_tab[0]: the predictor's table (receive: one-layer segments,
  // send: full ones), [1]: the corrector's (receive: full, send: one-layer).
  //   x1_out_direct   the stage just run has stored its x1 strips: the exchange posted next packs without them
  //   xchg_x1_direct  the exchange begun last leaves the x1 ghost columns in the receive buffers (unpacks without them)
  //   x1_in_recv      ... and has completed: the next stage reads them there.  Always together with a one-layer exchange
  //                   (remote_ghosts_thin: whoever else reads ghost zones repeats the exchange in full) or inside a cycle.
  bool x1_on = true;  // apk_sim_set_x1_direct / APK_X1_DIRECT=0 (A/B)
  void *d_x1_tab[3] = {nullptr, nullptr, nullptr};  // ([2]: full messages both ways -- the RK integrators)
  bool x1_out_direct = false, xchg_x1_direct = false, x1_in_recv = false;
  long long x1_direct_exchanges = 0;
  long long turb_dt_kicks = 0;  // kicks that estimated the time step without storing primitives (apk_turb_apply_dt)
  int pending_c2p = 0;  // the exchange in flight converts ghost zones as it fills them (GHOST_C2P / GHOST_PRIM_ONLY)
  int pending_cons = 0;  // cons buffer whose ghost zones the exchange in flight fills (roles may swap meanwhile)
  // device window tables (apk_stage_args.window, 8 ints per block) of the split stages:
  // x1 sweep: main / low slab / high slab; 3-D donor-cell stage: main / z lo,hi / y lo,hi / x lo,hi
  struct WindowTable {
    int *d = nullptr;
    int rl = 0, rows = 0;
    bool any = false;  // some block has work in this table
  };
  WindowTable x1win[3], dcwin[7];
  WindowTable k3win[3];  // two-kernel stages: plane windows of the x3 sweep (main, low slab, high slab)
  unsigned *d_late_regions = nullptr;  // per block: bit (sx+1)+3(sy+1)+9(sz+1) = that neighbour region is filled late
  // Direct neighbour addressing (apk_stage_args.face_neighbor; uniform 3-D meshes): per local block
  // and face the local index of the same-rank block behind it, or -1.  While every stage of the
  // cycle is one of the kernels that follow the table, the same-rank ghost copies are skipped
  // altogether and the ghost zones behind those faces go stale; whoever needs them (accessors, a
  // stage form that reads ghost zones) calls sync_ghosts() / materialize_local_ghosts() first.
  int *d_face_nbr = nullptr;
  bool local_ghosts_stale = false;
  bool direct_on = true;  // apk_sim_set_direct_neighbors
  // The primitives of the current (full-step) state are not in memory: the last stage of the cycle computed them for
  // the time-step estimate only (apk_stage_args.fill_derived = 3) because the first stage of the next cycle -- the
  // donor-cell predictor -- derives its input from the conserved state (prim_from_cons).  Whatever else reads
  // primitives goes through sync_ghosts(), which materialises them (ConsToPrim of every block).
  bool prim_stale = false;
  bool prim_free_on = true;  // apk_sim_set_prim_free / APK_PRIM_FREE=0 (A/B)
  long long skipped_local_exchanges = 0;
  // ... or the criterion itself was reduced with the time-step estimate at the end of the last stage
  // (apk_tag_blocks_dt_from_cons): amr_tags_begin hands this request on instead of launching one
  bool amr_tags_posted = false;
  int amr_posted_criterion = -1, amr_posted_pending = 0;
  double amr_posted_p0 = 0.0, amr_posted_p1 = 0.0;
  bool amr_tag_vars_stored = false;  // the primitives a refinement criterion reads are those of the current state (amr_prim_free_cycle)
  long long amr_c2p_passes_skipped = 0;  // ConsToPrim passes between the stages a refined mesh did without (amr_prim_free_cycle)
  // mesh refinement (parthenon/mesh/refinement = static | adaptive; one rank): the forest of
  // blocks, the index-box plans of the multilevel ghost exchange / flux correction and their device
  // forms, one set per cons buffer the exchange can target
  std::unique_ptr<apk::AmrTree> amr;
  apk::AmrGeom amr_geom;
  apk::AmrPlans amr_plans;
  bool amr_adaptive = false;
  int amr_derefine_count = 10, amr_check_interval = 1;
  long long amr_refined = 0, amr_derefined = 0;  // blocks refined / merged so far
  long long zone_cycles = 0;                     // sum over cycles of the interior cells updated
  long long perf_zone_mark = 0;                  // its value where the timed part of apk_sim_execute began
  double *d_coarse = nullptr;                    // [nblocks][amr_geom.coarse_doubles]
  // distribution over ranks: contiguous Z-order ranges; amr_plans is the global plan (every rank
  // builds the same one), amr_local this rank's share with local block numbers, copies that cross
  // ranks turned into packs / unpacks of one message per peer
  apk::AmrPartition amr_part;
  struct AmrLocalPlans {
    std::vector<apk::AmrRefOp> restrict_own, prolongate, flux_restrict[3];
    std::vector<apk::BoxRegion> fill, fill_pack, fill_unpack, coarse_bc[3], fine_bc[3];
    std::vector<apk::BoxRegion> flux_copy[3], flux_pack[3], flux_unpack[3];
    // the correction after a fused stage averages the fine fluxes of same-rank faces inside the fix kernel: the
    // restriction operators of flux_copy[d], entry for entry, and those whose coarse side lives on another rank (the
    // only ones that still run as operators there)
    std::vector<apk::AmrRefOp> flux_fused_ops[3], flux_restrict_remote[3];
    // the exchange of the stage loop: without the boxes that fill ghost zones behind edges and corners
    // (BoxRegion::corner); its messages are laid out by the same walk over the filtered global list
    std::vector<apk::AmrRefOp> prolongate_faces;
    std::vector<apk::BoxRegion> fill_faces, fill_pack_faces, fill_unpack_faces;
    // ... and without the same-rank copies between blocks of one level either (BoxRegion::same_face): the exchange
    // after a stage when the next one reads those neighbours through the face table (amr_direct)
    std::vector<apk::BoxRegion> fill_direct;
    // the exchange before a refinement check when the next cycle's first stage reads at most AMR_SHELL_DEPTH ghost
    // layers: every ghost zone, edges and corners too, but only that deep (BuildAmrPlans(fill_depth))
    std::vector<apk::AmrRefOp> prolongate_shell;
    std::vector<apk::BoxRegion> fill_shell, fill_pack_shell, fill_unpack_shell, fine_bc_shell[3];
    std::vector<apk::BoxRegion> fill_shell_direct;  // fill_shell without the same-level face copies
  } amr_local;
  struct MsgSet {
    apk::AmrMessages plan;
    std::vector<double *> send, recv;
    std::vector<int64_t> send_cap, recv_cap;
  };
  MsgSet amr_halo, amr_fluxmsg, amr_move, amr_halo_faces, amr_halo_shell;
  // refined meshes: the stage loop fills (and converts to primitives) only the ghost zones behind block FACES
  // (AMR_GHOSTS_FACES), or all of them AMR_SHELL_DEPTH layers deep before a refinement check (AMR_GHOSTS_SHELL: what
  // the tagging criteria and a donor-cell / PLM first stage read); accessors and regridding complete them first
  // (sync_ghosts)
  int amr_ghost_state = 0;  // AMR_GHOSTS_COMPLETE
  bool amr_full_exchange = false;  // apk_sim_set_amr_full_exchange
  const MsgSet *active_msgs = nullptr;  // the message set apk_sim_peer reports (null: the uniform mesh's)
  long long msg_generation = 0;         // bumped whenever that set, its sizes or its buffers change
  struct AmrDevice {
    std::vector<apk_refine_plan *> restrict_own[2], prolongate[2], flux_restrict[3], flux_restrict_remote[3];
    apk_copy_plan *fill[2] = {nullptr, nullptr}, *fill_pack[2] = {nullptr, nullptr}, *fill_unpack[2] = {nullptr, nullptr};
    apk_copy_plan *flux_pack[3] = {nullptr, nullptr, nullptr}, *flux_unpack[3] = {nullptr, nullptr, nullptr};
    // the correction applied after a fused stage instead (apk_flux_fix_plan): same-rank faces and
    // faces whose fine side arrived in a message, per cons buffer the stage wrote
    // The launches of the multilevel exchange before / after the message exchange take no per-cycle
    // arguments: each half is captured once per mesh into a hipGraph and replayed as ONE launch
    // (void* = hipGraphExec_t; null = not captured, the plans are launched one by one)
    void *xchg_pre[2] = {nullptr, nullptr}, *xchg_post[2] = {nullptr, nullptr};
    // no messages between the two halves (one rank): the `pre` graphs hold both halves -- one graph launch per exchange
    bool xchg_whole = false;
    // the faces-only exchange of the stage loop (AmrLocalPlans::fill_faces ...)
    std::vector<apk_refine_plan *> prolongate_faces[2];
    apk_copy_plan *fill_faces[2] = {nullptr, nullptr}, *fill_pack_faces[2] = {nullptr, nullptr}, *fill_unpack_faces[2] = {nullptr, nullptr};
    void *xchg_pre_faces[2] = {nullptr, nullptr}, *xchg_post_faces[2] = {nullptr, nullptr};
    apk_copy_plan *fill_direct[2] = {nullptr, nullptr};  // (AmrLocalPlans::fill_direct; the other plans of the faces-only exchange)
    void *xchg_pre_direct[2] = {nullptr, nullptr};
    // the shell exchange (AmrLocalPlans::fill_shell ...)
    std::vector<apk_refine_plan *> prolongate_shell[2];
    apk_copy_plan *fill_shell[2] = {nullptr, nullptr}, *fill_pack_shell[2] = {nullptr, nullptr}, *fill_unpack_shell[2] = {nullptr, nullptr};
    apk_copy_plan *fine_bc_shell[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    void *xchg_pre_shell[2] = {nullptr, nullptr}, *xchg_post_shell[2] = {nullptr, nullptr};
    // the shell exchange without the same-rank same-level face copies (AMR_XCHG_SHELL_DIRECT: the tag kernel, ConsToPrim
    // and the next predictor follow the face table there)
    apk_copy_plan *fill_shell_direct[2] = {nullptr, nullptr};
    void *xchg_pre_shell_direct[2] = {nullptr, nullptr};
    apk_flux_fix_plan *flux_fix[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    apk_flux_fix_plan *flux_fix_all[2] = {nullptr, nullptr}, *flux_fix_unpack_all[2] = {nullptr, nullptr};  // all directions in one launch
    // the faces (6 * local block + face) with a coarser or finer block behind them: the only boundary-plane
    // fluxes the correction after a fused stage reads (apk_calculate_fluxes_boundary_list)
    int *d_cf_faces = nullptr;
    int n_cf_faces = 0;
    apk_flux_fix_plan *flux_fix_unpack[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    apk_copy_plan *coarse_bc[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    apk_copy_plan *fine_bc[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    apk_copy_plan *flux_copy[3] = {nullptr, nullptr, nullptr};
  } amr_dev;
  long long overlapped = 0;
  int perf_cycles = 0;        // cycles inside loop_seconds (after parthenon/time/perf_cycle_offset)
  double loop_seconds = 0.0;  // wall time of the last apk_sim_execute main loop (device synchronised)
  std::string err;
};
