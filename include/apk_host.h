/*
 * apk_host.h -- C-ABI of the host-side mini-driver that sits ABOVE the hot-path boundary
 * (include/apk_amd.h) in the standalone build.  It mirrors the reference's control flow
 * for this path only: input deck -> Hydro::Initialize options (src/hydro/hydro.cpp:264-826)
 * -> per-cycle PreStepMeshUserWorkInLoop (hydro.cpp:102-143) -> per-stage task order of
 * HydroDriver::MakeTaskCollection (src/hydro/hydro_driver.cpp:347-673) -> dt control.
 * In a real AthenaPK build Parthenon plays this role and only apk_amd.h is bound
 * (INTEGRATION.md).  Everything here is C++ compiled into libapk_amd.so; Python (tests,
 * bench.py) talks to it through ctypes and supplies device memory / torch.distributed.
 */
#ifndef APK_HOST_H_
#define APK_HOST_H_

#include <stddef.h>
#include <stdint.h>

#include "apk_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct apk_sim apk_sim;

/* Device-memory provider.  NULL => hipMalloc/hipFree.  bench.py / tests pass callbacks
 * that hand out torch CUDA tensors, so that message buffers can be given to
 * torch.distributed (RCCL) without copies.  `tag` names the buffer ("cons", "send:3", ...). */
typedef struct apk_allocator {
  void *user;
  void *(*alloc)(void *user, const char *tag, size_t bytes);
  void (*release)(void *user, void *ptr);
} apk_allocator;

/* Inter-rank operations (one process per GPU).  NULL => single rank.
 *  exchange: all per-peer send buffers are packed and the pack kernels are complete on the
 *            sim's stream when this is called; on return the per-peer receive buffers must
 *            be ready to be read by work enqueued on that stream.
 *  allreduce_min: in-place MIN over ranks of n doubles (dt; mindx/dt_hyp: hydro.cpp:122-128).
 *  allreduce_sum: in-place SUM over ranks (history output).
 *  exchange_begin / exchange_end (optional, both or neither): the same exchange in two halves so
 *            that it can overlap with compute: begin posts the sends/receives (same precondition
 *            as `exchange`) and returns without waiting; end makes work enqueued on the sim's
 *            stream afterwards wait for the receives.  Between the two calls the driver runs the
 *            x1 sweep of the next stage on every cell that does not need the data in flight
 *            (RCCL executes the transfers on its own stream). */
typedef struct apk_comm_ops {
  void *user;
  int (*exchange)(void *user);
  int (*allreduce_min)(void *user, double *vals, int n);
  int (*allreduce_sum)(void *user, double *vals, int n);
  int (*exchange_begin)(void *user);
  int (*exchange_end)(void *user);
} apk_comm_ops;

/* ---- creation from an Athena-style input deck ("<block>" / "key = value") -------------
 * `deck` is the deck TEXT; `overrides` are "block/key=value" strings exactly like the
 * reference's command line (README.md:113-119).  rank/nranks select this process' share
 * of the meshblocks (Morton-ordered contiguous ranges, like Parthenon).
 * On failure returns a negative apk_status and, if errbuf != NULL, a message. */
int apk_sim_create(const char *deck, const char *const *overrides, int noverrides, int rank,
                   int nranks, const apk_allocator *allocator, const apk_comm_ops *comm,
                   apk_stream_t stream, apk_sim **out, char *errbuf, size_t errlen);
void apk_sim_destroy(apk_sim *sim);
const char *apk_sim_last_error(const apk_sim *sim);

/* ---- native RCCL transport (csrc/host/comm_rccl.cpp) -------------------------------------
 * Instead of callbacks the driver can move its messages itself: grouped ncclSend / ncclRecv per
 * peer on a dedicated halo stream (overlapped with compute through events), ncclAllReduce on a second
 * communicator for the per-cycle reductions.  Create the sim with comm = NULL and nranks > 1, then:
 *   rank 0:    apk_rccl_unique_ids(ids, sizeof ids)      two ncclUniqueIds (2 x 128 bytes)
 *   launcher:  broadcast `ids` to all ranks (torch.distributed / MPI / a file)
 *   all ranks: apk_sim_comm_rccl(sim, ids, sizeof ids)    (collective: ncclCommInitRank)
 * before apk_sim_initialize.  One GPU per rank (RCCL refuses two ranks on one device).
 * Replaces Parthenon's boundary communication (hydro_driver.cpp:506, 567-568) and the
 * MPI_Allreduce calls of the package (hydro.cpp:127-128). */
#define APK_RCCL_ID_BYTES 128
int apk_rccl_unique_ids(char *ids, size_t len);
int apk_sim_comm_rccl(apk_sim *sim, const char *ids, size_t len);
/* exchanges posted / reductions done by the native transport so far */
int apk_sim_comm_stats(const apk_sim *sim, long long *exchanges, long long *reductions);
const char *apk_sim_comm_error(const apk_sim *sim);
/* one-rank exercise of the transport on the current device (n doubles sent to self, min / sum
 * reductions); writes "ok" or the failure into msg */
int apk_rccl_selftest(int n, char *msg, size_t len);

/* host-only mode: builds mesh / partition / ghost plans without touching a GPU (used by the
 * CPU tests of the host logic and the world_size-2 gloo tests). */
int apk_sim_create_host_only(const char *deck, const char *const *overrides, int noverrides,
                             int rank, int nranks, apk_sim **out, char *errbuf, size_t errlen);

/* ---- problem setup ------------------------------------------------------------------- */
/* Runs the problem generator selected by <job> problem_id (linear_wave, sod, orszag_tang,
 * synthetic), then ghost exchange -> FillDerived -> first EstimateTimestep (App. A.4). */
int apk_sim_initialize(apk_sim *sim);

/* ---- time integration ---------------------------------------------------------------- */
int apk_sim_step(apk_sim *sim);                      /* one cycle */
int apk_sim_run(apk_sim *sim, int nlim, int *ncycles); /* until tlim or nlim cycles (<0: none) */
double apk_sim_time(const apk_sim *sim);
double apk_sim_dt(const apk_sim *sim);
double apk_sim_tlim(const apk_sim *sim);
double apk_sim_c_h(const apk_sim *sim);
int apk_sim_ncycle(const apk_sim *sim);
long long apk_sim_fofc_count(const apk_sim *sim);
/* 0 = flux-array path (CalculateFluxes + Update + Dedner), 1 = fused stage path */
int apk_sim_set_fused(apk_sim *sim, int fused);
/* 1 (default) = overlap the halo exchange between two stages of a cycle with the x1 sweep of the
 * next stage where possible (remote neighbours, exchange_begin/end given, fused path, >= 2-D, next
 * stage not donor cell); 0 = always exchange synchronously.  Results are identical. */
int apk_sim_set_overlap(apk_sim *sim, int overlap);
/* number of stage boundaries so far at which the exchange was overlapped */
long long apk_sim_overlapped_exchanges(const apk_sim *sim);
/* Direct neighbour addressing (apk_stage_args.face_neighbor): on uniform 3-D meshes whose stages are
 * all single-march donor-cell or two-kernel stages (no passive scalars, floors, extended Dedner source
 * or flux correction) the stage kernels read the interiors of same-rank neighbour
 * blocks directly and the same-rank ghost-zone copies are skipped; ghost zones are brought up to date
 * on demand (every accessor does).  Results are identical.  APK_DIRECT_NEIGHBORS=0 in the environment
 * switches it off.  Returns the number of stage boundaries so far whose same-rank copies were skipped. */
long long apk_sim_skipped_local_exchanges(const apk_sim *sim);
int apk_sim_set_direct_neighbors(apk_sim *sim, int on); /* 1 (default) / 0 = always copy */
/* Refined meshes, VL2 with a high-order corrector in the two-kernel form, default equation-of-state limits, no passive
 * scalars: the corrector derives its input from the half-step conserved state (apk_stage_args.prim_from_cons = 2, its
 * result over the register u1) and the flux correction's boundary planes likewise
 * (apk_calculate_fluxes_boundary_list_from_cons), so that no ConsToPrim pass runs between the two stages.  Results are
 * identical.  apk_sim_set_prim_free(sim, 0) / APK_AMR_PRIM_FREE=0 in the environment switch it off.  Returns the number
 * of passes done without. */
long long apk_sim_amr_c2p_passes_skipped(const apk_sim *sim);
/* Full-step primitives kept out of memory (on by default where it applies: uniform 3-D meshes, VL2 -- a donor-cell
 * predictor followed by a two-kernel stage --, default equation-of-state limits, no passive scalars, no extended Dedner
 * source, no forcing): the last stage of a cycle computes the primitives of the new state for the time-step estimate
 * only (apk_stage_args.fill_derived = 3) and the predictor of the next cycle derives its input from the conserved state
 * (prim_from_cons) -- the reference stores them in FillDerived (hydro_driver.cpp:571-577) and reads them back in
 * CalculateFluxes.  Every accessor materialises them on demand; results are identical.  APK_PRIM_FREE=0 in the
 * environment switches it off.  apk_sim_prim_is_stale: 1 while the primitives of the current state are not in memory.
 * The same switch governs the stage loop of refined meshes (apk_sim_amr_c2p_passes_skipped above). */
int apk_sim_set_prim_free(apk_sim *sim, int on);
/* One-layer exchanges (on by default where they apply: N > 1 ranks, uniform periodic 3-D meshes, VL2, no passive
 * scalars, no extended Dedner source, no forcing): the exchange at the end of a cycle delivers ONE layer of ghost
 * cells -- all the donor-cell predictor of the next cycle reads; the reference exchanges nghost layers after every
 * stage (hydro_driver.cpp:567-568) -- and the corrector's exchange stays a full one.  The messages are a prefix of
 * the same buffers; the transports find the current sizes in apk_sim_peer at every exchange
 * (apk_sim_message_generation tells when they changed).  Accessors that read ghost zones complete them with a full
 * exchange first: a COLLECTIVE then, to be called on every rank, like the accessors of a refined mesh.  Results are
 * identical.  APK_THIN_EXCHANGE=0 in the environment switches it off.  Returns (apk_sim_thin_exchanges) the number
 * of one-layer exchanges so far. */
int apk_sim_set_thin_exchange(apk_sim *sim, int on);
long long apk_sim_thin_exchanges(const apk_sim *sim);
/* x1 strips without pack / unpack kernels (on by default where it applies, uniform periodic 3-D meshes over N > 1 ranks:
 * the VL2 cycles that take one-layer exchanges and store no full-step primitives, and the RK integrators whose stages all
 * derive their input from the conserved state in the two-kernel form): the finishing kernel of a stage
 * stores its x1 boundary columns straight into the send buffers and the kernels of the next stage read their x1 ghost
 * columns straight from the receive buffers (apk_stage_args.x1_halo in apk_amd.h); the pack and unpack plans leave the
 * x1 faces out.  The messages are byte for byte what the pack kernel would have written, so a rank may use the path or
 * not independently of its peers; results are identical.  Accessors that read ghost zones repeat such an exchange in full
 * first (a COLLECTIVE, as after a one-layer exchange).  APK_X1_DIRECT=0 in the environment switches it off.
 * apk_sim_x1_direct_exchanges: the number of exchanges so far whose x1 strips went that way. */
int apk_sim_set_x1_direct(apk_sim *sim, int on);
long long apk_sim_x1_direct_exchanges(const apk_sim *sim);
int apk_sim_prim_is_stale(const apk_sim *sim);
/* forced turbulence in a cycle that stores no primitives: number of kicks that estimated the time step without storing
 * them (apk_turb_apply_dt) so far */
long long apk_sim_turb_dt_kicks(const apk_sim *sim);
/* Refined meshes: the stage loop's exchange leaves out the ghost zones behind block edges and corners
 * (no sweep or flux correction reads them; accessors, tagging and the last exchange of a cycle that
 * checks the refinement criteria complete them).  1 = always exchange in full: same results, for A/B timing and
 * for the tests of that statement. */
int apk_sim_set_amr_full_exchange(apk_sim *sim, int on);

/* ---- introspection ------------------------------------------------------------------- */
typedef struct apk_sim_info {
  int fluid, recon, riemann, integrator;
  int nx[3], mb[3], ng, nhydro, nscalars, ndim;
  int nblocks_total, nblocks_local, first_gid;
  int rank, nranks, npeers;
  int fofc, dedner_extended, fused;
  double cfl, gamma, glmmhd_alpha;
  double xmin[3], xmax[3], dx[3];
  int64_t cells_per_block; /* incl. ghosts */
  int64_t zones_local;     /* interior cells on this rank */
  int64_t zones_total;
} apk_sim_info;
int apk_sim_get_info(const apk_sim *sim, apk_sim_info *info);
/* global block id and logical (bx,by,bz) of local block lb */
int apk_sim_block_location(const apk_sim *sim, int lb, int *gid, int loc[3]);
/* Mesh refinement (parthenon/mesh/refinement = static | adaptive, numlevel, derefine_count,
 * <parthenon/static_refinement#>, <refinement>): refinement level of a local block (0 on uniform
 * meshes); apk_sim_block_location then returns the logical location at that level.  Statistics:
 * blocks refined / sibling groups merged so far, deepest level allowed, zone-cycles done.
 * apk_sim_regrid runs one tag -> refine / derefine -> transfer pass on demand. */
/* hydro/first_order_flux_correct = true: stages with gam0 = 0 run fused and are only redone through
 * the flux-array sequence + FirstOrderFluxCorrect when a cell fails its admissibility test; this counts
 * those fallbacks */
long long apk_sim_fofc_fallback_stages(const apk_sim *s);
int apk_sim_block_level(const apk_sim *s, int lb);
int apk_sim_amr_stats(const apk_sim *s, long long *refined, long long *derefined, int *max_level,
                      long long *zone_cycles);
int apk_sim_regrid(apk_sim *s, int *changed);
/* a regridding pass for given per-block tags (+1 / 0 / -1, one per block of the whole forest, the
 * same array on every rank): forest update with 2:1 balance and derefine_count, new distribution,
 * new plans; on a device sim also the transfer of the state (copy / prolongate / restrict), ghost
 * exchange and ConsToPrim on the new mesh */
int apk_sim_amr_apply_tags(apk_sim *s, const int *tags, int ntags, int *changed);
/* device pointers of local block lb: field 0 = cons, 1 = prim, 2 = u1.cons */
void *apk_sim_block_ptr(const apk_sim *sim, int lb, int field);
/* copy the interior of every local block of `field` into a global-shaped host array
 * [nvar][nx3][nx2][nx1] (only this rank's cells are written) */
int apk_sim_gather(apk_sim *sim, int field, double *host_out);
/* copy one full block (incl. ghosts) host<->device.  Ghost zones are brought up to date first: the stage loop
 * leaves same-rank ghost zones of uniform meshes (direct neighbour addressing) and the ghost zones behind edges
 * and corners of refined meshes unfilled.  On a REFINED mesh distributed over several ranks that completion
 * is a halo exchange, i.e. collective: the first accessor after a cycle must be called on every rank (as the
 * outputs of apk_sim_execute are).  On uniform meshes it is local. */
int apk_sim_read_block(apk_sim *sim, int lb, int field, double *host_out);
int apk_sim_write_block(apk_sim *sim, int lb, int field, const double *host_in);
/* history sums over the whole mesh (allreduce'd): mass,1-mom,2-mom,3-mom,KE,tot-E,ME,relDivB */
int apk_sim_history(apk_sim *sim, double *out8);
/* linear-wave L1 errors vs the initial condition (src/pgen/linear_wave.cpp:183-335) */
/* few-modes turbulence driver (job/problem_id = turbulence).  History: volume sums of Ms, Ma,
 * plasma beta (src/pgen/turbulence.cpp:47-101; divide by the box volume for the means).  The
 * fmft_* calls expose the host spectral state (FewModesFT, src/utils/few_modes_ft.cpp) and also
 * work on a host-only sim: var_hat is [3][num_modes][2]; evolve advances the OU process by dt
 * (consuming RNG draws exactly as a driven step does); phases fills [2][num_modes][n] for cells
 * g0..g0+n-1 of `axis`.  read_acc copies block lb's acceleration field [3][Nk][Nj][Ni]. */
int apk_sim_turbulence_history(apk_sim *sim, double *out3);
/* field_loop's extra history column "UserRelDivB" (src/pgen/field_loop.cpp:60-103), summed over ranks */
int apk_sim_user_reldivb(apk_sim *s, double *out);
int apk_sim_fmft_num_modes(const apk_sim *sim);
int apk_sim_fmft_var_hat(const apk_sim *sim, double *out);
int apk_sim_fmft_evolve(apk_sim *sim, double dt);
int apk_sim_fmft_phases(const apk_sim *sim, int axis, int n, int g0, double *out);
int apk_sim_read_acc(apk_sim *sim, int lb, double *host_out);
/* The refinement criterion configured by the deck's <refinement> block (src/hydro/hydro.cpp:
 * 788-816: type = pressure_gradient | xyvelocity_gradient | maxdensity with their thresholds),
 * evaluated on every local block (pkg->CheckRefinementBlock): tags[lb] = +1 refine, 0 same,
 * -1 derefine; crit (may be NULL) receives the reduced criterion.  The mesh stays uniform. */
int apk_sim_check_refinement(apk_sim *sim, int *tags, double *crit);

/* ---- text outputs in the reference's formats ---------------------------------------------
 * History: one row per call, "time dt cycle nbtotal" followed by the package's history list
 * (src/hydro/hydro.cpp:422-441: mass 1-mom 2-mom 3-mom KE tot-E [ME relDivB]; turbulence adds
 * Ms [Ma plasma_beta], src/pgen/turbulence.cpp:104-116), under the two '#' header lines
 * Parthenon's history writer emits ("[n]=label" columns), so numpy.genfromtxt-based analysis
 * written for the reference reads it unchanged (e.g. tst/regression/test_suites/turbulence/
 * turbulence.py:42-52 takes Ms, Ma from the third- and second-to-last columns).  The header is
 * written when the file does not exist yet.  All ranks must call; rank 0 writes.
 * Error file: src/pgen/linear_wave.cpp:296-334 ("linearwave-errors.dat": header when new,
 * otherwise append; the Nx2-twice quirk of the reference's row is kept). */
int apk_sim_history_labels(const apk_sim *sim, char *buf, size_t len); /* space separated */
int apk_sim_write_history(apk_sim *sim, const char *path);
int apk_sim_write_linear_wave_errors(apk_sim *sim, const char *path);
/* main loop of a deck run (what `athenaPK -i deck` does for the scope here): initialize, step to
 * tlim / nlim, write every <parthenon/output*> block with file_type = hst to
 * <outdir>/<parthenon/job problem_id, default "parthenon">.out<N>.hst at its `dt` cadence (t = 0
 * and the final time included), and, for linear_wave with compute_error, the error file. */
int apk_sim_execute(apk_sim *sim, const char *outdir, int *ncycles);
/* wall seconds of the main loop of the last apk_sim_execute (initialisation excluded, device
 * synchronised): zones * cycles / this = Parthenon's "zone-cycles/wallsecond" */
double apk_sim_loop_seconds(const apk_sim *sim);
/* cycles covered by apk_sim_loop_seconds: those after parthenon/time/perf_cycle_offset (default 0) */
int apk_sim_loop_cycles(const apk_sim *sim);
/* interior cells updated inside that timed loop, summed over its cycles (the block count of an
 * adaptive mesh changes from cycle to cycle): zone-cycles / wallsecond = this / loop_seconds */
long long apk_sim_loop_zone_cycles(const apk_sim *sim);
/* circularly polarised Alfven wave (job/problem_id = cpaw, 3-D): L1 errors against the initial state
 * and their RMS (src/pgen/cpaw.cpp:127-186; err8 = d, M1, M2, M3, E, B1, B2, B3), and the
 * reference's "cpaw-errors.dat" row (cpaw.cpp:188-220). */
int apk_sim_cpaw_errors(apk_sim *sim, double *rms, double *err8);
int apk_sim_write_cpaw_errors(apk_sim *sim, const char *path);
int apk_sim_linear_wave_errors(apk_sim *sim, double *rms, double *l1_5, double *max_5);
/* MHD linear waves (job/problem_id = linear_wave_mhd, src/pgen/linear_wave_mhd.cpp: the seven wave families of
 * adiabatic MHD on a magnetised background, wave_flag 0..6 = fast-, Alfven-, slow-, entropy, slow+, Alfven+, fast+;
 * B = curl A of the cell-centred potential :370-441): L1 / max errors of d, M1, M2, M3, E, B1, B2, B3 against the
 * analytic wave and their RMS (:177-276).  apk_sim_write_linear_wave_errors writes this problem's 8-column row. */
int apk_sim_linear_wave_mhd_errors(apk_sim *sim, double *rms, double *l1_8, double *max_8);
/* individual driver steps, exposed for tests */
int apk_sim_exchange_ghosts(apk_sim *sim);
int apk_sim_fill_derived(apk_sim *sim);
int apk_sim_estimate_timestep(apk_sim *sim, double *dt);
/* after replacing the state through apk_sim_write_block (+ exchange_ghosts + fill_derived): derive the time step
 * as apk_sim_initialize does after the problem generator (no growth limit from earlier steps) */
int apk_sim_reset_time_step(apk_sim *sim);

/* kernel timing of the sim's hot-path handle (apk_kernel_timing_* of apk_amd.h) */
int apk_sim_kernel_timing_enable(apk_sim *sim, int on);
int apk_sim_kernel_timing_read(apk_sim *sim, int slot, double *total_ms, long long *launches);

/* ---- ghost-exchange plan introspection (host logic; valid in host-only mode) ----------- */
typedef struct apk_peer_info {
  int rank;
  int64_t send_count, recv_count; /* doubles */
  void *send_buf, *recv_buf;      /* device pointers (NULL in host-only mode) */
} apk_peer_info;
int apk_sim_peer(const apk_sim *sim, int p, apk_peer_info *info);
/* The message set the next comm_ops.exchange moves.  On uniform meshes it never changes (the halo
 * buffers; apk_sim_info.npeers entries).  On refined meshes the driver makes the halo, the
 * flux-correction or the regridding messages current before it calls exchange, and regridding
 * changes peers, sizes and buffers: an exchange callback re-reads the set whenever
 * apk_sim_message_generation has changed since it last looked. */
int apk_sim_num_peers(const apk_sim *sim);
/* introspection: report the halo (1) / flux-correction (2) message set of a refined mesh through
 * apk_sim_peer (0 = the uniform mesh's set); 3 / 4: the halo sets of the faces-only and of the shell
 * exchange of the stage loop; 5 = the one-layer set of a uniform mesh (apk_sim_set_thin_exchange) */
int apk_sim_select_messages(apk_sim *sim, int which);
long long apk_sim_message_generation(const apk_sim *sim);
/* number of box copies in each phase: 0 local, 1 pack, 2 unpack, 3..5 physical BC x1..x3, 6 / 7 pack / unpack of the
 * one-layer exchange; on
 * refined meshes 10 = all copies of the multilevel exchange, 11..13 = coarse-buffer boundaries,
 * 14..16 = block boundaries, 17..19 = flux-correction copies x1..x3 (10..19: the global plan
 * every rank builds, global block numbers); this rank's share with local block numbers and message
 * buffers: 20 fill copies, 21 packs, 22 unpacks, 25..27 / 28..30 / 31..33 flux-correction copies /
 * packs / unpacks x1..x3, 34..36 coarse-buffer boundaries, 37..39 block boundaries; the exchanges of
 * the stage loop: 40..42 fill copies / packs / unpacks without the boxes behind edges and corners, 43
 * the copies of 40 without those between same-rank blocks of one level (the stages read these
 * neighbours through apk_stage_args.face_neighbor), 44..46 / 47..49 fill copies, packs, unpacks /
 * block boundaries of the exchange that fills every ghost zone two layers deep only (before a
 * refinement check) */
int apk_sim_plan_size(const apk_sim *sim, int phase);
/* region r of a phase, with src/dst expressed as (kind, block, element offset):
 * kind 0 = local block cons, 1 = send buffer of peer `block`, 2 = recv buffer of peer `block`,
 * 3 = coarse buffer of the block, 4..6 = its x1..x3 flux array */
typedef struct apk_region_info {
  int src_kind, src_block, dst_kind, dst_block;
  int64_t src_off, dst_off;
  int ext[3], nvar, flip_var;
  int64_t src_stride[4], dst_stride[4];
} apk_region_info;
int apk_sim_plan_region(const apk_sim *sim, int phase, int r, apk_region_info *info);
/* operator lists of the multilevel exchange, in execution order restrict-own (which = 0) ->
 * phase 10 -> 11..13 -> prolongate (1) -> 14..16; flux correction: per direction d, which = 2 + d
 * then phase 17 + d.  which + 10 = this rank's share (local block numbers); 15 / 16: this rank's
 * prolongations of the faces-only and of the shell exchange.  kind is an APK_RO_* value; index boxes in
 * coarse-buffer indices. */
typedef struct apk_amr_op_info {
  int kind, level;
  int src_kind, src_block, dst_kind, dst_block;
  int lo[3], hi[3];
  double xmin[3], dx[3]; /* lower interior corner and cell widths of the block the operator works on */
  int cng;
  int64_t coarse_doubles; /* allocation of one coarse buffer */
} apk_amr_op_info;
int apk_sim_amr_ops_size(const apk_sim *sim, int which);
int apk_sim_amr_op(const apk_sim *sim, int which, int n, apk_amr_op_info *info);

#ifdef __cplusplus
}
#endif
#endif /* APK_HOST_H_ */
