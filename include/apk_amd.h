/*
 * apk_amd.h -- C-ABI of the MI355X-native AthenaPK hot path (libapk_amd.so).
 *
 * Drop-in boundary: these entry points are what a Parthenon/AthenaPK build binds in place
 * of the Kokkos task functions of the flux-divergence update.  Every entry point cites
 * the reference interface (file:line under the AthenaPK tree) it replaces; the adapter a
 * maintainer adds on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; device memory is BORROWED for the duration of a call
 *    (Parthenon owns all field storage; src/hydro/hydro.cpp:777-786).
 *  - every call is asynchronous on the caller's HIP stream (`apk_stream_t` = hipStream_t
 *    cast to void*; NULL = default stream) unless it returns a host scalar.
 *  - return value: APK_OK (0) or a negative APK_ERR_* code.  Never throws, never aborts.
 *    Device-side "PARTHENON_REQUIRE" conditions (negative density / pressure in
 *    ConsToPrim, src/eos/adiabatic_hydro.hpp:77-79,111-113) are latched in a flag word
 *    read back with apk_poll_device_flags().
 *  - block field layout: [nvar][Nk][Nj][Ni] doubles, i fastest, Ni = nx1 + 2*ng (no
 *    ghosts in a collapsed dimension); flux(d, v, k, j, i) is the flux through the LOWER
 *    d-face of cell (k,j,i)  (SURVEY.md 8, App. A.5).
 *  - variable order: cons (rho, m1, m2, m3, E [,B1,B2,B3,psi], scalars...),
 *    prim (rho, v1, v2, v3, p [,B1,B2,B3,psi], scalars...)   src/main.hpp:19-33
 */
#ifndef APK_AMD_H_
#define APK_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APK_AMD_VERSION 1

/* ---- option enums: numeric values = position in the reference's enum classes,
 *      src/main.hpp:35-38 ---- */
enum apk_riemann { APK_RS_UNDEFINED = 0, APK_RS_NONE = 1, APK_RS_HLLE = 2, APK_RS_LLF = 3,
                   APK_RS_HLLC = 4, APK_RS_HLLD = 5 };
enum apk_recon { APK_RC_UNDEFINED = 0, APK_RC_DC = 1, APK_RC_PLM = 2, APK_RC_PPM = 3,
                 APK_RC_WENOZ = 4, APK_RC_WENO3 = 5, APK_RC_LIMO3 = 6 };
enum apk_integrator { APK_INT_UNDEFINED = 0, APK_INT_RK1 = 1, APK_INT_RK2 = 2,
                      APK_INT_VL2 = 3, APK_INT_RK3 = 4 };
enum apk_fluid { APK_FLUID_UNDEFINED = 0, APK_FLUID_EULER = 1, APK_FLUID_GLMMHD = 2 };

enum apk_status {
  APK_OK = 0,
  APK_ERR_INVALID = -1,      /* bad argument / inconsistent pack */
  APK_ERR_UNSUPPORTED = -2,  /* (fluid,recon,riemann) not in the registry, hydro.cpp:386-416 */
  APK_ERR_NGHOST = -3,       /* too few ghost zones for the reconstruction, hydro.cpp:444-447 */
  APK_ERR_DEVICE = -4,       /* HIP runtime error (see apk_last_error) */
  APK_ERR_NO_DEVICE = -5     /* no HIP device / extension not usable: fail loudly */
};

/* device flag bits latched by kernels (apk_poll_device_flags) */
#define APK_FLAG_NEG_DENSITY 1u
#define APK_FLAG_NEG_PRESSURE 2u

typedef void *apk_stream_t; /* hipStream_t */

/* AdiabaticHydroEOS / AdiabaticGLMMHDEOS parameters: src/eos/eos.hpp:33-61,
 * defaults src/hydro/hydro.cpp:507-537 (floors <= 0 disabled, ceilings +inf disabled) */
typedef struct apk_eos {
  double gamma;
  double pfloor, dfloor, efloor;
  double vceil, eceil;
} apk_eos;

/* what CalculateFluxes is templated on: src/hydro/hydro.hpp:43-46,53 (FluxFunKey_t) */
typedef struct apk_flux_cfg {
  int fluid;   /* apk_fluid */
  int recon;   /* apk_recon */
  int riemann; /* apk_riemann */
} apk_flux_cfg;

/* one meshblock of a MeshBlockPack: what `pack(b)(v,k,j,i)`, `pack(b).flux(d,v,k,j,i)`
 * and `pack.GetCoords(b).Dxc<d>()` resolve to (src/hydro/hydro.cpp:1041-1073).
 * All pointers are DEVICE pointers; flux[d] may be NULL if the caller only uses the fused
 * stage path (or for inactive dimensions). */
typedef struct apk_block_desc {
  double *cons;
  double *prim;
  double *flux[3];
  double dx[3];
} apk_block_desc;

/* MeshData / MeshBlockPack: src/hydro/hydro.cpp:1026-1063 */
typedef struct apk_pack_desc {
  int nblocks;  /* cons_in.GetDim(5) */
  int nhydro;   /* pkg->Param<int>("nhydro"): 5 or 9 */
  int nscalars; /* pkg->Param<int>("nscalars") */
  int nx[3];    /* block_size.nx(X1DIR..X3DIR); 1 = collapsed dimension */
  int ng;       /* parthenon/mesh/nghost */
  const apk_block_desc *blocks; /* HOST array [nblocks] */
  /* Element strides of the j, k and variable index of every array of the pack (cons, prim, flux): `pack(b)(v,k,j,i)` is
   * base[v * stride[2] + k * stride[1] + j * stride[0] + i] -- the reference's accessor is stride-agnostic
   * (src/hydro/hydro.cpp:1041-1073), Parthenon's allocations are LayoutRight.  0 = that natural layout:
   * Ni, Ni * Nj, Ni * Nj * Nk with Ni = nx[0] + 2 ng etc.  Anything else must satisfy stride[0] >= Ni,
   * stride[1] >= stride[0] * Nj, stride[2] >= stride[1] * Nk (and be the same for every block).  What it is for: rows
   * at a pitch that is a multiple of a cache line (16 doubles) from a base chosen so that the first INTERIOR cell of a
   * row sits on a line boundary -- the stage kernels then load and store the 128-cell rows of a 128^3 block as 8 whole
   * lines instead of 9 - 10 partial ones (the standalone driver: apk_amd/row_pitch = aligned). */
  int64_t stride[3];
} apk_pack_desc;

typedef struct apk_ctx apk_ctx;   /* workspace: flag words, reduction scratch */
typedef struct apk_pack apk_pack; /* device-resident copy of an apk_pack_desc */

/* ---- lifetime ---------------------------------------------------------------------- */
/* Creates the workspace on the CURRENT HIP device.  Fails with APK_ERR_NO_DEVICE when no
 * gfx950-capable device is visible: there is no CPU fallback. */
int apk_create(apk_ctx **out);
void apk_destroy(apk_ctx *ctx);
const char *apk_last_error(const apk_ctx *ctx);
int apk_version(void);
/* 1 if this library was built with -ffp-contract=off (bit-parity build), else 0 */
int apk_fp_strict(void);
/* Measurement aid of bench.py (no counterpart in the reference): the issue floor of the GLM-MHD PPM + HLLD scheme on this
 * device -- `steps` sweep steps (nine PPM reconstructions + one HLLD solve per lane, nothing else: no global memory, no
 * limiter block entered) in every lane of two waves per SIMD, `reps` launches timed with events on the null stream.
 * A cell-stage of the 3-D scheme is three such steps per cell. */
int apk_bench_scheme_floor(int steps, int reps, double *ms_per_launch, long long *lane_steps_per_launch);

/* Uploads the descriptors (tiny H2D copy, synchronous).  Rebuild after remeshing, like
 * Parthenon rebuilds its packs. */
int apk_pack_create(apk_ctx *ctx, const apk_pack_desc *desc, apk_pack **out);
void apk_pack_destroy(apk_pack *pack);

/* ---- the hot path ------------------------------------------------------------------- */

/* Replaces Hydro::CalculateFluxes<fluid,recon,rsolver>(std::shared_ptr<MeshData<Real>>&)
 * src/hydro/hydro.cpp:1025-1208 (type FluxFun_t, hydro.hpp:43-46; registry
 * hydro.cpp:386-416; (dc,llf) maps to CalculateFluxesTight hydro.cpp:980-1022).
 * Reads prim, writes flux[d] over the reference's loop extents. */
int apk_calculate_fluxes(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg,
                         const apk_eos *eos, double c_h, apk_stream_t stream);
/* The same fluxes on the faces of interior cells only (the loop limits of CalculateFluxesTight,
 * hydro.cpp:1006-1009).  The reference's sweeps also cover one transverse ghost row / plane
 * (hydro.cpp:1031-1039), which neither the flux divergence, the first-order flux correction nor
 * the coarse-fine flux correction consume: 27 % of the faces of a 16^3 meshblock, 56 % of an 8^3. */
int apk_calculate_fluxes_tight(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg,
                               const apk_eos *eos, double c_h, apk_stream_t stream);
/* Only the faces on the boundary of each meshblock (two planes per active direction, interior
 * transverse extent): what the coarse-fine flux correction needs when the stage itself ran fused
 * (apk_stage_fused never materialises face fluxes; see apk_flux_fix_plan below). */
int apk_calculate_fluxes_boundary(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg,
                                  const apk_eos *eos, double c_h, apk_stream_t stream);
/* The same for a DEVICE list of (block, face) pairs only, in ONE launch (the input of the coarse-fine flux
 * correction, hydro_driver.cpp:527-531, when the stage ran fused): faces[n] = 6 * block + face,
 * face = {x1 lower, x1 upper, x2 lower, x2 upper, x3 lower, x3 upper} -- on a refined mesh the faces
 * with a coarser or finer block behind them (the fine side's fluxes are averaged, the coarse side's own
 * flux is what the average replaces); the planes of unlisted faces are left untouched. */
int apk_calculate_fluxes_boundary_list(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg, const apk_eos *eos,
                                       double c_h, const int *faces, int nfaces, apk_stream_t stream);
/* The same with the stencil cells taken from the CONSERVED state and converted in registers (the ConsToPrim the pass
 * over the blocks applies, adiabatic_hydro.hpp:61-169, without its floor / ceiling blocks: eos without dfloor, efloor,
 * pfloor, vceil, eceil only -- APK_ERR_UNSUPPORTED otherwise; no passive scalars): for cycles of a refined mesh whose
 * stages derive their input from the conserved state (apk_stage_args.prim_from_cons) and store no primitives.
 * cons_delta: offset in doubles from a block's `cons` array of the pack to the array that holds the input state (0: the
 * pack's own -- the stage's u0 --; the distance to u1's array for a stage whose input is u1). */
int apk_calculate_fluxes_boundary_list_from_cons(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg, const apk_eos *eos,
                                                 double c_h, const int *faces, int nfaces, long long cons_delta,
                                                 apk_stream_t stream);

/* Replaces parthenon::Update::UpdateWithFluxDivergence<MeshData<Real>>(u0,u1,gam0,gam1,
 * beta_dt); call site src/hydro/hydro_driver.cpp:534-537.
 * u0.cons <- gam0*u0.cons + gam1*u1.cons + beta_dt * (-div F(u0.flux)) on interior cells. */
int apk_update_with_flux_divergence(apk_ctx *ctx, const apk_pack *u0, const apk_pack *u1,
                                    double gam0, double gam1, double beta_dt,
                                    apk_stream_t stream);

/* Replaces GLMMHD::DednerSource<extended>(MeshData<Real>*, Real beta_dt)
 * src/hydro/glmmhd/dedner_source.cpp:17-75 (called through Hydro::AddUnsplitSources,
 * src/hydro/hydro.cpp:227-246).  c_h, mindx, alpha are the package params of
 * dedner_source.cpp:27-29. */
int apk_dedner_source(apk_ctx *ctx, const apk_pack *md, int extended, double alpha,
                      double c_h, double mindx, double beta_dt, apk_stream_t stream);

/* Fused fast path for one RK stage of one pack = CalculateFluxes -> UpdateWithFlux-
 * Divergence -> DednerSource (hydro_driver.cpp:510-544) without materialising the flux
 * arrays in HBM.  Not usable when first-order flux correction or AMR flux correction needs
 * the face fluxes.  u1 may alias u0 only if gam1 == 0. */
/* x1 strips of a halo exchange that pass through no pack / unpack kernel (apk_stage_args.x1_halo).  The reference
 * fills every ghost zone through boundary buffers (hydro_driver.cpp:506-569 SendBoundBufs / ReceiveBoundBufs / SetBounds);
 * a row of an x1 strip is nghost doubles -- 24 of every 128-byte line a copy kernel fetches from, or writes into, a block.
 * Here the stage kernels go to the buffers themselves:
 *   recv[side]  the buffer segment holding the ghost columns behind the block's lower (0) / upper (1) x1 face: the stage
 *               reads those columns from there instead of the block's ghost zone.  NULL: the ghost zone (or the block
 *               face_neighbor names).
 *   send[side]  the buffer segment the stage ALSO stores the interior columns next to that face into, as it retires
 *               them.  NULL: nothing.
 * A segment is [nvar][nx3][nx2][depth] doubles, `depth` columns in x1 order over the interior extent in x2 and x3 (the
 * ghost columns is-depth..is-1 / ie+1..ie+depth, the interior columns is..is+depth-1 / ie-depth+1..ie): the layout of an
 * x1-face segment of the standalone driver's messages.  recv_depth < nghost: the deeper ghost columns are read from the
 * block as usual.  Needs nx1 >= 2 send_depth.  send_field: 0 = the updated conserved state, 1 = the new primitives
 * (needs fill_derived = 1 or 2).  Honoured by the stage forms apk_stage_x1_halo() reports; others refuse
 * (APK_ERR_UNSUPPORTED). */
typedef struct apk_x1_halo_block {
  const double *recv[2];
  double *send[2];
} apk_x1_halo_block;
typedef struct apk_x1_halo {
  const apk_x1_halo_block *blocks; /* DEVICE array, one entry per block of the pack */
  int recv_depth, send_depth;      /* columns per segment row (0: no segment of that kind is used) */
  int send_field;
} apk_x1_halo;

typedef struct apk_stage_args {
  apk_flux_cfg cfg;
  apk_eos eos;
  double c_h;
  double gam0, gam1, beta_dt; /* integrator->gam0/gam1/beta[stage-1]*dt */
  int dedner;                 /* 0 = off (euler), 1 = plain, 2 = extended */
  double glmmhd_alpha, mindx;
  /* Optional: let the finishing sweep also do the work of the NEXT two tasks on the cells it
   * updates, saving two passes over the interior:
   *  fill_derived = 1: apply ConsToPrim (floors included) and store prim in place
   *                    (Update::FillDerived, hydro_driver.cpp:571-577).  The caller then only
   *                    needs apk_cons_to_prim_ghosts() after the ghost exchange.  Not in 1-D,
   *                    not with dedner == 2 (APK_ERR_UNSUPPORTED).
   *  fill_derived = 2: the same (dedner == 2 allowed), but the new primitives are written into u1's prim arrays
   *                    ("u1.prim", which AthenaPK also carries: hydro_driver.cpp:484-493) and
   *                    u0's are left untouched; the caller swaps the roles of the two prim
   *                    arrays afterwards.  This is what lets a 3-D donor-cell stage (the VL2
   *                    predictor) run as a single march with no flux-difference array at all.
   *  estimate_dt  = 1: (needs fill_derived) also min-reduce dx_d/(|v_d|+c_d) over those cells
   *                    (EstimateHyperbolicTimestep, hydro.cpp:828-896); read it with
   *                    apk_stage_dt_read(). */
  int fill_derived;
  int estimate_dt;
  /* Optional: split the stage around a halo exchange that is still in flight (2-D / 3-D).  Work
   * that does not read the ghost zones still in flight runs first on per-block index windows:
   *  phase = 0   the whole stage (window must be NULL)
   *  phase = 1   partial work restricted per block by `window`, a DEVICE array of 8 ints per block
   *              {i0, rl, ilo, ihi, jlo, jhi, klo, khi}: rows jlo..jhi of planes klo..khi are
   *              flattened with rl columns starting at column i0 and only the cells ilo..ihi of a
   *              row are updated ({0, Ni, is, ie, js, je, ks, ke} is the whole block; rl = 0 skips
   *              the block).  May be called several times with disjoint windows whose union covers
   *              every interior cell exactly once.  An x1-sweep window must start two columns
   *              below ilo and end one above ihi (i0 <= ilo - 2, i0 + rl - 1 >= ihi + 1): the
   *              sweep hands states / PPM interface values from lane to lane.
   *                - stages whose x1 sweep is its own kernel (reconstruction other than dc): the
   *                  x1 sweep only (it reads x1 ghost zones only); j/k ranges must be the full
   *                  interior.  estimate_dt / fill_derived take effect in phase 2.
   *                - 3-D donor-cell stages (one kernel for the whole stage): the whole stage on the
   *                  window; estimate_dt is not available in this mode.
   *  phase = 2   the remaining sweeps (x2 [, x3], update, sources, fill_derived, dt); nothing for
   *              3-D donor-cell stages.
   * window_rl / window_rows: the largest rl and (jhi - jlo + 1) in `window` (size the launch). */
  int phase;
  const int *window;
  int window_rl, window_rows;
  /* trial = 1: the caller may discard this stage's result (the optimistic stage of first-order
   * flux correction): the negative-density / -pressure flags its FillDerived latches go to a
   * separate trial word, which apk_trial_flags() merges into the latched word (keep = 1) or drops
   * (keep = 0).  Flags latched by earlier kernels are never touched by a discarded stage. */
  int trial;
  /* count_unphysical = 1: the finishing sweep also applies FirstOrderFluxCorrect's admissibility
   * test (hydro.cpp:1297-1306: rho <= 0 or E - KE [- ME] <= 0) to every updated cell BEFORE any
   * floor acts (and, with fill_derived, counts cells whose ConsToPrim latched a flag); read the
   * number of failing cells with apk_stage_unphysical_read().  Not with passive scalars. */
  int count_unphysical;
  /* cons_out_delta != 0: the updated conserved state is written to cons + cons_out_delta (in
   * elements, the same offset for every block: a second allocation of identical layout) instead of
   * over u0, which stays intact -- the trial stage of first-order flux correction when gam0 != 0
   * (RK2 / RK3 later stages), whose fallback needs the old u0.  Interior cells only; not with
   * passive scalars. */
  int64_t cons_out_delta;
  /* face_neighbor != NULL: direct neighbour addressing.  A DEVICE array of 6 ints per block of the
   * pack {x1 lower, x1 upper, x2 lower, x2 upper, x3 lower, x3 upper}: the pack index of the block
   * of the same size behind that face whose INTERIOR stands in for this block's ghost zone there,
   * or -1 to read the ghost zone itself (a physical boundary, a neighbour on another rank).  With
   * it the same-rank ghost-zone copies of a uniform mesh -- 11 % of a VL2 cycle on 8 x 128^3 --
   * are not needed at all: the reference fills every ghost zone through boundary buffers
   * (hydro_driver.cpp:506-569 SendBoundBufs / SetBounds), the stage kernels here follow the table.
   * Only the face neighbours are read (the unsplit sweeps never touch edge or corner ghost cells);
   * the ghost zones behind faces with an entry >= 0 are neither read nor written.  3-D only, for the
   * stage forms whose kernels follow the table: donor cell (single march, fill_derived 0 or 2) and
   * the two-kernel stage (apk_stage_split_axis() == 3); not with passive scalars nor dedner = 2
   * (APK_ERR_UNSUPPORTED). */
  const int *face_neighbor;
  /* cons_store: which cells of the updated CONSERVED state the stage must leave in memory.
   *   0  every interior cell (the reference's semantics).
   *   1  only the cells within nghost layers of a face of their meshblock -- what ghost-zone copies, message packing
   *      and physical boundary conditions read.
   *   2  none.
   * For a stage whose conserved result nobody reads in full: the predictor of VL2, whose corrector has gam0 = 0 and
   * takes its fluxes from the predictor's PRIMITIVES (fill_derived = 2) -- the half-step conserved state is then
   * dead but for the strips the exchange reads (1), or altogether when every neighbour is read directly through
   * face_neighbor (2): 72 of the 288 bytes the donor-cell stage moves per cell, and that stage runs at the memory
   * system's rate.  Honoured by the 3-D single-march donor-cell stage in its lean form; every other stage form
   * stores all cells (always a valid reading of 1 and 2).  The cells not stored keep whatever they held. */
  int cons_store;
  /* prim_from_cons = 1: u0.prim does NOT hold the primitives of the stage's input state; the kernel derives them from
   * u1.cons, which must be that state (true in stage 1 of every integrator: u1 = u0 there, hydro_driver.cpp:474-495).
   * The single-march 3-D donor-cell stage and the two-kernel 3-D stage, in their lean forms only (else
   * APK_ERR_UNSUPPORTED).  Together with fill_derived = 3 in the last stage of the previous cycle it removes the
   * full-step primitives from memory altogether: 72 B per cell less to store there and 72 B less to load here, both on
   * kernels that run at the memory system's rate.
   * prim_from_cons = 2 (two-kernel stage): the input state is u0.cons itself -- the stages with gam0 != 0 of RK2 / RK3,
   * whose input is the state they update.  Neighbouring waves read the old values while a wave writes new ones, so the
   * result must go elsewhere: cons_out_delta != 0 is required.  With 1 / 2 in every stage and fill_derived = 0 (3 in the
   * last) an RK integrator keeps no primitives in memory at all.
   * fill_derived = 3 (listed here, with estimate_dt = 1): ConsToPrim of the updated cells for the time-step estimate
   * only; neither u0.prim nor u1.prim is written.  Two-kernel 3-D stage in its lean form only. */
  int prim_from_cons;
  /* x1 strips straight from / into exchange buffers (see apk_x1_halo above), or NULL.  Whole stages only (phase 0 or 2). */
  const apk_x1_halo *x1_halo;
} apk_stage_args;
int apk_stage_fused(apk_ctx *ctx, const apk_pack *u0, const apk_pack *u1,
                    const apk_stage_args *args, apk_stream_t stream);
/* Which ghost zones phase 1 of a split stage does NOT need, i.e. how its windows are laid out:
 *   1  the x1 sweep is its own kernel: column windows, late x1 faces matter (see `phase` above)
 *   3  two-kernel stage (3-D, reconstruction with a stencil, fill_derived 0 or 2, meshblocks at
 *      least 32 cells wide): phase 1 is the x3 sweep, which writes its flux difference and reads x3
 *      ghost zones only -- windows {0, Ni, is, ie, js, je, klo, khi} select whole planes klo..khi,
 *      late x3 faces matter; phase 2 is ONE march doing x1 + x2 and finishing the stage
 *   0  3-D donor-cell stage: one kernel for the whole stage, 3-D index windows
 * The two-kernel form moves 8.6 GB instead of 13.2 GB per PPM + HLLD stage of 8 x 128^3. */
/* number of cells that failed the test of the last apk_stage_fused(count_unphysical = 1);
 * synchronises the stream */
int apk_stage_unphysical_read(apk_ctx *ctx, long long *count, apk_stream_t stream);
int apk_stage_split_axis(const apk_pack *u0, const apk_flux_cfg *cfg, int fill_derived);
/* 1 if a whole-block stage of this scheme with these options follows apk_stage_args.x1_halo: the lean two-row donor-cell
 * march (3-D, fill_derived 0 / 2, nx2 even) and the lean two-kernel stage's finishing march -- from stored primitives or
 * (prim_from_cons != 0) from a conserved state, but not the stages that take the single march (apk_stage_single_march) --
 * without passive scalars.  A caller checks BEFORE it leaves x1 strips out of its pack / unpack plans. */
int apk_stage_x1_halo(const apk_pack *u0, const apk_flux_cfg *cfg, const apk_eos *eos, int fill_derived, int dedner, int prim_from_cons);
/* 1 if a whole-block stage of this scheme in its lean form with prim_from_cons != 0 runs as ONE march (hydro with PLM:
 * x1 by wave shifts, two x2 rows per lane, x3 carried along the march -- no flux-difference array; the three tasks
 * hydro.cpp:1025-1208 + hydro_driver.cpp:534-544 in a single pass over the conserved state), 0 if it takes the two-kernel
 * form.  A split stage (phase != 0) always takes the two kernels: a caller that can choose leaves such a stage whole. */
int apk_stage_single_march(const apk_pack *u0, const apk_flux_cfg *cfg);

/* Replaces EquationOfState::ConservedToPrimitive(MeshData<Real>*) over the ENTIRE block
 * src/eos/adiabatic_hydro.cpp:33-55, adiabatic_glmmhd.cpp:33-56 (pkg->FillDerivedMesh,
 * src/hydro/hydro.cpp:705-713). */
int apk_cons_to_prim(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos,
                     apk_stream_t stream);

/* apk_cons_to_prim() of every cell and, in the same pass, EstimateHyperbolicTimestep (hydro.cpp:828-896) of the
 * interior cells from the primitives it has just computed: the last FillDerived of a cycle and the time-step
 * estimate that follows it (hydro_driver.cpp:571-603) as one kernel.  The minimum goes to the context's stage
 * word: read it (times cfl) with apk_stage_dt_read() / apk_stage_dt_flags_read(), as after
 * apk_stage_fused(estimate_dt = 1).  ghost_depth < 0: every cell; >= 0: the interior and the ghost cells at
 * most that many layers outside it (edges and corners included) -- the shell a shallow ghost exchange has
 * filled; primitives further out are left as they were. */
int apk_cons_to_prim_dt(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, int ghost_depth, apk_stream_t stream);
/* ... leaving out the ghost cells straight behind a face whose entry in face_neighbor (device, [nblocks][6], or NULL) is
 * >= 0: zones an exchange that follows the face table has not filled and no reader of the table visits. */
int apk_cons_to_prim_dt_skip(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, int ghost_depth,
                             const int *face_neighbor, apk_stream_t stream);
/* ... storing only the primitives whose bit is set in store_vars (bit n = primitive n of the reference's order: IDN,
 * IV1, IV2, IV3, IPR [, IB1, IB2, IB3, IPS]; 0: none -- the time-step estimate alone; every bit: apk_cons_to_prim_dt_skip):
 * the end of a cycle whose successor derives its input from the conserved state (apk_stage_args.prim_from_cons) and
 * whose only reader of stored primitives is a refinement criterion -- the pressure for pressure_gradient, the two
 * velocities for xyvelocity_gradient, the density for maxdensity (apk_tag_blocks).  ghost_depth >= 0 (0: the interior).
 * Without floors and ceilings in eos (they write conserved values back that belong with all primitives of a cell:
 * APK_ERR_UNSUPPORTED), packs without passive scalars. */
int apk_cons_to_prim_dt_select(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, int ghost_depth,
                               const int *face_neighbor, unsigned store_vars, apk_stream_t stream);
/* ConsToPrim (Update::FillDerived, hydro_driver.cpp:571-577; src/eos/adiabatic_hydro.cpp:33) of the interior
 * and of the ghost cells straight behind a block FACE only (at most one ghost coordinate): what the unsplit
 * sweeps (hydro.cpp:1025-1199) and the flux correction read.  The refined-mesh stage loop of the standalone
 * driver fills and converts only those (edges and corners are 37 % of the ghost cells of a 16^3 block with
 * nghost = 4); primitives behind edges and corners are left as they were.  (The tagging criteria,
 * refinement/gradient.cpp:33-36, DO read behind edges and corners: a cycle that ends with a refinement check
 * exchanges and converts in full.) */
int apk_cons_to_prim_faces(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, apk_stream_t stream);
/* The same, leaving alone the ghost zone behind every face f of block b with face_neighbor[6 b + f] >= 0
 * (apk_stage_args.face_neighbor: the stages read that neighbour's interior instead, so nobody fills or
 * converts the zone).  face_neighbor: device pointer, [nblocks][6]. */
int apk_cons_to_prim_faces_skip(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, const int *face_neighbor,
                                apk_stream_t stream);
/* apk_cons_to_prim_faces() / apk_cons_to_prim_faces_skip() (face_neighbor may be NULL) with the time-step
 * estimate of the interior cells reduced on the way, as apk_cons_to_prim_dt() does: the last stage of a
 * refined-mesh cycle that is not followed by a refinement check. */
int apk_cons_to_prim_faces_dt(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, const int *face_neighbor,
                              apk_stream_t stream);
/* ConsToPrim restricted to the ghost zones of every block (interior cells untouched): the
 * companion of apk_stage_fused(fill_derived = 1). */
int apk_cons_to_prim_ghosts(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos,
                            apk_stream_t stream);
/* The same in two parts around a halo exchange that is still in flight (see
 * apk_stage_args.phase).  late_regions: DEVICE array, one word per block; bit
 * (sx+1) + 3 (sy+1) + 9 (sz+1) set when the ghost region towards the neighbour at block offset
 * (sx, sy, sz) is only filled when the exchange completes (neighbour on another rank, physical
 * boundary).  part = 1 converts the ghost cells of the other regions (same-rank copies, ready
 * early), part = 2 those. */
int apk_cons_to_prim_ghosts_split(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos,
                                  const unsigned *late_regions, int part, apk_stream_t stream);

/* Result of the last apk_stage_fused(estimate_dt = 1) on this context: *dt_out = cfl * min.
 * Synchronises `stream`. */
int apk_stage_dt_read(apk_ctx *ctx, double cfl, double *dt_out, apk_stream_t stream);
/* apk_stage_dt_read + apk_poll_device_flags with ONE synchronisation (the per-cycle host round
 * trip of a driver: new dt and the latched negative-density / -pressure flags).  CONSUMES the reduction: the device word
 * goes back to +max for the next one (on every platform); the handle remembers the minimum, and repeated calls -- or an
 * apk_stage_dt_read after it -- return the same value until the next stage / ConsToPrim with a time-step estimate starts
 * a new reduction. */
int apk_stage_dt_flags_read(apk_ctx *ctx, double cfl, double *dt_out, unsigned *flags,
                            apk_stream_t stream);

/* Replaces Hydro::EstimateHyperbolicTimestep<fluid>(MeshData<Real>*)
 * src/hydro/hydro.cpp:828-910.  Synchronises `stream`; *dt_out = cfl * min(...). */
int apk_estimate_timestep(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos,
                          double cfl, double *dt_out, apk_stream_t stream);

/* Replaces Hydro::FirstOrderFluxCorrect<fluid>(u0,u1,gam0,gam1,beta_dt)
 * src/hydro/hydro.cpp:1223-1342.  Iterates <= 4 attempts, one host sync per attempt like
 * the reference's parallel_reduce.  *num_corrected (optional) = cells corrected. */
int apk_first_order_flux_correct(apk_ctx *ctx, const apk_pack *u0, const apk_pack *u1,
                                 int fluid, const apk_eos *eos, double c_h, double gam0,
                                 double gam1, double beta_dt, long long *num_corrected,
                                 apk_stream_t stream);
/* The admissibility test of FirstOrderFluxCorrect (hydro.cpp:1297-1306: density > 0 and
 * E - kinetic [- magnetic] energy > 0) on the conserved state of the pack's interior cells; count =
 * cells that fail it.  Lets a caller run apk_stage_fused optimistically where the reference enables
 * first_order_flux_correct and fall back to the flux-array sequence only when a cell needs it
 * (stages with gam0 = 0, whose inputs the fused stage leaves intact).  Synchronises. */
int apk_count_unphysical(apk_ctx *ctx, const apk_pack *md, int fluid, long long *count,
                         apk_stream_t stream);

/* Replaces the HydroHst<...> reductions src/hydro/hydro.cpp:145-208:
 * out[8] = mass, 1-mom, 2-mom, 3-mom, KE, tot-E, ME, relDivB.  Synchronises `stream`. */
int apk_history(apk_ctx *ctx, const apk_pack *md, int fluid, double *out8,
                apk_stream_t stream);

/* Reads-and-clears the device flag word (APK_FLAG_*).  Synchronises `stream`. */
int apk_poll_device_flags(apk_ctx *ctx, unsigned *flags, apk_stream_t stream);
/* commit (keep = 1) or drop (keep = 0) the flags of stages run with apk_stage_args.trial = 1;
 * asynchronous on `stream`.  Mirrors the reference, where the FOFC trial update
 * (hydro.cpp:1283-1306) never reaches ConservedToPrimitive unless it is accepted. */
int apk_trial_flags(apk_ctx *ctx, int keep, apk_stream_t stream);

/* ---- ghost zones (the step either side of the path; Parthenon bvals, call sites
 *      src/hydro/hydro_driver.cpp:506,567-568) ------------------------------------------ */
/* One strided box copy: dst(v, k, j, i) = sign_v * src(v, k', j', i').  With a zero source
 * stride it broadcasts (outflow); with a negative stride it mirrors (reflecting,
 * src/bvals/boundary_conditions_apk.hpp:38-85).  Used for same-rank neighbour copies,
 * packing into / unpacking from contiguous message buffers, and physical boundaries. */
typedef struct apk_copy_region {
  const double *src;
  double *dst;
  int ext[3];             /* box extent (ni, nj, nk) */
  int nvar;
  int64_t src_stride[4];  /* element strides for (i, j, k, v) */
  int64_t dst_stride[4];
  int flip_var;           /* variable index whose sign is flipped, or -1 */
} apk_copy_region;
typedef struct apk_copy_plan apk_copy_plan;
int apk_copy_plan_create(apk_ctx *ctx, const apk_copy_region *regions, int n,
                         apk_copy_plan **out);
void apk_copy_plan_destroy(apk_copy_plan *plan);
int apk_copy_plan_run(apk_ctx *ctx, const apk_copy_plan *plan, apk_stream_t stream);
/* The same copy for plans whose destinations are ghost zones of cons arrays, with
 * ConservedToPrimitive of every destination cell fused in: the primitives are written at
 * dst + prim_delta (in doubles; the distance from a block's cons array to its prim array, the
 * same for all regions of the plan).  Saves the separate ghost-zone ConsToPrim pass.  Refused
 * (APK_ERR_UNSUPPORTED) when a floor or ceiling of the EOS is active: those write cons back and
 * must keep acting after all boundary phases, as in the reference.  latch_flags = 0 keeps the
 * negative-density / -pressure flags quiet: a physical-boundary phase that is followed by
 * another one copies corner cells whose sources are only filled by that later phase. */
int apk_copy_plan_run_c2p(apk_ctx *ctx, const apk_copy_plan *plan, int fluid, const apk_eos *eos,
                          int64_t prim_delta, int latch_flags, apk_stream_t stream);
/* The same, storing the primitives ONLY: for ghost zones whose conserved values nothing reads -- the half-step
 * state of VL2, whose corrector (gam0 = 0) takes its fluxes from the primitives and updates the full-step state
 * (apk_stage_args.cons_store).  `dst` still names the cons array of the destination block (the primitives go to
 * dst + prim_delta); nothing is written at dst itself. */
int apk_copy_plan_run_c2p_prim_only(apk_ctx *ctx, const apk_copy_plan *plan, int fluid, const apk_eos *eos,
                                    int64_t prim_delta, int latch_flags, apk_stream_t stream);

/* ---- few-modes turbulence driver (BASELINE config 4 forcing; "next" row of SURVEY 8(f)) -------
 * The device side of turbulence::Driving (src/pgen/turbulence.cpp:373-482), which AthenaPK
 * enrols as Hydro::ProblemSourceFirstOrder (src/main.cpp:115, run after the last stage,
 * src/hydro/hydro_driver.cpp:559-560), and of FewModesFT::Generate's inverse transform
 * (src/utils/few_modes_ft.cpp:322-347).  The spectral state (3 x num_modes complex numbers),
 * its host RNG and the Ornstein-Uhlenbeck update stay on the host, as in the reference. */
typedef struct apk_fmft_block {
  double *acc;            /* [3][Nk][Nj][Ni] acceleration field ("acc", turbulence.cpp:119-127) */
  const double *phases_i; /* [2][num_modes][nx1]: (re|im, mode, cell), the shape of the reference's
                             "<prefix>_phases_i" variable (few_modes_ft.cpp:60-62, 143-160) */
  const double *phases_j; /* [2][num_modes][nx2] */
  const double *phases_k; /* [2][num_modes][nx3] */
} apk_fmft_block;
typedef struct apk_fmft apk_fmft;
int apk_fmft_create(apk_ctx *ctx, const apk_fmft_block *blocks /* host array */, int nblocks,
                    int num_modes, apk_fmft **out);
void apk_fmft_destroy(apk_fmft *f);
/* acc(b, n, k, j, i) = sum_m 2 (Re var_hat(n,m) Re phase - Im var_hat(n,m) Im phase) on the
 * interior; var_hat_host is [3][num_modes][2], copied to the device on `stream`: keep it valid
 * until the stream has been synchronised (apk_turb_mean_momentum, the next step, does) */
int apk_fmft_inverse(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, const double *var_hat_host,
                     apk_stream_t stream);
/* turbulence::Perturb in three device steps around the reference's two MPI_Allreduce's
 * (turbulence.cpp:395-436): sums4 = (mass, rho*acc_1..3) * cell volume  [synchronises];
 * subtract sums4[n+1]/sums4[0] from acc_n and return sum acc^2 * volume    [synchronises];
 * scale acc by norm and kick momentum / energy with dt (turbulence.cpp:446-469). */
int apk_turb_mean_momentum(apk_ctx *ctx, const apk_pack *md, const apk_fmft *f, double *sums4,
                           apk_stream_t stream);
int apk_turb_remove_mean(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, const double *sums4,
                         double *ampl_sum, apk_stream_t stream);
int apk_turb_apply(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, double norm, double dt,
                   apk_stream_t stream);
/* apk_turb_apply plus the two tasks that follow the kick on the same cells (hydro_driver.cpp:559-577, 589-603):
 * FillDerived -- ConsToPrim with the floors of `eos`, prim replaced in place -- and, with estimate_dt, the
 * min-reduction of dx_d / (|v_d| + c_d) into the stage's word (read it with apk_stage_dt_read); the caller
 * then converts ghost zones only (apk_copy_plan_run_c2p / apk_cons_to_prim_ghosts).  One pass over cons
 * instead of three. */
int apk_turb_apply_fill(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, double norm, double dt, int fluid,
                        const apk_eos *eos, int estimate_dt, apk_stream_t stream);
/* The same with the time-step estimate and WITHOUT storing the primitives: for cycles whose next stage derives its
 * input from the conserved state (apk_stage_args.prim_from_cons).  No passive scalars. */
int apk_turb_apply_dt(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, double norm, double dt, int fluid,
                      const apk_eos *eos, apk_stream_t stream);
/* TurbulenceHst<Ms|Ma|pb> (turbulence.cpp:47-101): out3 = volume sums of sonic Mach number,
 * Alfvenic Mach number, plasma beta.  Synchronises. */
int apk_turbulence_history(apk_ctx *ctx, const apk_pack *md, int fluid, double gamma, double *out3,
                           apk_stream_t stream);

/* ---- mesh-refinement operators and block tagging (SURVEY 8(f) rank 3 building blocks) -------
 * The operators AthenaPK registers for `cons` (src/hydro/hydro.cpp:780-781):
 * Hydro::refinement_ops::ProlongateCellMinModMultiD (src/hydro/prolongation/custom_ops.hpp:49-186)
 * and Parthenon's RestrictAverage for cells and, for the coarse-fine flux correction
 * (src/hydro/hydro_driver.cpp:527-531), for face fluxes.  A plan is a list of index boxes over
 * many meshblocks of one shape, executed in ONE launch (AMR meshes have many 16^3 blocks: a launch
 * per block and per buffer would be latency bound).  Fine arrays are [nvar][Nk][Nj][Ni] with `ng`
 * ghosts, coarse buffers [nvar][cNk][cNj][cNi] with nx/2 interior cells and `cng` ghosts per active
 * dimension; face arrays have one more entry along their own direction.  Index boxes are inclusive
 * and given in COARSE indices; prolongation reads one coarse cell beyond the box. */
typedef struct apk_refine_geom {
  int nx[3];    /* fine interior cells of a meshblock (1 in a collapsed dimension) */
  int ng;       /* fine ghost cells */
  int cng;      /* coarse-buffer ghost cells */
  double dx[3]; /* fine cell widths */
} apk_refine_geom;
enum {
  APK_RO_PROLONGATE = 0,    /* src = coarse buffer, dst = fine array */
  APK_RO_RESTRICT_CELL = 1, /* src = fine array,   dst = coarse buffer */
  APK_RO_RESTRICT_FACE1 = 2,
  APK_RO_RESTRICT_FACE2 = 3,
  APK_RO_RESTRICT_FACE3 = 4,
  /* the same area average for fluxes stored the way apk_block_desc.flux stores them (cell-shaped
   * arrays, face i = lower face of cell i), into a cell-shaped coarse buffer: the coarse-fine flux
   * correction of the standalone driver */
  APK_RO_RESTRICT_FLUX1 = 5,
  APK_RO_RESTRICT_FLUX2 = 6,
  APK_RO_RESTRICT_FLUX3 = 7
};
typedef struct apk_refine_op {
  int kind;
  const double *src; /* device */
  double *dst;       /* device */
  int lo[3], hi[3];
  double xmin[3]; /* lower interior corner of the block (enters the slope spacings as in the
                     reference, which differences cell-centre coordinates) */
  double dx[3];   /* fine cell widths of THIS box; all zero = the plan's apk_refine_geom.dx.  Lets one
                     plan (one launch) hold boxes of several refinement levels */
} apk_refine_op;
typedef struct apk_refine_plan apk_refine_plan;
int apk_refine_plan_create(apk_ctx *ctx, const apk_refine_geom *geom, int nvar,
                           const apk_refine_op *ops /* host */, int nops, apk_refine_plan **out);
void apk_refine_plan_destroy(apk_refine_plan *p);
int apk_refine_plan_run(apk_ctx *ctx, const apk_refine_plan *p, apk_stream_t stream);

/* Coarse-fine flux correction AFTER a fused stage (hydro_driver.cpp:527-537 puts it before the flux
 * divergence; the fused stage has already applied the coarse block's own face flux F, so the cells
 * next to a coarse-fine face are corrected by the difference): for every element of a region
 *   cons += beta_dt * scale * (fine_avg - coarse_flux)   [times psi_factor for variable psi_var]
 * with scale = +1/dx for the lower face of the cell, -1/dx for its upper face; fine_avg is the
 * area average of the fine blocks' fluxes (an APK_RO_RESTRICT_FLUX* result or a message buffer),
 * coarse_flux the coarse block's flux on the same face (apk_calculate_fluxes_boundary).  psi_factor
 * is the Dedner damping exp(-alpha c_h beta_dt / mindx) the stage applied after its update.
 * average = 0: fine_avg holds the averages, one per element.  average = d + 1 (1..3): fine_avg is the
 * fine block's own cell-shaped x_d flux array at the first fine face under the region (src_stride =
 * twice that array's strides), and each element averages the 2 (2-D) or 4 (3-D) fine faces under its
 * coarse face at fine_stride[], weighted with fine_area -- the arithmetic of APK_RO_RESTRICT_FLUX1 + d,
 * bit for bit, without the restricted plane in between (same-rank faces). */
typedef struct apk_flux_fix_region {
  const double *fine_avg;    /* strides src_stride */
  const double *coarse_flux; /* strides dst_stride */
  double *cons;              /* strides dst_stride */
  int ext[3];
  int nvar;
  int64_t src_stride[4], dst_stride[4];
  double scale;
  int average, ndim;
  int64_t fine_stride[3];
  double fine_area;
} apk_flux_fix_region;
typedef struct apk_flux_fix_plan apk_flux_fix_plan;
int apk_flux_fix_plan_create(apk_ctx *ctx, const apk_flux_fix_region *regions /* host */, int n,
                             apk_flux_fix_plan **out);
/* The regions of up to three directions as ONE plan that runs in ONE launch: regions = direction 0's, then 1's, then
 * 2's (n_by_dir).  A coarse cell on an edge of its block may lie next to two or three coarse-fine faces; the plans per
 * direction correct it direction by direction, and so does this one: the region of the lowest direction that holds a
 * cell applies the terms of all of them in that order (same additions, same order: bit-identical to the three plans
 * run one after the other).  field_base / block_elems: the cell-shaped field [block][var][k][j][i] the regions' cons
 * pointers point into and its elements per block.  APK_ERR_UNSUPPORTED if a region shares cells with more than 8
 * others or is not laid out that way (the caller keeps one plan per direction). */
int apk_flux_fix_plan_create_merged(apk_ctx *ctx, const apk_flux_fix_region *regions /* host */, const int n_by_dir[3],
                                    const double *field_base, int64_t block_elems, apk_flux_fix_plan **out);
void apk_flux_fix_plan_destroy(apk_flux_fix_plan *p);
int apk_flux_fix_plan_run(apk_ctx *ctx, const apk_flux_fix_plan *p, double beta_dt, int psi_var,
                          double psi_factor, apk_stream_t stream);

/* Block tagging, one launch for the whole pack: refinement::gradient::PressureGradient
 * (src/refinement/gradient.cpp:18-61: refine above p0, derefine below 0.25 p0),
 * VelocityGradient (:64-96: p0, 0.5 p0), refinement::other::MaxDensity (src/refinement/other.cpp:
 * 18-44: refine above p0, derefine below p1).  tags[b] = +1 refine / 0 same / -1 derefine,
 * crit[b] (may be NULL) = the reduced criterion.  Synchronises. */
enum { APK_TAG_PRESSURE_GRADIENT = 0, APK_TAG_VELOCITY_GRADIENT = 1, APK_TAG_MAX_DENSITY = 2 };
int apk_tag_blocks(apk_ctx *ctx, const apk_pack *md, int criterion, double p0, double p1, int *tags,
                   double *crit, apk_stream_t stream);
/* The same in two halves, so that the tags share a host round trip with whatever else the caller
 * reads back at the end of a cycle (the driver enqueues the time-step reduction between the two):
 * _begin launches the reduction and the read-back into pinned memory without waiting; _end waits
 * for the stream and converts (pending = what _begin returned). */
int apk_tag_blocks_begin(apk_ctx *ctx, const apk_pack *md, int criterion, int *pending, apk_stream_t stream);
/* ... with apk_stage_args.face_neighbor's table (device, [nblocks][6]): ghost cells straight behind a face whose entry is
 * >= 0 are read from that block's interior, so the exchange in front of the check may leave those zones out (edges and
 * corners are read from the block's own ghost zones as before).  Same criteria, bit for bit. */
int apk_tag_blocks_begin_skip(apk_ctx *ctx, const apk_pack *md, int criterion, const int *face_neighbor, int *pending,
                              apk_stream_t stream);
/* The pressure-gradient criterion (gradient.cpp:18-61) AND the hyperbolic time-step estimate (hydro.cpp:845-895) in one
 * pass over the CONSERVED state: the end of a cycle whose stage loop stores no primitives (apk_stage_args.prim_from_cons).
 * What apk_cons_to_prim_dt_select(pressure, two layers) followed by apk_tag_blocks_begin_skip(APK_TAG_PRESSURE_GRADIENT)
 * computes, bit for bit, with no pressure array written and read back: the minimum goes to the stage's time-step word
 * (apk_stage_dt_read / apk_stage_dt_flags_read), the maxima are read by apk_tag_blocks_end(APK_TAG_PRESSURE_GRADIENT,
 * pending).  3-D packs with at least two ghost layers whose blocks are narrow enough for a few pressure planes in LDS
 * (16^3 .. 64^3), no passive scalars, eos without floors and ceilings: APK_ERR_UNSUPPORTED otherwise. */
int apk_tag_blocks_dt_from_cons(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, const int *face_neighbor,
                                int *pending, apk_stream_t stream);
int apk_tag_blocks_end(apk_ctx *ctx, int nblocks, int criterion, int pending, double p0, double p1, int *tags,
                       double *crit, apk_stream_t stream);

/* field_loop::RelDivBHst (src/pgen/field_loop.cpp:60-95), the "UserRelDivB" history column of the
 * field-loop problem: sum of 0.5 |dx| |div B| / B0 * volume, B0 fixed.  Synchronises. */
int apk_history_user_reldivb(apk_ctx *ctx, const apk_pack *md, double B0, double *out,
                             apk_stream_t stream);

/* ---- in-library kernel timing (HIP events on the caller's stream) ------------------------
 * bench.py needs the average duration of individual kernels measured live on the stream
 * they are launched on.  When enabled, every kernel launch of the listed groups is
 * bracketed by hipEventRecord; apk_kernel_timing_read() synchronises, accumulates and
 * resets.  Off by default (zero overhead). */
enum apk_timing_slot {
  APK_T_FUSED_X1 = 0, /* fused x1 sweep            */
  APK_T_FUSED_X2 = 1, /* fused x2 march            */
  APK_T_FUSED_X3 = 2, /* fused x3 march            */
  APK_T_FLUXES = 3,   /* flux-array sweeps (all directions of one call) */
  APK_T_UPDATE = 4,   /* UpdateWithFluxDivergence  */
  APK_T_DEDNER = 5,   /* DednerSource              */
  APK_T_C2P = 6,      /* ConservedToPrimitive      */
  APK_T_MIN_DT = 7,   /* EstimateHyperbolicTimestep */
  APK_T_COPY = 8,     /* strided box copies (ghost zones) */
  APK_T_FUSED_DC_X1 = 9,  /* the three fused sweeps when the stage reconstructs with donor */
  APK_T_FUSED_DC_X2 = 10, /* cell (VL2 predictor, hydro.cpp:457-463); slots 0-2 then hold   */
  APK_T_FUSED_DC_X3 = 11, /* only the high-order stages                                    */
  APK_T_COUNT = 12
};
int apk_kernel_timing_enable(apk_ctx *ctx, int on);
/* total_ms / launches may be NULL */
int apk_kernel_timing_read(apk_ctx *ctx, int slot, double *total_ms, long long *launches);

#ifdef __cplusplus
}
#endif
#endif /* APK_AMD_H_ */
