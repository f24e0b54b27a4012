// sim.cpp -- the standalone host driver (include/apk_host.h): deck -> packages and mesh, device
// resources and plans, ghost exchange, the stage loop, outputs and the C API.
#include "sim_internal.hpp"

using namespace apk;

namespace apk {
namespace host {

int parse_bc(const std::string &v) {
  if (v == "periodic") return BC_PERIODIC;
  if (v == "outflow") return BC_OUTFLOW;
  if (v == "reflecting") return BC_REFLECT;
  throw std::runtime_error("unknown boundary condition: " + v);
}

// ---- Hydro::Initialize (src/hydro/hydro.cpp:264-826), options of this path only ---------
void hydro_initialize(apk_sim *s) {
  ParameterInput &pin = s->pin;
  HydroPackage &pkg = s->pkg;
  pkg.cfl = pin.GetOrAddReal("parthenon/time", "cfl", 0.3);
  const std::string fluid = pin.GetOrAddString("hydro", "fluid", "euler");
  if (fluid == "euler") {
    pkg.fluid = APK_FLUID_EULER;
    pkg.nhydro = 5;
  } else if (fluid == "glmmhd") {
    pkg.fluid = APK_FLUID_GLMMHD;
    pkg.nhydro = 9;
    const std::string src = pin.GetOrAddString("hydro", "glmmhd_source", "dedner_plain");
    if (src == "dedner_plain") pkg.glmmhd_source_extended = false;
    else if (src == "dedner_extended") pkg.glmmhd_source_extended = true;
    else throw std::runtime_error("AthenaPK hydro: Unknown glmmhd_source");
    pkg.glmmhd_alpha = pin.GetOrAddReal("hydro", "glmmhd_alpha", 0.1);
    pkg.calc_c_h = true;
  } else {
    throw std::runtime_error("AthenaPK hydro: Unknown fluid method.");
  }
  pkg.max_dt = pin.GetOrAddReal("hydro", "max_dt", -1.0);

  const std::string recon = pin.GetString("hydro", "reconstruction");
  int need_ng = 3;
  if (recon == "dc") pkg.recon = APK_RC_DC, need_ng = 1;
  else if (recon == "plm") pkg.recon = APK_RC_PLM, need_ng = 2;
  else if (recon == "ppm") pkg.recon = APK_RC_PPM, need_ng = 3;
  else if (recon == "limo3") pkg.recon = APK_RC_LIMO3, need_ng = 2;
  else if (recon == "weno3") pkg.recon = APK_RC_WENO3, need_ng = 2;
  else if (recon == "wenoz") pkg.recon = APK_RC_WENOZ, need_ng = 3;
  else throw std::runtime_error("AthenaPK hydro: Unknown reconstruction method.");

  pkg.calc_dt_hyp = true;
  const std::string riemann = pin.GetString("hydro", "riemann");
  if (riemann == "llf") {
    pkg.riemann = APK_RS_LLF;
    if (pkg.recon != APK_RC_DC) throw std::runtime_error("LLF Riemann solver only implemented with DC reconstruction.");
  } else if (riemann == "hlle") {
    pkg.riemann = APK_RS_HLLE;
  } else if (riemann == "hllc") {
    pkg.riemann = APK_RS_HLLC;
  } else if (riemann == "hlld") {
    pkg.riemann = APK_RS_HLLD;
  } else if (riemann == "none") {
    pkg.riemann = APK_RS_NONE;
    pkg.calc_dt_hyp = false;
    if (pkg.recon != APK_RC_DC) throw std::runtime_error("'none' Riemann solver only supported with DC reconstruction.");
  } else {
    throw std::runtime_error("AthenaPK hydro: Unknown riemann solver.");
  }
  if (pin.DoesParameterExist("hydro", "calc_dt_hyp")) pkg.calc_dt_hyp = pin.GetBoolean("hydro", "calc_dt_hyp");
  // registry check (flux_functions.at(...) throws in the reference, hydro.cpp:420)
  const bool hydro_ok = pkg.fluid == APK_FLUID_EULER && (pkg.riemann == APK_RS_HLLE || pkg.riemann == APK_RS_HLLC);
  const bool mhd_ok = pkg.fluid == APK_FLUID_GLMMHD && (pkg.riemann == APK_RS_HLLE || pkg.riemann == APK_RS_HLLD);
  if (!(hydro_ok || mhd_ok || pkg.riemann == APK_RS_LLF || pkg.riemann == APK_RS_NONE))
    throw std::runtime_error("AthenaPK hydro: no flux function for this fluid/riemann combination");

  const int nghost = pin.GetInteger("parthenon/mesh", "nghost");
  if (nghost < need_ng) throw std::runtime_error("AthenaPK hydro: Need more ghost zones for chosen reconstruction.");

  const std::string integ = pin.GetString("parthenon/time", "integrator");
  pkg.flux_other_stage = {pkg.fluid, pkg.recon, pkg.riemann};
  pkg.flux_first_stage = pkg.flux_other_stage;
  // Parthenon low-storage integrator coefficients
  if (integ == "rk1") {
    pkg.integrator = APK_INT_RK1;
    s->nstages = 1;
    s->beta[0] = 1.0, s->gam0[0] = 0.0, s->gam1[0] = 1.0;
  } else if (integ == "rk2") {
    pkg.integrator = APK_INT_RK2;
    s->nstages = 2;
    s->beta[0] = 1.0, s->gam0[0] = 0.0, s->gam1[0] = 1.0;
    s->beta[1] = 0.5, s->gam0[1] = 0.5, s->gam1[1] = 0.5;
  } else if (integ == "rk3") {
    pkg.integrator = APK_INT_RK3;
    s->nstages = 3;
    s->beta[0] = 1.0, s->gam0[0] = 0.0, s->gam1[0] = 1.0;
    s->beta[1] = 0.25, s->gam0[1] = 0.25, s->gam1[1] = 0.75;
    s->beta[2] = 2.0 / 3.0, s->gam0[2] = 2.0 / 3.0, s->gam1[2] = 1.0 / 3.0;
  } else if (integ == "vl2") {
    pkg.integrator = APK_INT_VL2;
    s->nstages = 2;
    s->beta[0] = 0.5, s->gam0[0] = 0.0, s->gam1[0] = 1.0;
    s->beta[1] = 1.0, s->gam0[1] = 0.0, s->gam1[1] = 1.0;
    // override first stage (predictor) to first order (hydro.cpp:457-463)
    pkg.flux_first_stage = {pkg.fluid, APK_RC_DC, pkg.riemann};
  } else {
    throw std::runtime_error("unknown integrator: " + integ);
  }
  pkg.first_order_flux_correct = pin.GetOrAddBoolean("hydro", "first_order_flux_correct", false);

  const std::string eos = pin.GetString("hydro", "eos");
  if (eos != "adiabatic") throw std::runtime_error("AthenaPK hydro: Unknown EOS");
  pkg.eos.gamma = pin.GetReal("hydro", "gamma");
  pkg.eos.dfloor = pin.GetOrAddReal("hydro", "dfloor", -1.0);
  pkg.eos.pfloor = pin.GetOrAddReal("hydro", "pfloor", -1.0);
  const double Tfloor = pin.GetOrAddReal("hydro", "Tfloor", -1.0);
  if (Tfloor > 0.0) throw std::runtime_error("Temperature floor requires units and gas composition (not part of this path).");
  pkg.eos.efloor = Tfloor;
  pkg.eos.vceil = pin.GetOrAddReal("hydro", "vceil", std::numeric_limits<double>::infinity());
  const double Tceil = pin.GetOrAddReal("hydro", "Tceil", std::numeric_limits<double>::infinity());
  if (Tceil < std::numeric_limits<double>::infinity())
    throw std::runtime_error("Temperature ceiling requires units and gas composition (not part of this path).");
  pkg.eos.eceil = Tceil;
  pkg.nscalars = pin.GetOrAddInteger("hydro", "nscalars", 0);
  if (pkg.nscalars < 0) throw std::runtime_error("hydro/nscalars must be >= 0");
}

void mesh_initialize(apk_sim *s) {
  ParameterInput &pin = s->pin;
  Mesh &m = s->mesh;
  const char *nxk[3] = {"nx1", "nx2", "nx3"};
  const char *mink[3] = {"x1min", "x2min", "x3min"}, *maxk[3] = {"x1max", "x2max", "x3max"};
  const char *ibc[3] = {"ix1_bc", "ix2_bc", "ix3_bc"}, *obc[3] = {"ox1_bc", "ox2_bc", "ox3_bc"};
  for (int d = 0; d < 3; ++d) {
    m.nx[d] = pin.GetOrAddInteger("parthenon/mesh", nxk[d], 1);
    m.mb[d] = pin.GetOrAddInteger("parthenon/meshblock", nxk[d], m.nx[d]);
    s->xmin[d] = pin.GetOrAddReal("parthenon/mesh", mink[d], -0.5);
    s->xmax[d] = pin.GetOrAddReal("parthenon/mesh", maxk[d], 0.5);
    s->dx[d] = (s->xmax[d] - s->xmin[d]) / (double)m.nx[d];
    m.bc_in[d] = parse_bc(pin.GetOrAddString("parthenon/mesh", ibc[d], "periodic"));
    m.bc_out[d] = parse_bc(pin.GetOrAddString("parthenon/mesh", obc[d], "periodic"));
  }
  const std::string refinement = pin.GetOrAddString("parthenon/mesh", "refinement", "none");
  if (refinement != "none" && refinement != "static" && refinement != "adaptive")
    throw std::runtime_error("parthenon/mesh/refinement must be none, static or adaptive");
  m.ng = pin.GetInteger("parthenon/mesh", "nghost");
  m.nvar = s->pkg.nhydro + s->pkg.nscalars;
  m.rank = s->rank;
  m.nranks = s->nranks;
  // (not a reference parameter: the one-GPU rehearsal of a rank of the 2 x 2 x 2 run, mesh.hpp "rehearse")
  m.rehearse = pin.GetOrAddBoolean("apk_amd", "rehearse_remote_faces", false) ? 1 : 0;
  if (m.rehearse && refinement != "none") throw std::runtime_error("apk_amd/rehearse_remote_faces needs a uniform mesh");
  // (not a reference parameter either: rows of the block arrays at a cache-line pitch, interior cells line-aligned --
  // uniform meshes without the turbulence driver, whose acceleration field has its own layout)
  // natural | aligned | auto (default).  auto = aligned where it was measured to pay (profiles/r06_row_pitch_ab.txt: the
  // nine-variable GLM-MHD marches on 128-cell rows, -0.7 .. -1.4 % per cycle; hydro PLM+HLLC with its outflow boundary
  // copies +0.8 %): 3-D GLM-MHD on blocks whose rows are whole lines.
  const std::string row_pitch = pin.GetOrAddString("apk_amd", "row_pitch", std::getenv("APK_ROW_PITCH") ? std::getenv("APK_ROW_PITCH") : "auto");
  if (row_pitch != "natural" && row_pitch != "aligned" && row_pitch != "auto") throw std::runtime_error("apk_amd/row_pitch must be natural, aligned or auto");
  const bool can_pad = refinement == "none" && s->problem_id != "turbulence";
  const bool pays = s->pkg.fluid == APK_FLUID_GLMMHD && m.mb[2] > 1 && m.mb[0] % 16 == 0 && m.mb[0] >= 64;
  if (can_pad && (row_pitch == "aligned" || (row_pitch == "auto" && pays))) {
    m.pitch = (m.mb[0] + 2 * m.ng + 15) / 16 * 16;
    m.lead = (16 - m.ng % 16) % 16;
  }
  m.Build();
  s->nblk = m.sn * m.nvar;
  s->nper = m.lead == 0 && m.pitch == 0 ? s->nblk : (s->nblk + m.lead + 15) / 16 * 16;
  if (refinement != "none") amr_initialize(s, refinement == "adaptive");
  s->tlim = pin.GetOrAddReal("parthenon/time", "tlim", 1.0);
  s->nlim = pin.GetOrAddInteger("parthenon/time", "nlim", -1);
}

// ---- device resources -----------------------------------------------------------------------
int dev_alloc(apk_sim *s, const char *tag, size_t bytes, double **out) {
  *out = nullptr;
  if (bytes == 0) return APK_OK;
  if (s->have_alloc) {
    *out = static_cast<double *>(s->alloc.alloc(s->alloc.user, tag, bytes));
    if (!*out) return fail(s, APK_ERR_DEVICE, std::string("allocator returned NULL for ") + tag);
    return APK_OK;
  }
  SIM_HIP(s, hipMalloc(out, bytes));
  return APK_OK;
}
void dev_free(apk_sim *s, double *p) {
  if (!p) return;
  if (s->have_alloc) {
    if (s->alloc.release) s->alloc.release(s->alloc.user, p);
  } else {
    (void)hipFree(p);
  }
}

int build_packs(apk_sim *s) {
  const int nlb = (int)s->mesh.local_gids.size();
  for (int p = 0; p < 3; ++p)
    for (int w = 0; w < 2; ++w) {
      if (s->mu0_of[p][w]) apk_pack_destroy(s->mu0_of[p][w]);
      if (s->mu1_of[p][w]) apk_pack_destroy(s->mu1_of[p][w]);
      s->mu0_of[p][w] = s->mu1_of[p][w] = nullptr;
      if (!s->d_prim2[w] || !s->d_cons2[p]) continue;
      double *spare = s->d_prim2[1 - w];  // may be null: then u1 carries no prim
      std::vector<apk_block_desc> b0(nlb), b1(nlb);
      for (int lb = 0; lb < nlb; ++lb) {
        b0[lb].cons = s->blk(s->d_cons2[p], lb);
        b0[lb].prim = s->blk(s->d_prim2[w], lb);
        b1[lb].cons = s->blk(s->d_cons2[p], lb);
        b1[lb].prim = spare ? s->blk(spare, lb) : nullptr;
        for (int d = 0; d < 3; ++d) {
          b0[lb].flux[d] = s->d_flux[d] ? s->blk(s->d_flux[d], lb) : nullptr;
          b1[lb].flux[d] = nullptr;
          b0[lb].dx[d] = b1[lb].dx[d] = level_dx(s, block_level(s, lb), d);
        }
      }
      apk_pack_desc d{};
      d.nblocks = nlb;
      d.nhydro = s->pkg.nhydro;
      d.nscalars = s->pkg.nscalars;
      for (int q = 0; q < 3; ++q) d.nx[q] = s->mesh.mb[q];
      d.ng = s->mesh.ng;
      if (s->mesh.pitch > 0) {
        d.stride[0] = s->mesh.sj;
        d.stride[1] = s->mesh.sk;
        d.stride[2] = s->mesh.sn;
      }
      d.blocks = b0.data();
      SIM_TRY(s, apk_pack_create(s->ctx, &d, &s->mu0_of[p][w]));
      d.blocks = b1.data();
      SIM_TRY(s, apk_pack_create(s->ctx, &d, &s->mu1_of[p][w]));
    }
  return APK_OK;
}

// third conserved buffer (output of trial stages that must keep their input), on first use
int ensure_trial_cons(apk_sim *s) {
  if (s->d_cons2[2]) return APK_OK;
  const size_t bytes = (size_t)s->nper * s->mesh.local_gids.size() * sizeof(double);
  SIM_TRY(s, dev_alloc(s, "cons3", bytes, &s->d_cons2[2]));
  SIM_HIP(s, hipMemsetAsync(s->d_cons2[2], 0, bytes, hs(s)));
  SIM_TRY(s, build_packs(s));
  return build_copy_plans(s);
}

// second primitive buffer, on first use
int ensure_spare_prim(apk_sim *s) {
  if (s->d_prim2[1 - s->pcur]) return APK_OK;
  const size_t bytes = (size_t)s->nper * s->mesh.local_gids.size() * sizeof(double);
  SIM_TRY(s, dev_alloc(s, "prim2", bytes, &s->d_prim2[1 - s->pcur]));
  SIM_HIP(s, hipMemsetAsync(s->d_prim2[1 - s->pcur], 0, bytes, hs(s)));
  SIM_TRY(s, build_packs(s));
  return build_prim_plans(s);
}

int ensure_flux_arrays(apk_sim *s) {
  bool changed = false;
  const size_t bytes = (size_t)s->nper * s->mesh.local_gids.size() * sizeof(double);
  const char *tags[3] = {"flux1", "flux2", "flux3"};
  for (int d = 0; d < s->mesh.ndim; ++d) {
    if (!s->d_flux[d]) {
      SIM_TRY(s, dev_alloc(s, tags[d], bytes, &s->d_flux[d]));
      SIM_HIP(s, hipMemsetAsync(s->d_flux[d], 0, bytes, hs(s)));
      changed = true;
    }
  }
  if (changed || !s->mu0()) return build_packs(s);
  return APK_OK;
}

bool stage_can_fuse(const apk_sim *s) {
  // (refined meshes included: the coarse-fine flux correction is applied after the fused stage from
  // boundary-plane fluxes, see amr_flux_fix)
  return s->fused && !s->pkg.first_order_flux_correct && s->pkg.riemann != APK_RS_NONE &&
         s->pkg.riemann != APK_RS_LLF;
}

// the plans of one field buffer (`field`: the first block's array; blocks follow at nper doubles)
static int make_plans(apk_sim *s, double *field, apk_copy_plan *(&out)[PH_COUNT]) {
  for (int ph = 0; ph < PH_COUNT; ++ph) {
    if (out[ph]) {
      apk_copy_plan_destroy(out[ph]);
      out[ph] = nullptr;
    }
    if (!field) continue;
    std::vector<apk_copy_region> regs;
    auto base = [&](int kind, int block) -> double * {
      if (kind == RK_BLOCK) return s->blk(field, block);
      return kind == RK_SEND ? s->send_buf[block] : s->recv_buf[block];
    };
    for (const BoxRegion &r : s->mesh.plan[ph]) {
      apk_copy_region c{};
      c.src = base(r.src_kind, r.src_block) + r.src_off;
      c.dst = base(r.dst_kind, r.dst_block) + r.dst_off;
      for (int q = 0; q < 3; ++q) c.ext[q] = r.ext[q];
      c.nvar = r.nvar;
      for (int q = 0; q < 4; ++q) {
        c.src_stride[q] = r.src_stride[q];
        c.dst_stride[q] = r.dst_stride[q];
      }
      c.flip_var = r.flip_var;
      regs.push_back(c);
    }
    SIM_TRY(s, apk_copy_plan_create(s->ctx, regs.data(), (int)regs.size(), &out[ph]));
  }
  return APK_OK;
}

int build_copy_plans(apk_sim *s) {
  for (int par = 0; par < 3; ++par) SIM_TRY(s, make_plans(s, s->d_cons2[par], s->plans_of[par]));
  return build_prim_plans(s);
}

// (exchanges of stored primitives -- GHOST_PRIM_COPY -- only happen on uniform meshes with remote neighbours or
// same-rank copies: nothing to build where no plan has a region)
int build_prim_plans(apk_sim *s) {
  for (int w = 0; w < 2; ++w) SIM_TRY(s, make_plans(s, s->amr ? nullptr : s->d_prim2[w], s->pplans_of[w]));
  return APK_OK;
}

// EvolutionDriver::SetGlobalTimeStep
void set_global_dt(apk_sim *s, double dt_est) {
  double dt = s->dt;
  if (dt < 0.1 * kHuge) dt *= 2.0;
  if (dt_est < dt) dt = dt_est;
  if (s->time < s->tlim && (s->tlim - s->time) < dt) dt = s->tlim - s->time;
  s->dt = dt;
}

// Hydro::EstimateTimestep<fluid> over this rank's pack + global min (hydro.cpp:913-977)
// The time-step estimate in two halves (round-2 advisor finding: a regridding pass between "measure" and
// "use" must not inherit side effects of an estimate made on the old mesh).
//   _read    device work + ONE host round trip: the local minimum and the device flag word.  Consumes the
//            finishing sweep's pending reduction; touches nothing of the package.
//   _commit  the reference's reaction to the flags (PARTHENON_REQUIRE in ConsToPrim), the reduction over
//            ranks, and the hyperbolic estimate the next cycle's c_h needs (hydro.cpp:102-143, 903-908).
int estimate_timestep_read(apk_sim *s, DtEstimate *e) {
  double dt = kHuge;
  unsigned flags = 0;
  bool have_flags = false;
  if (s->pkg.calc_dt_hyp) {
    if (s->stage_dt_pending) {  // already reduced by the finishing sweep of the last stage
      SIM_TRY(s, apk_stage_dt_flags_read(s->ctx, s->pkg.cfl, &dt, &flags, s->stream));  // one host round trip
      have_flags = true;
      s->stage_dt_pending = false;
    } else {
      if (s->prim_stale) SIM_TRY(s, sync_ghosts(s));  // (the estimate reads stored primitives)
      SIM_TRY(s, apk_estimate_timestep(s->ctx, s->mu0(), s->pkg.fluid, &s->pkg.eos, s->pkg.cfl, &dt, s->stream));
    }
  }
  if (!have_flags) SIM_TRY(s, apk_poll_device_flags(s->ctx, &flags, s->stream));
  e->dt_hyp_local = dt;
  e->flags |= flags;  // (flags latched before an earlier read of the same cycle stay raised)
  return APK_OK;
}

int estimate_timestep_commit(apk_sim *s, const DtEstimate &e, double *dt_out) {
  double dt = e.dt_hyp_local;
  if (s->pkg.max_dt > 0.0 && s->pkg.max_dt < dt) dt = s->pkg.max_dt;
  // one reduction for both minima: the time step, and the hyperbolic estimate that the next cycle's
  // c_h needs (hydro.cpp:102-143 reduces it in PreStepMeshUserWorkInLoop; same value, one message less).
  // The negative-state flags travel with them (two more slots, MIN of -1 / 0): a rank that latched a flag must not
  // leave the collective to its peers -- every rank takes part, then every rank fails.
  double mins[4] = {dt, e.dt_hyp_local, (e.flags & APK_FLAG_NEG_DENSITY) ? -1.0 : 0.0, (e.flags & APK_FLAG_NEG_PRESSURE) ? -1.0 : 0.0};
  if (s->have_comm && s->nranks > 1) {
    if (s->comm.allreduce_min(s->comm.user, mins, 4) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_min failed");
  }
  if (mins[2] < 0.0)
    return fail(s, APK_ERR_INVALID, "Got negative density. Consider enabling first-order flux correction or setting a reasonble density floor.");
  if (mins[3] < 0.0)
    return fail(s, APK_ERR_INVALID, "Got negative pressure. Consider enabling first-order flux correction or setting a reasonble pressure or temperature floor.");
  if (s->pkg.calc_dt_hyp && s->pkg.fluid == APK_FLUID_GLMMHD && mins[1] < s->pkg.dt_hyp) s->pkg.dt_hyp = mins[1];  // hydro.cpp:903-908
  s->dt_hyp_is_global = true;
  *dt_out = mins[0];
  return APK_OK;
}

int estimate_timestep(apk_sim *s, double *dt_out) {
  DtEstimate e;
  SIM_TRY(s, estimate_timestep_read(s, &e));
  return estimate_timestep_commit(s, e, dt_out);
}

// Ghost exchange in two halves.  begin: same-rank copies, message packing, post the transfers;
// end: wait for them, unpack, physical boundaries (x1, x2, x3).  With c2p the copies that fill
// ghost zones also convert them to primitives (apk_copy_plan_run_c2p), which replaces the separate
// ghost ConsToPrim pass -- and, around an exchange in flight, splits it by construction: the
// same-rank part in `begin`, the rest in `end`.
bool ghost_c2p_fusable(const apk_sim *s) {
  const apk_eos &e = s->pkg.eos;
  return !(e.dfloor > 0.0 || e.pfloor > 0.0 || e.efloor > 0.0 || e.vceil < 1.0e300 || e.eceil < 1.0e300);
}

// c2p: GHOST_COPY (plain), GHOST_C2P (cons and prim), GHOST_PRIM_ONLY (sim_internal.hpp)
int run_ghost_plan(apk_sim *s, int buf, int phase, int c2p, apk_stream_t stream) {
  if (!stream) stream = s->stream;
  if (!c2p) return apk_copy_plan_run(s->ctx, s->plans_of[buf][phase], stream);
  if (c2p == GHOST_PRIM_COPY) return apk_copy_plan_run(s->ctx, s->pplans_of[s->xchg_prim][phase], stream);
  const int64_t delta = s->d_prim2[s->pcur] - s->d_cons2[buf];
  // a boundary phase that is followed by another non-empty one copies corner cells from ghost
  // zones only that later phase fills: their (overwritten) primitives must not raise flags
  int latch = 1;
  for (int later = phase + 1; phase >= PH_BC1 && later <= PH_BC3; ++later)
    if (!s->mesh.plan[later].empty()) latch = 0;
  // direct neighbour addressing: the corner cells of a boundary phase are copied out of same-rank ghost
  // zones nobody fills (or reads); the other cells repeat interior cells, whose flags are latched there
  if (phase >= PH_BC1 && s->local_ghosts_stale) latch = 0;
  if (c2p == GHOST_PRIM_ONLY) return apk_copy_plan_run_c2p_prim_only(s->ctx, s->plans_of[buf][phase], s->pkg.fluid, &s->pkg.eos, delta, latch, stream);
  return apk_copy_plan_run_c2p(s->ctx, s->plans_of[buf][phase], s->pkg.fluid, &s->pkg.eos, delta, latch, stream);
}

// make the one-layer (or the full) message set the one the transports see (apk_sim_peer)
void select_thin_messages(apk_sim *s, bool thin) {
  if (s->thin_msgs == thin) return;
  s->thin_msgs = thin;
  s->msg_generation += 1;
}

int exchange_begin(apk_sim *s, bool async, int c2p, bool skip_local, bool thin) {
  const bool remote = !s->mesh.peers.empty();
  thin = thin && remote;
  s->xchg_thin = thin;
  // (x1 strips that never pass through a copy kernel: the stage just run has stored them into the send buffers, and the
  // stage that follows this exchange reads them from the receive buffers -- do_stage has checked that it will)
  const bool nox1 = remote && s->x1_out_direct;
  s->x1_out_direct = false;
  s->xchg_x1_direct = nox1;
  s->x1_in_recv = false;
  if (remote) {
    select_thin_messages(s, thin);
    s->remote_ghosts_thin = thin;  // (once this exchange is complete)
    if (thin) s->thin_exchanges += 1;
    if (nox1) s->x1_direct_exchanges += 1;
    if (c2p == GHOST_PRIM_COPY) {
      if (thin) return fail(s, APK_ERR_INVALID, "exchange_begin: a one-layer exchange carries the conserved state");
      s->xchg_prim = s->pcur;
      SIM_TRY(s, apk_copy_plan_run(s->ctx, s->pplans_of[s->pcur][nox1 ? PH_PACK_NOX1 : PH_PACK], s->stream));
    } else {
      SIM_TRY(s, apk_copy_plan_run(s->ctx, s->plan(thin ? (nox1 ? PH_PACK_THIN_NOX1 : PH_PACK_THIN) : (nox1 ? PH_PACK_NOX1 : PH_PACK)), s->stream));
    }
  } else if (c2p == GHOST_PRIM_COPY) {
    s->xchg_prim = s->pcur;
  }
  if (skip_local) {
    // direct neighbour addressing: the stages read their same-rank neighbours' interiors
    s->local_ghosts_stale = true;
    s->skipped_local_exchanges += 1;
  } else {
    SIM_TRY(s, run_ghost_plan(s, s->cur, PH_LOCAL, c2p));
  }
  if (remote) {
    if (!s->have_comm || !s->comm.exchange) return fail(s, APK_ERR_INVALID, "remote neighbours but no comm ops");
    if (async) {
      if (s->comm.exchange_begin(s->comm.user) != 0) return fail(s, APK_ERR_DEVICE, std::string("halo exchange (begin) failed ") + apk_sim_comm_error(s));
    } else if (s->comm.exchange(s->comm.user) != 0) {
      return fail(s, APK_ERR_DEVICE, "halo exchange failed");
    }
  }
  if (async) {
    s->exchange_pending = true;
    s->pending_cons = s->cur;
    s->pending_c2p = c2p;
  }
  return APK_OK;
}

int exchange_end(apk_sim *s, int c2p) {
  // an exchange left in flight targets the buffer that held the state when it was posted: the
  // first stage of the next cycle has swapped the buffer roles by the time it completes it
  const int buf = s->exchange_pending ? s->pending_cons : s->cur;
  if (s->exchange_pending) {
    if (!s->mesh.peers.empty() && s->comm.exchange_end(s->comm.user) != 0)
      return fail(s, APK_ERR_DEVICE, std::string("halo exchange (end) failed ") + apk_sim_comm_error(s));
    s->exchange_pending = false;
  }
  if (!s->mesh.peers.empty()) {
    const bool nox1 = s->xchg_x1_direct;
    SIM_TRY(s, run_ghost_plan(s, buf, s->xchg_thin ? (nox1 ? PH_UNPACK_THIN_NOX1 : PH_UNPACK_THIN) : (nox1 ? PH_UNPACK_NOX1 : PH_UNPACK), c2p));
    s->x1_in_recv = nox1;
  }
  for (int ph = PH_BC1; ph <= PH_BC3; ++ph) SIM_TRY(s, run_ghost_plan(s, buf, ph, c2p));
  return APK_OK;
}

int exchange_ghosts(apk_sim *s, int c2p, bool skip_local, bool thin) {
  if (s->amr) return amr_exchange(s, s->cur);
  if (!skip_local) s->local_ghosts_stale = false;  // (a full exchange of the current state)
  SIM_TRY(s, exchange_begin(s, false, c2p, skip_local, thin));
  return exchange_end(s, c2p);
}

// Can the stages of this simulation read same-rank neighbours directly (apk_stage_args.face_neighbor)
// so that the same-rank ghost copies can be skipped?  Every stage of the cycle must be one of the
// kernels that follow the table, and nothing else in the cycle may read ghost zones.  (do_stage skips
// the copies only after stages whose FillDerived was fused -- into the finishing sweep or, with the
// turbulence driver, into the kick; an exchange followed by a full-block ConsToPrim is a complete one.)
// one rank, every active direction periodic: the face table covers every face of every block, so an exchange that
// follows it has no ghost zone left to fill (edges and corners are read by no stage that follows the table)
bool table_covers_all_faces(const apk_sim *s) {
  const Mesh &m = s->mesh;
  if (!m.peers.empty()) return false;
  for (int d = 0; d < 3; ++d)
    if (m.Active(d) && (m.bc_in[d] != BC_PERIODIC || m.bc_out[d] != BC_PERIODIC)) return false;
  return true;
}

bool direct_neighbors(const apk_sim *s) {
  static const int mode = std::getenv("APK_DIRECT_NEIGHBORS") ? std::atoi(std::getenv("APK_DIRECT_NEIGHBORS")) : 1;  // A/B switch
  static const int dc_mode = std::getenv("APK_DC_MODE") ? std::atoi(std::getenv("APK_DC_MODE")) : 2;
  const HydroPackage &pkg = s->pkg;
  if (!mode || !s->direct_on || !s->d_face_nbr || s->amr || s->mesh.ndim != 3) return false;
  // first-order flux correction: every stage runs as the optimistic fused stage (do_stage) -- the same kernels with the
  // admissibility test in the finishing sweep; a stage that fails it is redone through the flux arrays, which read ghost
  // zones: do_stage fills them first (materialize_local_ghosts).  One-rank periodic boxes, no forcing.
  const bool optimistic = s->fused && pkg.first_order_flux_correct && !s->fmft && pkg.riemann != APK_RS_NONE && pkg.riemann != APK_RS_LLF &&
                          table_covers_all_faces(s);
  if (!stage_can_fuse(s) && !optimistic) return false;
  // (floors and ceilings: ConsToPrim is not fused into the ghost fills then, and the separate pass over the ghost zones
  // would convert the zones nobody filled -- unless no zone is left to fill at all.  What the stages read across a face
  // is the neighbour's floored interior state either way: the values the reference's ConsToPrim of a ghost cell produces
  // from the same conserved input.)
  if (pkg.nscalars != 0 || (!ghost_c2p_fusable(s) && !table_covers_all_faces(s))) return false;
  if (pkg.fluid == APK_FLUID_GLMMHD && pkg.glmmhd_source_extended) return false;
  const apk_flux_cfg *cfgs[2] = {&pkg.flux_first_stage, &pkg.flux_other_stage};
  for (const apk_flux_cfg *cfg : cfgs) {
    if (cfg->recon == APK_RC_DC) {
      if (dc_mode != 2) return false;
    } else if (apk_stage_split_axis(s->mu0(), cfg, 2) != 3) {
      return false;
    }
  }
  return true;
}

// Refined meshes: may the stages read a same-rank neighbour of the SAME level through the face table (built in
// amr_rebuild) instead of the ghost zone behind that face?  Then the faces-only exchange of the stage loop skips
// those copies and their ConsToPrim (AMR_XCHG_DIRECT).  The stage forms that follow the table: the single-march
// donor-cell stage and the two-kernel stage (launch_fused_stage); refined-mesh stages run without FillDerived.
bool amr_direct(const apk_sim *s) {
  static const int mode = std::getenv("APK_DIRECT_NEIGHBORS") ? std::atoi(std::getenv("APK_DIRECT_NEIGHBORS")) : 1;  // A/B switch
  const HydroPackage &pkg = s->pkg;
  if (!mode || !s->direct_on || !s->amr || !s->d_face_nbr || s->mesh.ndim != 3 || !stage_can_fuse(s) || !amr_faces_only(s)) return false;
  if (pkg.nscalars != 0 || (pkg.fluid == APK_FLUID_GLMMHD && pkg.glmmhd_source_extended)) return false;
  const apk_flux_cfg *cfgs[2] = {&pkg.flux_first_stage, &pkg.flux_other_stage};
  for (const apk_flux_cfg *cfg : cfgs)
    if (cfg->recon != APK_RC_DC && apk_stage_split_axis(s->mu0(), cfg, 0) != 3) return false;
  return true;
}

// does the cycle in progress end with a check of the refinement criteria?  Those read the full ring of ghost
// cells round a block -- edges and corners too (refinement/gradient.cpp:33-36 loops k, j, i over [s-1, e+1]
// and differences each of them) -- so the exchange after the last stage of such a cycle is a complete one, or at
// least two layers deep all round (amr_shell_before_check).
bool regrid_check_follows(const apk_sim *s) {
  return s->amr && s->amr_adaptive && s->amr_check_interval > 0 && (s->ncycle + 1) % s->amr_check_interval == 0;
}

// may the stage loop of a refined mesh skip the ghost zones behind edges and corners?  (apk_sim_set_amr_full_exchange(1):
// never)
bool amr_faces_only(const apk_sim *s) {
  return s->amr && !s->amr_full_exchange && s->mesh.ndim >= 2;
}

// Refined meshes, the exchange after the last stage of a cycle that ends with a refinement check (regrid_check_follows):
// may it fill the ghost zones AMR_SHELL_DEPTH layers deep only?  Tagging reads that far (refinement/gradient.cpp:33-36),
// and so must the first stage of the next cycle: a donor-cell stage (the VL2 predictor) reads one layer, PLM two.
// (Whoever needs more -- accessors, the data transfer of a regridding that does change the mesh -- calls sync_ghosts.)
bool amr_shell_before_check(const apk_sim *s) {
  const HydroPackage &pkg = s->pkg;
  if (!s->amr || !amr_faces_only(s) || !amr_has_shell(s) || !stage_can_fuse(s) || !pkg.calc_dt_hyp) return false;
  // the shell is AMR_SHELL_DEPTH layers deep: the first stage of the next cycle may read no deeper -- its stencil
  // half width plus the face it reconstructs for (DC 1 layer, PLM 2; PPM / WENO-Z 3 would read stale cells)
  const int recon = pkg.flux_first_stage.recon;
  const int reach = (recon == APK_RC_DC) ? 1 : ((recon == APK_RC_PPM || recon == APK_RC_WENOZ) ? 3 : 2);
  return reach <= AMR_SHELL_DEPTH && !(pkg.fluid == APK_FLUID_GLMMHD && pkg.glmmhd_source_extended && AMR_SHELL_DEPTH < 2);
}

// fill the ghost zones that direct neighbour addressing left stale: cons of buffer `buf` (default: the current state)
// and the stored primitives, which are that buffer's
int materialize_local_ghosts(apk_sim *s, int buf) {
  if (!s->local_ghosts_stale) return APK_OK;
  s->local_ghosts_stale = false;
  if (buf < 0) buf = s->cur;
  // (floors / ceilings: plain copies, then the pass over the ghost zones -- the order of a cycle without the table)
  const int mode = ghost_c2p_fusable(s) ? GHOST_C2P : GHOST_COPY;
  SIM_TRY(s, run_ghost_plan(s, buf, PH_LOCAL, mode));
  // (physical boundaries copy corner cells out of ghost zones the same-rank copies fill)
  for (int ph = PH_BC1; ph <= PH_BC3; ++ph) SIM_TRY(s, run_ghost_plan(s, buf, ph, mode));
  // (no stored primitives: the caller converts whole blocks, materialize_prim)
  if (mode == GHOST_COPY && !s->prim_stale)
    SIM_TRY(s, apk_cons_to_prim_ghosts(s->ctx, s->mu0_of[buf][s->pcur], s->pkg.fluid, &s->pkg.eos, s->stream));
  return APK_OK;
}

// Can stage 1 of a cycle take its input from the conserved state, so that the last stage of the cycle before it need
// not store primitives?  The single-march donor-cell stage in its lean form (uniform 3-D mesh, VL2), and a last stage
// that is the lean two-kernel stage with the time-step estimate fused in.
bool prim_free_cycle(const apk_sim *s) {
  static const int mode = std::getenv("APK_PRIM_FREE") ? std::atoi(std::getenv("APK_PRIM_FREE")) : 1;  // A/B switch
  static const int dc_mode = std::getenv("APK_DC_MODE") ? std::atoi(std::getenv("APK_DC_MODE")) : 2;
  const HydroPackage &pkg = s->pkg;
  if (!mode || !s->prim_free_on || s->amr || s->fmft || s->mesh.ndim != 3 || !stage_can_fuse(s) || dc_mode != 2) return false;
  if (pkg.nscalars != 0 || (pkg.fluid == APK_FLUID_GLMMHD && pkg.glmmhd_source_extended) || !pkg.calc_dt_hyp) return false;
  const apk_eos &e = pkg.eos;
  if (!(e.vceil > 1.0e300 && e.eceil > 1.0e300 && e.pfloor <= 0.0)) return false;  // (eos_is_lean)
  if (pkg.flux_first_stage.recon != APK_RC_DC || pkg.flux_other_stage.recon == APK_RC_DC || s->nstages < 2) return false;
  return apk_stage_split_axis(s->mu0(), &pkg.flux_other_stage, 2) == 3;
}

// The same for integrators whose stages are all two-kernel stages (RK1 / RK2 / RK3 with PLM, PPM, WENO-Z ...): every
// stage derives its input from the conserved state (apk_stage_args.prim_from_cons: u1's in stages with gam0 = 0, u0's
// with an out-of-place result in the others -- a third buffer in rotation, as for the trial stages of first-order flux
// correction) and stores no primitives; the last one computes them for the time-step estimate (fill_derived = 3).
// APK_RK_PRIM_FREE=0 switches it off (A/B).
bool rk_prim_free_cycle(const apk_sim *s) {
  static const int mode = std::getenv("APK_RK_PRIM_FREE") ? std::atoi(std::getenv("APK_RK_PRIM_FREE")) : 1;  // A/B switch
  const HydroPackage &pkg = s->pkg;
  // (forced turbulence included: its kick after the last stage estimates the time step without storing primitives,
  // apk_turb_apply_dt)
  if (!mode || !s->prim_free_on || s->amr || s->mesh.ndim != 3 || !stage_can_fuse(s)) return false;
  if (pkg.nscalars != 0 || (pkg.fluid == APK_FLUID_GLMMHD && pkg.glmmhd_source_extended) || !pkg.calc_dt_hyp) return false;
  const apk_eos &e = pkg.eos;
  if (!(e.vceil > 1.0e300 && e.eceil > 1.0e300 && e.pfloor <= 0.0)) return false;  // (eos_is_lean)
  // Stages that are not the last store their result without ConsToPrim (fill_derived = 0): a density or internal-energy
  // floor would act on the register copy the next stage converts but never reach the stored conserved state, where the
  // reference's FillDerived after every stage writes the floored values back (adiabatic_hydro.hpp:81,129-136).  With
  // floors the cycle keeps its primitives (every stage fill_derived = 2).
  if (e.dfloor > 0.0 || e.efloor > 0.0) return false;
  if (pkg.flux_first_stage.recon == APK_RC_DC || pkg.flux_other_stage.recon == APK_RC_DC) return false;
  return apk_stage_split_axis(s->mu0(), &pkg.flux_first_stage, 0) == 3 && apk_stage_split_axis(s->mu0(), &pkg.flux_other_stage, 0) == 3;
}

// Refined meshes, VL2 with a high-order corrector in the two-kernel form (BASELINE config 5): may the corrector derive
// its input from the half-step CONSERVED state (apk_stage_args.prim_from_cons = 2) -- so that no ConsToPrim pass runs
// between the two stages, 38 of 810 us per cycle on config 5's mesh -- and the flux correction's boundary planes likewise
// (apk_calculate_fluxes_boundary_list_from_cons)?  The corrector's result goes over the register u1, cell by cell the
// value the lane has just read, and the two buffers swap roles.  No floors or ceilings (the in-register ConsToPrim is
// the lean one and writes nothing back), no passive scalars, no forcing.  apk_sim_set_prim_free(0) / APK_AMR_PRIM_FREE=0
// switch it off (A/B).
bool amr_prim_free_cycle(const apk_sim *s) {
  static const int mode = std::getenv("APK_AMR_PRIM_FREE") ? std::atoi(std::getenv("APK_AMR_PRIM_FREE")) : 1;
  static const int dc_mode = std::getenv("APK_DC_MODE") ? std::atoi(std::getenv("APK_DC_MODE")) : 2;  // (the predictor's form)
  const HydroPackage &pkg = s->pkg;
  if (dc_mode != 2) return false;
  if (!mode || !s->prim_free_on || !s->amr || s->fmft || s->mesh.ndim != 3 || !stage_can_fuse(s) || !amr_faces_only(s)) return false;
  if (pkg.nscalars != 0 || (pkg.fluid == APK_FLUID_GLMMHD && pkg.glmmhd_source_extended)) return false;
  const apk_eos &e = pkg.eos;
  if (!(e.vceil > 1.0e300 && e.eceil > 1.0e300 && e.pfloor <= 0.0 && e.dfloor <= 0.0 && e.efloor <= 0.0)) return false;
  if (pkg.flux_first_stage.recon != APK_RC_DC || pkg.flux_other_stage.recon == APK_RC_DC || s->nstages != 2) return false;
  if (s->gam0[1] != 0.0) return false;  // (the corrector must not read the old u0: VL2)
  return apk_stage_split_axis(s->mu0(), &pkg.flux_other_stage, 0) == 3 && apk_stage_split_axis(s->mu0(), &pkg.flux_other_stage, 2) == 3;
}

// May the exchange at the end of a cycle deliver ONE layer of ghost cells (mesh.hpp PH_PACK_THIN)?  The first stage of the
// next cycle must be the single-march donor-cell stage (it reads one layer; the corrector's exchange stays a full one),
// nothing else in a cycle may read ghost zones (no forcing, no refinement), and the box must be periodic: a physical
// boundary phase copies corner cells out of ghost zones the messages fill.  Uniform 3-D meshes, exchanges left in
// flight (the path of N > 1 runs).  APK_THIN_EXCHANGE=0 switches it off (A/B).
bool thin_exchange_cycle(const apk_sim *s) {
  static const int mode = std::getenv("APK_THIN_EXCHANGE") ? std::atoi(std::getenv("APK_THIN_EXCHANGE")) : 1;
  static const int dc_mode = std::getenv("APK_DC_MODE") ? std::atoi(std::getenv("APK_DC_MODE")) : 2;
  const HydroPackage &pkg = s->pkg;
  const Mesh &mm = s->mesh;
  if (!mode || !s->thin_on || s->amr || s->fmft || mm.ndim != 3 || mm.peers.empty() || !stage_can_fuse(s) || dc_mode != 2) return false;
  if (mm.ng <= kThinDepth || pkg.nscalars != 0 || (pkg.fluid == APK_FLUID_GLMMHD && pkg.glmmhd_source_extended)) return false;
  if (pkg.flux_first_stage.recon != APK_RC_DC || s->nstages < 2) return false;
  for (int d = 0; d < 3; ++d)
    if (mm.bc_in[d] != BC_PERIODIC || mm.bc_out[d] != BC_PERIODIC) return false;
  return true;
}

// May the x1 strips of this cycle's exchanges bypass the pack / unpack kernels (apk_stage_args.x1_halo)?  On a uniform
// periodic 3-D mesh with remote neighbours, in the leanest forms of a cycle:
//   1  VL2: the predictor reads the conserved state (prim_free_cycle) one layer deep (thin_exchange_cycle) and sends
//      primitives (GHOST_PRIM_COPY), the corrector reads those and sends one layer of the conserved state;
//   2  the RK integrators whose stages all derive their input from the conserved state (rk_prim_free_cycle) in the
//      two-kernel form: every exchange moves the conserved state nghost deep, and every finishing march stores its x1
//      strips into the messages and reads the ones of the stage before from them;
// both in stage forms that follow the table.  0: neither.  APK_X1_DIRECT=0: off (A/B).
int x1_direct_kind(const apk_sim *s) {
  static const int mode = std::getenv("APK_X1_DIRECT") ? std::atoi(std::getenv("APK_X1_DIRECT")) : 1;
  const HydroPackage &pkg = s->pkg;
  const Mesh &mm = s->mesh;
  if (!mode || !s->x1_on || !s->d_x1_tab[0] || mm.mb[0] < 2 * mm.ng || mm.peers.empty()) return 0;
  if (!direct_neighbors(s) || !ghost_c2p_fusable(s)) return 0;
  for (int d = 0; d < 3; ++d)
    if (mm.bc_in[d] != BC_PERIODIC || mm.bc_out[d] != BC_PERIODIC) return 0;
  const int ded = (pkg.fluid == APK_FLUID_GLMMHD) ? 1 : 0;
  if (s->nstages == 2 && thin_exchange_cycle(s) && prim_free_cycle(s)) {
    return (apk_stage_x1_halo(s->mu0(), &pkg.flux_first_stage, &pkg.eos, 2, ded, 1) == 1 &&
            apk_stage_x1_halo(s->mu0(), &pkg.flux_other_stage, &pkg.eos, 3, ded, 0) == 1) ? 1 : 0;
  }
  if (rk_prim_free_cycle(s)) {
    // (stage 1 reads u1's state, the others u0's with an out-of-place result; the last computes primitives for dt only)
    return (apk_stage_x1_halo(s->mu0(), &pkg.flux_first_stage, &pkg.eos, s->nstages == 1 ? 3 : 0, ded, 1) == 1 &&
            apk_stage_x1_halo(s->mu0(), &pkg.flux_other_stage, &pkg.eos, 0, ded, 2) == 1 &&
            apk_stage_x1_halo(s->mu0(), &pkg.flux_other_stage, &pkg.eos, 3, ded, 2) == 1) ? 2 : 0;
  }
  return 0;
}
bool x1_direct_cycle(const apk_sim *s) { return x1_direct_kind(s) != 0; }

// the per-block segment tables of apk_stage_args.x1_halo (apk_sim::d_x1_tab), from Mesh::x1_send / x1_recv
int build_x1_tables(apk_sim *s) {
  const Mesh &m = s->mesh;
  if (m.peers.empty() || m.ndim != 3) return APK_OK;
  const size_t nlb = m.local_gids.size();
  std::vector<apk_x1_halo_block> pred(nlb), corr(nlb), full(nlb);
  for (size_t lb = 0; lb < nlb; ++lb)
    for (int side = 0; side < 2; ++side) {
      const X1Segment &sd = m.x1_send[lb][side], &rv = m.x1_recv[lb][side];
      pred[lb].send[side] = sd.peer >= 0 ? s->send_buf[sd.peer] + sd.off : nullptr;
      corr[lb].send[side] = sd.peer >= 0 ? s->send_buf[sd.peer] + sd.off_thin : nullptr;
      pred[lb].recv[side] = rv.peer >= 0 ? s->recv_buf[rv.peer] + rv.off_thin : nullptr;
      corr[lb].recv[side] = rv.peer >= 0 ? s->recv_buf[rv.peer] + rv.off : nullptr;
      full[lb].send[side] = pred[lb].send[side];  // (the RK integrators: full messages both ways)
      full[lb].recv[side] = corr[lb].recv[side];
    }
  const char *tags[3] = {"x1_halo_predictor", "x1_halo_corrector", "x1_halo_full"};
  const std::vector<apk_x1_halo_block> *tabs[3] = {&pred, &corr, &full};
  for (int q = 0; q < 3; ++q) {
    double *p = nullptr;
    SIM_TRY(s, dev_alloc(s, tags[q], sizeof(apk_x1_halo_block) * nlb, &p));
    s->d_x1_tab[q] = p;
    SIM_HIP(s, hipMemcpy(p, tabs[q]->data(), sizeof(apk_x1_halo_block) * nlb, hipMemcpyHostToDevice));
  }
  return APK_OK;
}

int materialize_prim(apk_sim *s) {
  if (!s->prim_stale) return APK_OK;
  s->prim_stale = false;
  return fill_derived(s);  // (every cell of every block: the ghost zones have been brought up to date by the caller)
}

// The last exchange was a one-layer one: repeat it in full (cons; and prim unless no primitives of this state are
// stored).  A collective over the ranks, like the completion of a refined mesh's ghost zones below: accessors that
// reach it are to be called on every rank.
int materialize_remote_ghosts(apk_sim *s) {
  if (!s->remote_ghosts_thin && !s->x1_in_recv) return APK_OK;  // (... or it left its x1 strips in the receive buffers)
  if (s->exchange_pending) return fail(s, APK_ERR_INVALID, "materialize_remote_ghosts: an exchange is in flight");
  if (!s->have_comm || !s->comm.exchange) return fail(s, APK_ERR_INVALID, "remote neighbours but no comm ops");
  // (the message half of exchange_begin / exchange_end: same-rank ghost zones are none of its business)
  const int mode = (!s->prim_stale && ghost_c2p_fusable(s)) ? GHOST_C2P : GHOST_COPY;
  select_thin_messages(s, false);
  s->xchg_thin = s->remote_ghosts_thin = false;
  s->xchg_x1_direct = s->x1_in_recv = s->x1_out_direct = false;
  SIM_TRY(s, apk_copy_plan_run(s->ctx, s->plan(PH_PACK), s->stream));
  if (s->comm.exchange(s->comm.user) != 0) return fail(s, APK_ERR_DEVICE, "halo exchange failed");
  SIM_TRY(s, run_ghost_plan(s, s->cur, PH_UNPACK, mode));
  for (int ph = PH_BC1; ph <= PH_BC3; ++ph) SIM_TRY(s, run_ghost_plan(s, s->cur, ph, mode));
  if (mode == GHOST_COPY && !s->prim_stale) SIM_TRY(s, apk_cons_to_prim_ghosts(s->ctx, s->mu0(), s->pkg.fluid, &s->pkg.eos, s->stream));
  return APK_OK;
}

int sync_ghosts(apk_sim *s) {
  if (!s->amr && s->prim_stale) {
    SIM_TRY(s, finish_pending(s));
    SIM_TRY(s, materialize_remote_ghosts(s));
    SIM_TRY(s, materialize_local_ghosts(s));
    return materialize_prim(s);
  }
  if (s->amr) {
    if (s->amr_ghost_state == AMR_GHOSTS_COMPLETE) return materialize_prim(s);  // (amr_prim_free_cycle may have left them stale)
    // the stage loop left the ghost zones behind edges and corners alone (or filled all of them a few layers deep):
    // complete exchange + ConsToPrim
    SIM_TRY(s, amr_exchange(s, s->cur, AMR_XCHG_FULL));
    s->prim_stale = false;
    return fill_derived(s);
  }
  SIM_TRY(s, finish_pending(s));
  SIM_TRY(s, materialize_remote_ghosts(s));
  return materialize_local_ghosts(s);
}


// index windows of the split stages, per local block (see apk_stage_args.window)
int upload_window(apk_sim *s, const char *tag, const std::vector<int> &w, apk_sim::WindowTable &t) {
  const size_t nlb = w.size() / 8;
  t.rl = t.rows = 0;
  t.any = false;
  for (size_t lb = 0; lb < nlb; ++lb) {
    const int *q = &w[8 * lb];
    if (q[1] <= 0 || q[3] < q[2] || q[5] < q[4] || q[7] < q[6]) continue;
    t.any = true;
    t.rl = std::max(t.rl, q[1]);
    t.rows = std::max(t.rows, q[5] - q[4] + 1);
  }
  double *p = nullptr;
  SIM_TRY(s, dev_alloc(s, tag, sizeof(int) * w.size(), &p));
  t.d = reinterpret_cast<int *>(p);
  SIM_HIP(s, hipMemcpy(t.d, w.data(), sizeof(int) * w.size(), hipMemcpyHostToDevice));
  return APK_OK;
}

int build_windows(apk_sim *s) {
  const Mesh &m = s->mesh;
  const int nlb = (int)m.local_gids.size();
  const int W = m.ng;
  const int S[3] = {m.is, m.js, m.ks}, E[3] = {m.ie, m.je, m.ke};
  auto put = [](std::vector<int> &t, int lb, int i0, int rl, int ilo, int ihi, int jlo, int jhi, int klo, int khi) {
    int *q = &t[8 * (size_t)lb];
    q[0] = i0, q[1] = rl, q[2] = ilo, q[3] = ihi, q[4] = jlo, q[5] = jhi, q[6] = klo, q[7] = khi;
  };
  std::vector<int> x1[3], dc[7], k3[3];
  for (auto &t : x1) t.assign(8 * (size_t)nlb, 0);
  for (auto &t : k3) t.assign(8 * (size_t)nlb, 0);
  for (auto &t : dc) t.assign(8 * (size_t)nlb, 0);
  std::vector<unsigned> late(nlb, 0u);
  // (two-kernel stage: when no block has BOTH its x3 faces late -- the 2 x 2 x 2 brick: every block is a corner -- the
  // low and the high slabs are one table, one launch with twice the waves: a slab launch of half the blocks fills half
  // the GPU for the length of a whole march prologue)
  bool k3_one_slab = true;
  for (int lb = 0; lb < nlb; ++lb)
    if (m.Active(2) && m.LateFace(lb, 2, -1) && m.LateFace(lb, 2, +1)) k3_one_slab = false;
  for (int lb = 0; lb < nlb; ++lb) {
    // "late" faces: ghost zones filled by messages of other ranks (physical boundaries are applied after them)
    int L[3][2];
    for (int d = 0; d < 3; ++d) L[d][0] = (m.Active(d) && m.LateFace(lb, d, -1)) ? 1 : 0;
    for (int d = 0; d < 3; ++d) L[d][1] = m.LateFace(lb, d, +1) ? 1 : 0;
    // x1 sweep of a high-order stage: everything farther than nghost from a late x1 face, then the slabs
    put(x1[0], lb, 0, m.ni, m.is + W * L[0][0], m.ie - W * L[0][1], m.js, m.je, m.ks, m.ke);
    // (two columns of margin on the left: the L state of the cell below the first retired one needs, with
    // PPM's shared interface values, the lane below it as well -- x1_first_lane in fused_kernel.hpp)
    put(x1[1], lb, m.is - 2, L[0][0] ? W + 3 : 0, m.is, m.is + W - 1, m.js, m.je, m.ks, m.ke);
    put(x1[2], lb, m.ie - W - 1, L[0][1] ? W + 3 : 0, m.ie - W + 1, m.ie, m.js, m.je, m.ks, m.ke);
    // two-kernel stage (3-D): its first kernel is the x3 sweep, which reads x3 ghost zones only --
    // every plane farther than nghost from a late x3 face, then the slabs next to those faces
    put(k3[0], lb, 0, m.ni, m.is, m.ie, m.js, m.je, m.ks + W * L[2][0], m.ke - W * L[2][1]);
    if (k3_one_slab) {
      const bool hi = L[2][1] != 0;
      put(k3[1], lb, 0, (L[2][0] || L[2][1]) ? m.ni : 0, m.is, m.ie, m.js, m.je, hi ? m.ke - W + 1 : m.ks, hi ? m.ke : m.ks + W - 1);
      put(k3[2], lb, 0, 0, m.is, m.ie, m.js, m.je, m.ke - W + 1, m.ke);
    } else {
      put(k3[1], lb, 0, L[2][0] ? m.ni : 0, m.is, m.ie, m.js, m.je, m.ks, m.ks + W - 1);
      put(k3[2], lb, 0, L[2][1] ? m.ni : 0, m.is, m.ie, m.js, m.je, m.ke - W + 1, m.ke);
    }
    // single-kernel donor-cell stage (3-D): everything but the one-cell layers next to late
    // faces, then disjoint slabs: z (whole planes), y (rows of the remaining planes), x (columns)
    const int lo[3] = {S[0] + L[0][0], S[1] + L[1][0], S[2] + L[2][0]};
    const int hi[3] = {E[0] - L[0][1], E[1] - L[1][1], E[2] - L[2][1]};
    put(dc[0], lb, 0, m.ni, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]);
    for (int side = 0; side < 2; ++side) {
      const int kk = side ? E[2] : S[2], jj = side ? E[1] : S[1], ii = side ? E[0] : S[0];
      put(dc[1 + side], lb, 0, L[2][side] ? m.ni : 0, S[0], E[0], S[1], E[1], kk, kk);
      put(dc[3 + side], lb, 0, L[1][side] ? m.ni : 0, S[0], E[0], jj, jj, lo[2], hi[2]);
      put(dc[5 + side], lb, ii - 1, L[0][side] ? 3 : 0, ii, ii, lo[1], hi[1], lo[2], hi[2]);
    }
    // lateness per neighbour region for the split ghost ConsToPrim
    int bc[3], nbc[3];
    m.Loc(m.local_gids[lb], bc);
    for (int sz = -1; sz <= 1; ++sz)
      for (int sy = -1; sy <= 1; ++sy)
        for (int sx = -1; sx <= 1; ++sx) {
          if (!sx && !sy && !sz) continue;
          if ((sx && !m.Active(0)) || (sy && !m.Active(1)) || (sz && !m.Active(2))) continue;
          const int o[3] = {sx, sy, sz};
          const bool is_late = !m.Neighbor(bc, o, nbc) || m.NeighborRank(bc, o, nbc) != m.rank;
          if (is_late) late[lb] |= 1u << ((sx + 1) + 3 * (sy + 1) + 9 * (sz + 1));
        }
  }
  {
    double *p = nullptr;
    SIM_TRY(s, dev_alloc(s, "late_regions", sizeof(unsigned) * (size_t)nlb, &p));
    s->d_late_regions = reinterpret_cast<unsigned *>(p);
    SIM_HIP(s, hipMemcpy(s->d_late_regions, late.data(), sizeof(unsigned) * (size_t)nlb, hipMemcpyHostToDevice));
  }
  const char *x1tags[3] = {"win_x1_main", "win_x1_lo", "win_x1_hi"};
  for (int q = 0; q < 3; ++q) SIM_TRY(s, upload_window(s, x1tags[q], x1[q], s->x1win[q]));
  if (m.ndim == 3) {
    const char *k3tags[3] = {"win_k3_main", "win_k3_lo", "win_k3_hi"};
    for (int q = 0; q < 3; ++q) SIM_TRY(s, upload_window(s, k3tags[q], k3[q], s->k3win[q]));
    const char *dctags[7] = {"win_dc_main", "win_dc_zlo", "win_dc_zhi", "win_dc_ylo", "win_dc_yhi", "win_dc_xlo", "win_dc_xhi"};
    for (int q = 0; q < 7; ++q) SIM_TRY(s, upload_window(s, dctags[q], dc[q], s->dcwin[q]));
  }
  return APK_OK;
}

// apk_stage_args.face_neighbor of this rank's pack: the same-rank block behind every face, or -1
// (a physical boundary, a block of another rank: those ghost zones are filled by the exchange)
int build_face_table(apk_sim *s) {
  const Mesh &m = s->mesh;
  const int nlb = (int)m.local_gids.size();
  std::vector<int> tab(6 * (size_t)nlb, -1);
  for (int lb = 0; lb < nlb; ++lb) {
    int bc[3], nbc[3];
    m.Loc(m.local_gids[lb], bc);
    for (int d = 0; d < 3; ++d)
      for (int side = 0; side < 2; ++side) {
        int o[3] = {0, 0, 0};
        o[d] = side ? 1 : -1;
        if (!m.Active(d) || !m.Neighbor(bc, o, nbc)) continue;
        const int ngid = m.Gid(nbc);
        if (m.NeighborRank(bc, o, nbc) == m.rank) tab[6 * (size_t)lb + 2 * d + side] = m.gid_local.at(ngid);
      }
  }
  double *p = nullptr;
  SIM_TRY(s, dev_alloc(s, "face_neighbors", sizeof(int) * tab.size(), &p));
  s->d_face_nbr = reinterpret_cast<int *>(p);
  SIM_HIP(s, hipMemcpy(s->d_face_nbr, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice));
  return APK_OK;
}

// can the exchange posted after a stage stay in flight while the stage `next` (1-based) starts?
bool can_overlap_next(const apk_sim *s, int next) {
  const Mesh &m = s->mesh;
  // (messages to other ranks and / or same-rank copies on the copy stream)
  if (!(s->overlap && !s->amr && stage_can_fuse(s) && m.ndim >= 2 && s->x1win[0].d)) return false;
  if (!m.peers.empty() && !(s->have_comm && s->comm.exchange_begin && s->comm.exchange_end)) return false;
  if (m.peers.empty()) return false;
  const apk_flux_cfg &cfg = (next == 1) ? s->pkg.flux_first_stage : s->pkg.flux_other_stage;
  const bool ext_dedner = s->pkg.fluid == APK_FLUID_GLMMHD && s->pkg.glmmhd_source_extended;
  if (cfg.recon == APK_RC_DC) {
    // single-kernel donor-cell stage: 3-D, out-of-place FillDerived, no dt in the kernel
    return m.ndim == 3 && !ext_dedner && m.mb[0] >= 4 && m.mb[1] >= 4 && m.mb[2] >= 4 &&
           !(next == s->nstages && s->pkg.calc_dt_hyp) && !(s->fmft && next == s->nstages);
  }
  // A stage that runs as ONE march when left whole (hydro PLM in a prim-free RK cycle: 0.73 ms on 8 x 128^3 against 0.90
  // for the two kernels a split stage is made of) is left whole: the wire time the split could hide is 0.1 - 0.2 ms.
  if (rk_prim_free_cycle(s) && apk_stage_single_march(s->mu0(), &cfg)) return false;
  // x1 column windows of the three-sweep schedule / x3 plane windows of the two-kernel one
  const bool planes = m.ndim == 3 && apk_stage_split_axis(s->mu0(), &cfg, 2) == 3;
  return planes ? m.mb[2] >= 4 * m.ng : m.mb[0] >= 4 * m.ng;
}

// complete an exchange left in flight (accessors, end of run): ghosts of cons and prim are valid after
int finish_pending(apk_sim *s) {
  if (!s->exchange_pending) return APK_OK;
  apk_pack *state = s->mu0_of[s->pending_cons][s->pcur];
  const int c2p = s->pending_c2p;
  SIM_TRY(s, exchange_end(s, c2p));
  if (c2p) return APK_OK;  // the ghost zones were converted as they were filled
  // (no primitives of this state are stored anywhere: whoever wants them runs materialize_prim over whole blocks)
  if (s->prim_stale) return APK_OK;
  return apk_cons_to_prim_ghosts(s->ctx, state, s->pkg.fluid, &s->pkg.eos, s->stream);
}

int fill_derived(apk_sim *s) {
  return apk_cons_to_prim(s->ctx, s->mu0(), s->pkg.fluid, &s->pkg.eos, s->stream);
}

// Hydro::PreStepMeshUserWorkInLoop (hydro.cpp:102-143)
int pre_step(apk_sim *s) {
  if (!s->pkg.calc_c_h) return APK_OK;
  // CalculateGlobalMinDx (hydro.cpp:65-95): over the blocks that exist, i.e. the finest level present
  int finest = 0;
  if (s->amr)
    for (const AmrLeaf &l : s->amr->leaves) finest = std::max(finest, l.level);
  double mindx = level_dx(s, finest, 0);
  if (s->mesh.Active(1)) mindx = std::fmin(mindx, level_dx(s, finest, 1));
  if (s->mesh.Active(2)) mindx = std::fmin(mindx, level_dx(s, finest, 2));
  double mins[3] = {mindx, s->pkg.dt_hyp, kHuge};
  // (the cell widths are the same on every rank -- the forest is replicated -- and estimate_timestep
  // has reduced dt_hyp already)
  if (s->have_comm && s->nranks > 1 && !s->dt_hyp_is_global) {
    if (s->comm.allreduce_min(s->comm.user, mins, 3) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_min failed");
  }
  s->pkg.mindx = mins[0];
  s->pkg.dt_hyp = mins[1];
  s->pkg.c_h = s->pkg.cfl * s->pkg.mindx / s->pkg.dt_hyp;
  return APK_OK;
}

// acc field, per-block phase tables (FewModesFT::SetPhases, few_modes_ft.cpp:142-195) and the
// device descriptor of the driver
int turbulence_device_setup(apk_sim *s) {
  const Mesh &m = s->mesh;
  const int nlb = (int)m.local_gids.size();
  const int M = s->fmft->num_modes();
  const size_t acc_per = 3 * (size_t)m.sn;
  const size_t ph_per = (size_t)(m.mb[0] + m.mb[1] + m.mb[2]) * M * 2;
  SIM_TRY(s, dev_alloc(s, "acc", acc_per * nlb * sizeof(double), &s->d_acc));
  SIM_TRY(s, dev_alloc(s, "turbulence_phases", ph_per * nlb * sizeof(double), &s->d_phases));
  SIM_HIP(s, hipMemset(s->d_acc, 0, acc_per * nlb * sizeof(double)));
  std::vector<double> ph(ph_per * nlb);
  std::vector<apk_fmft_block> desc(nlb);
  for (int lb = 0; lb < nlb; ++lb) {
    int bc[3];
    m.Loc(m.local_gids[lb], bc);
    double *h = ph.data() + ph_per * lb;
    double *d = s->d_phases + ph_per * lb;
    size_t off = 0;
    const double *dptr[3];
    for (int ax = 0; ax < 3; ++ax) {
      s->fmft->Phases(ax, m.mb[ax], bc[ax] * m.mb[ax], m.nx[ax], h + off);
      dptr[ax] = d + off;
      off += (size_t)m.mb[ax] * M * 2;
    }
    desc[lb].acc = s->d_acc + acc_per * lb;
    desc[lb].phases_i = dptr[0];
    desc[lb].phases_j = dptr[1];
    desc[lb].phases_k = dptr[2];
  }
  SIM_HIP(s, hipMemcpy(s->d_phases, ph.data(), ph.size() * sizeof(double), hipMemcpyHostToDevice));
  SIM_TRY(s, apk_fmft_create(s->ctx, desc.data(), nlb, M, &s->fm_dev));
  return APK_OK;
}

// turbulence::Driving = Generate + Perturb (src/pgen/turbulence.cpp:373-482), the first-order
// operator-split source run after the last stage (src/hydro/hydro_driver.cpp:559-560)
int turbulence_driving(apk_sim *s, double dt, bool fill, bool no_prim) {
  s->fmft->Evolve(dt);
  const auto &vh = s->fmft->var_hat();
  std::vector<double> flat(vh.size() * 2);
  for (size_t q = 0; q < vh.size(); ++q) {
    flat[2 * q] = vh[q].real();
    flat[2 * q + 1] = vh[q].imag();
  }
  SIM_TRY(s, apk_fmft_inverse(s->ctx, s->mu0(), s->fm_dev, flat.data(), s->stream));
  double sums[4];
  SIM_TRY(s, apk_turb_mean_momentum(s->ctx, s->mu0(), s->fm_dev, sums, s->stream));  // synchronises: flat is free
  const bool mpi = s->have_comm && s->nranks > 1;
  if (mpi && s->comm.allreduce_sum(s->comm.user, sums, 4) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_sum failed");
  double ampl = 0.0;
  SIM_TRY(s, apk_turb_remove_mean(s->ctx, s->mu0(), s->fm_dev, sums, &ampl, s->stream));
  if (mpi && s->comm.allreduce_sum(s->comm.user, &ampl, 1) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_sum failed");
  const double box = (s->xmax[0] - s->xmin[0]) * (s->xmax[1] - s->xmin[1]) * (s->xmax[2] - s->xmin[2]);
  const double norm = s->accel_rms / std::sqrt(ampl / box);
  // (fill: the kick also does FillDerived and the time-step estimate of the cells it touches -- the two tasks that
  // follow it, hydro_driver.cpp:559-577, 589-603 -- instead of a full ConsToPrim pass and a dt pass afterwards)
  // (no_prim: the stages of this cycle stored no primitives and the next one derives its input from the conserved state --
  // rk_prim_free_cycle --: the kick estimates the time step and leaves the primitives where they are, stale)
  if (fill && no_prim) {
    SIM_TRY(s, apk_turb_apply_dt(s->ctx, s->mu0(), s->fm_dev, norm, dt, s->pkg.fluid, &s->pkg.eos, s->stream));
    s->turb_dt_kicks += 1;
  }
  else if (fill) SIM_TRY(s, apk_turb_apply_fill(s->ctx, s->mu0(), s->fm_dev, norm, dt, s->pkg.fluid, &s->pkg.eos, s->pkg.calc_dt_hyp ? 1 : 0, s->stream));
  else SIM_TRY(s, apk_turb_apply(s->ctx, s->mu0(), s->fm_dev, norm, dt, s->stream));
  return APK_OK;
}


// one stage of HydroDriver::MakeTaskCollection (hydro_driver.cpp:474-577)

int do_stage(apk_sim *s, int stage) {
  HydroPackage &pkg = s->pkg;
  const double g0 = s->gam0[stage - 1], g1 = s->gam1[stage - 1];
  const double beta_dt = s->beta[stage - 1] * s->dt;
  const size_t field_bytes = (size_t)s->nper * s->mesh.local_gids.size() * sizeof(double);
  // (a stage form that reads ghost zones after stages that did not fill the same-rank ones)
  const bool direct = direct_neighbors(s);
  if (!direct && s->local_ghosts_stale) SIM_TRY(s, sync_ghosts(s));
  // (the full-step primitives were not stored: only the donor-cell predictor can do without them)
  const bool prim_free = prim_free_cycle(s);
  const bool rk_free = rk_prim_free_cycle(s);
  // (refined meshes, amr_prim_free_cycle: both stages read the conserved state whether or not primitives are stored --
  // one set of kernels whatever happened between the cycles)
  const bool amr_pf = amr_prim_free_cycle(s);
  const bool from_cons = amr_pf || (s->prim_stale && ((stage == 1 && prim_free) || rk_free));
  if (s->prim_stale && !from_cons) SIM_TRY(s, sync_ghosts(s));
  if (amr_pf) s->amr_tag_vars_stored = s->amr_tags_posted = false;  // (the state they were taken from is about to be replaced)
  // (ghost zones one layer deep: enough for the donor-cell predictor they were left for, and for nothing else)
  if (s->remote_ghosts_thin && !(stage == 1 && thin_exchange_cycle(s))) SIM_TRY(s, sync_ghosts(s));
  // (... and their x1 strips still in the receive buffers: for a predictor that reads them there, x1_direct_cycle)
  if ((s->exchange_pending ? s->xchg_x1_direct : s->x1_in_recv) && !x1_direct_cycle(s)) SIM_TRY(s, sync_ghosts(s));
  if (stage == 1) {
    // "init u1" (hydro_driver.cpp:474-495) without the copy: the buffer holding u0 becomes the
    // register u1 and the stage writes the new u0 into the other buffer.  Valid because
    // gam0[0] == 0 for rk1/rk2/vl2/rk3, i.e. stage 1 never reads the old contents of its output.
    if (g0 != 0.0) {
      SIM_HIP(s, hipMemcpyAsync(s->d_cons2[s->u1buf], s->d_cons2[s->cur], field_bytes, hipMemcpyDeviceToDevice, hs(s)));
    }
    {
      const int was_u1 = s->u1buf;
      s->u1buf = s->cur;
      s->cur = was_u1;
    }
  }
  const apk_flux_cfg cfg = (stage == 1) ? pkg.flux_first_stage : pkg.flux_other_stage;
  bool fused_fill = false;
  bool ghost_cons_dead = false;  // the conserved values of this stage's result are read in no ghost zone
  s->stage_dt_pending = false;

  // an exchange left in flight is completed inside the fused stage below; anything else first
  if (s->exchange_pending && !can_overlap_next(s, stage)) SIM_TRY(s, finish_pending(s));
  if (stage_can_fuse(s)) {
    apk_stage_args a{};
    a.cfg = cfg;
    a.eos = pkg.eos;
    a.c_h = pkg.c_h;
    a.gam0 = g0;
    a.gam1 = g1;
    a.beta_dt = beta_dt;
    a.dedner = (pkg.fluid == APK_FLUID_GLMMHD) ? (pkg.glmmhd_source_extended ? 2 : 1) : 0;
    a.glmmhd_alpha = pkg.glmmhd_alpha;
    a.mindx = pkg.mindx;
    a.face_neighbor = (direct || amr_direct(s)) ? s->d_face_nbr : nullptr;
    // let the finishing sweep do FillDerived (and, in the last stage, the dt estimate) on the
    // cells it updates; only the ghost zones are converted after the exchange
    // (not when the turbulence driver kicks the state after this stage)
    // nor in a 3-D donor-cell stage (the VL2 predictor): its single-march kernel leaves prim
    // untouched and the full ConservedToPrimitive pass is cheaper than the du round trip it avoids
    // nor on refined meshes (the flux correction changes cells after the stage; the full pass after
    // the multilevel exchange converts everything)
    fused_fill = (s->mesh.ndim >= 2) && !(s->fmft && stage == s->nstages) && !s->amr;
    // A 3-D donor-cell stage (the VL2 predictor) runs as ONE march whose lanes read their
    // neighbours' primitives from memory, so it cannot replace prim in place: it writes the new
    // primitives into the spare buffer ("u1.prim") and the two prim buffers swap roles.
    static const int dc_mode = std::getenv("APK_DC_MODE") ? std::atoi(std::getenv("APK_DC_MODE")) : 2;  // A/B switch
    const bool dc3 = cfg.recon == APK_RC_DC && s->mesh.ndim == 3 && dc_mode != 0;
    bool swap_prim = false;
    // waves of the finishing march if it cannot be cut into segments (an in-place ConsToPrim forbids
    // that): lanes along x1, one wave per transverse row (or 2 / 4 rows for narrow blocks)
    const Mesh &mm = s->mesh;
    const int rpw_est = (mm.mb[0] <= 16) ? 4 : ((mm.mb[0] <= 32) ? 2 : 1);
    const int64_t final_waves = (int64_t)((mm.mb[0] + 64 / rpw_est - 1) / (64 / rpw_est)) *
                                ((mm.ndim == 3 ? mm.mb[1] : 1) + rpw_est - 1) / rpw_est * (int64_t)mm.local_gids.size();
    const bool few_waves = mm.ndim >= 2 && final_waves < 2048;
    // (the two-kernel 3-D stage: its finishing march reads x1 neighbours from memory -- out of place)
    const bool two_kernel = mm.ndim == 3 && cfg.recon != APK_RC_DC && apk_stage_split_axis(s->mu0(), &cfg, 2) == 3;
    if (fused_fill && ((dc3 && dc_mode == 2) || a.dedner == 2 || few_waves || two_kernel)) {
      // (the extended Dedner source reads neighbouring primitives as well: out of place, too; and a
      // finishing march with too few waves to fill the GPU -- 2-D meshes, small packs -- runs out of
      // place so that it can be cut into segments)
      SIM_TRY(s, ensure_spare_prim(s));
      swap_prim = true;
    } else if (dc3) {
      fused_fill = false;  // dc_mode 1: single march, separate full ConservedToPrimitive
    }
    a.fill_derived = fused_fill ? (swap_prim ? 2 : 1) : 0;
    a.estimate_dt = (fused_fill && stage == s->nstages && pkg.calc_dt_hyp) ? 1 : 0;
    a.prim_from_cons = from_cons ? 1 : 0;
    // the last stage of a cycle whose successor's predictor reads the conserved state: primitives for the dt estimate only
    const bool no_prim = (prim_free && stage == s->nstages && two_kernel && swap_prim && a.estimate_dt) || (rk_free && two_kernel && fused_fill);
    if (no_prim) a.fill_derived = a.estimate_dt ? 3 : 0;  // (rk_free: stages that are not the last compute no primitives at all)
    int outbuf = s->cur;
    if (rk_free && from_cons && g0 != 0.0) {
      // the input is the state this stage updates: its result goes to the free buffer, which becomes the current one
      SIM_TRY(s, ensure_trial_cons(s));
      outbuf = s->freebuf();
      a.prim_from_cons = 2;
      a.cons_out_delta = s->d_cons2[outbuf] - s->d_cons2[s->cur];
    }
    // (refined meshes: the corrector from the half-step conserved state, over u1 -- amr_prim_free_cycle)
    const bool amr_fc = amr_pf && stage == 2 && two_kernel && a.fill_derived == 0;
    if (amr_pf && stage == 2 && !amr_fc) return fail(s, APK_ERR_INVALID, "do_stage: the corrector of a refined mesh's prim-free cycle is not the two-kernel stage");
    if (amr_fc) {
      outbuf = s->u1buf;
      a.prim_from_cons = 2;
      a.cons_out_delta = s->d_cons2[outbuf] - s->d_cons2[s->cur];
    }
    {
      // The predictor of VL2: the corrector has gam0 = 0 and takes its fluxes from the predictor's primitives, so the
      // half-step CONSERVED state is read by nobody but the ghost exchange -- the nghost-deep shell of every block --
      // and by nothing at all when every face is crossed through the face table (apk_stage_args.cons_store).
      const bool dead = dc3 && swap_prim && stage < s->nstages && s->gam0[stage] == 0.0 && !s->amr && !s->fmft && pkg.nscalars == 0;
      bool all_periodic = true;
      for (int d = 0; d < 3; ++d)
        if (mm.Active(d) && (mm.bc_in[d] != BC_PERIODIC || mm.bc_out[d] != BC_PERIODIC)) all_periodic = false;
      // (physical boundary phases copy conserved values out of ghost zones filled before them: periodic boxes only)
      // On a periodic box the exchange after this stage moves the stored primitives themselves (GHOST_PRIM_COPY below;
      // floors / ceilings keep the unfused order copy, then ConsToPrim of the ghost zones, which reads the shell):
      // then nothing reads any conserved value of this stage's result.
      if (dead) a.cons_store = (all_periodic && ((direct && mm.peers.empty()) || ghost_c2p_fusable(s))) ? 2 : 1;
      ghost_cons_dead = a.cons_store != 0 && all_periodic;
    }
    if (s->exchange_pending && cfg.recon == APK_RC_DC && !(dc3 && swap_prim)) SIM_TRY(s, finish_pending(s));
    {
      // The single-march donor-cell stage split into windows (one main window + six slabs) runs one row per lane and
      // seven launches; whole, it runs two rows per lane (3.5 Riemann problems per cell instead of 4).  The one-GPU
      // rehearsal of an 8-GPU rank (bench.py) measures the split at +0.3 ms per cycle against 0.19 ms of wire time it
      // could hide: the exchange in flight at the start of a cycle is completed before the predictor instead
      // (round 3 split it: measured slower).
      if (s->exchange_pending && dc3 && swap_prim) SIM_TRY(s, finish_pending(s));
    }
    // x1 strips straight into / from the exchange buffers (x1_direct_cycle): the predictor sends its primitives nghost
    // deep and reads the one-layer conserved strips the corrector of the cycle before sent; the corrector the other way
    // round.  The receive side only when the exchange this stage follows left the strips in the buffers.
    apk_x1_halo x1h{};
    {
      const int x1kind = x1_direct_kind(s);
      const bool from_buffers = s->exchange_pending ? s->xchg_x1_direct : s->x1_in_recv;
      if (x1kind == 1) {
        const bool predictor = stage == 1 && dc3 && swap_prim && ghost_cons_dead && a.cons_store == 2;
        const bool corrector = stage == s->nstages && stage > 1 && two_kernel && no_prim;
        if (predictor || corrector) {
          x1h.blocks = static_cast<const apk_x1_halo_block *>(s->d_x1_tab[predictor ? 0 : 1]);
          x1h.recv_depth = from_buffers ? (predictor ? kThinDepth : s->mesh.ng) : 0;
          x1h.send_depth = predictor ? s->mesh.ng : kThinDepth;
          x1h.send_field = predictor ? 1 : 0;
          a.x1_halo = &x1h;
        }
      } else if (x1kind == 2 && two_kernel && rk_free && from_cons && (a.fill_derived == 0 || a.fill_derived == 3)) {
        // (an RK stage from the conserved state: the full messages both ways, the conserved state nghost deep)
        x1h.blocks = static_cast<const apk_x1_halo_block *>(s->d_x1_tab[2]);
        x1h.recv_depth = from_buffers ? s->mesh.ng : 0;
        // (forced turbulence: the kick after the last stage changes the state the strips were taken from -- that exchange
        // packs its x1 faces again)
        x1h.send_depth = (s->fmft && stage == s->nstages) ? 0 : s->mesh.ng;
        x1h.send_field = 0;
        if (x1h.recv_depth > 0 || x1h.send_depth > 0) a.x1_halo = &x1h;
      }
    }
    if (s->x1_in_recv && !(a.x1_halo && x1h.recv_depth > 0))
      return fail(s, APK_ERR_INVALID, "do_stage: x1 ghost columns were left in the receive buffers for a stage that does not read them there");
    if (s->exchange_pending) {
      // The previous stage's halo messages are still in flight.  Ghost zones filled by same-rank
      // copies are ready: convert them, run whatever does not touch a late face (the x1 sweep of
      // a high-order stage / the whole single-kernel donor-cell stage, on index windows), then
      // complete the exchange and do the thin slabs next to those faces and the rest.
      apk_pack *state = s->mu0_of[s->pending_cons][s->pcur];  // (stage 1 has swapped the cons roles already)
      const int c2p_in_copy = s->pending_c2p;  // then the ghost zones are converted as they are filled
      const bool convert_ghosts = !c2p_in_copy && !s->prim_stale;  // (stale: the predictor reads the conserved state)
      if (convert_ghosts)
        SIM_TRY(s, apk_cons_to_prim_ghosts_split(s->ctx, state, pkg.fluid, &pkg.eos, s->d_late_regions, 1, s->stream));
      const bool whole = dc3 && swap_prim;  // single-kernel stage
      const bool planes = !whole && apk_stage_split_axis(s->mu0(), &cfg, a.fill_derived) == 3;
      const apk_sim::WindowTable *tabs = whole ? s->dcwin : (planes ? s->k3win : s->x1win);
      const int ntabs = whole ? 7 : 3;
      a.phase = 1;
      for (int q = 0; q < ntabs; ++q) {
        if (q == 1) {
          SIM_TRY(s, exchange_end(s, c2p_in_copy));
          if (convert_ghosts)
            SIM_TRY(s, apk_cons_to_prim_ghosts_split(s->ctx, state, pkg.fluid, &pkg.eos, s->d_late_regions, 2, s->stream));
        }
        if (!tabs[q].any) continue;
        a.window = tabs[q].d;
        a.window_rl = tabs[q].rl;
        a.window_rows = tabs[q].rows;
        SIM_TRY(s, apk_stage_fused(s->ctx, s->mu0(), s->mu1(), &a, s->stream));
      }
      a.phase = 2;
      a.window = nullptr;
      a.window_rl = a.window_rows = 0;
      s->overlapped += 1;
    }
    // (refined meshes: the flux correction's boundary-plane fluxes beside the stage -- the stage stores no primitives there)
    bool planes_ahead = false;
    if (s->amr && a.fill_derived == 0) {
      SIM_TRY(s, ensure_flux_arrays(s));
      // (amr_pf: the planes from the conserved state the stage reads -- u1's buffer in stage 1, the current one in stage 2)
      planes_ahead = amr_flux_planes_ahead(s, cfg, amr_pf ? (stage == 1 ? s->u1buf : s->cur) : -1);
    }
    {
      const int rc_stage = apk_stage_fused(s->ctx, s->mu0(), s->mu1(), &a, s->stream);
      // (the boundary-plane fluxes forked onto the side stream are joined on the error path too)
      if (rc_stage != APK_OK && planes_ahead) (void)hipStreamWaitEvent(hs(s), reinterpret_cast<hipEvent_t>(s->ev_join), 0);
      SIM_TRY(s, rc_stage);
    }
    s->stage_dt_pending = a.estimate_dt != 0;
    s->x1_in_recv = false;                                // (read; the exchange below starts afresh)
    s->x1_out_direct = a.x1_halo && x1h.send_depth > 0;  // (consumed by exchange_begin)
    // (a stage that stored primitives -- the predictor's half-step ones -- makes the current buffer valid again; one that
    // stored none leaves them stale: the stages of a prim-free RK cycle, and its last stage under the turbulence driver,
    // whose kick then estimates the time step without storing them either)
    if (no_prim) s->prim_stale = true;
    else if (a.fill_derived == 1 || a.fill_derived == 2) {
      s->prim_stale = false;
      if (swap_prim) s->pcur = 1 - s->pcur;
    }
    {
      const int inbuf = s->cur;
      s->cur = outbuf;  // (its ghost zones are filled by the exchange below)
      if (amr_fc) s->u1buf = inbuf;  // (the half-step state: scratch from here on)
      if (s->amr) {
        SIM_TRY(s, ensure_flux_arrays(s));
        const double psi_factor = a.dedner != 0 ? std::exp(-pkg.glmmhd_alpha * pkg.c_h * beta_dt / pkg.mindx) : 1.0;
        // (boundary planes not computed beside the stage: from the state the stage read -- now the register's buffer)
        SIM_TRY(s, amr_flux_fix(s, cfg, beta_dt, psi_factor, planes_ahead, amr_pf ? (stage == 1 ? s->u1buf : inbuf) : -1));
      }
    }
  } else {
    // first_order_flux_correct and a stage that does not read the old u0 (gam0 = 0: every VL2 stage,
    // the first stage of the others): run the fused stage optimistically -- it leaves its inputs
    // (prim, u1) intact -- and test the new state the way FirstOrderFluxCorrect tests its trial
    // update.  No cell fails (the rule, away from strong shocks): done, with the result the
    // flux-array sequence would have produced bit for bit.  Otherwise that sequence runs after all.
    // The optimistic stage also does FillDerived (out of place: the old primitives are the fallback's
    // input) and, in the last stage, the dt estimate, exactly like a stage without flux correction.
    bool done = false;
    // (Refined meshes: the test sees the update before the coarse-fine flux correction, as
    // FirstOrderFluxCorrect does in the reference's task order; the correction follows, and ConsToPrim
    // stays the full pass after the exchange.)
    // A stage that does read the old u0 (gam0 != 0: the later stages of RK2 / RK3) writes its trial
    // result into a third buffer instead, so that u0 survives a rejected trial; an accepted one makes
    // that buffer the current state.  (Not with passive scalars: their kernel updates in place; not on
    // refined meshes: the flux correction after the stage addresses the current buffer.)
    // (With the face table, so does a later stage that does not read it -- VL2's corrector: a rejected trial is redone
    // through the flux arrays, whose sweeps read ghost primitives the table-following stages left stale, and those are
    // regenerated from the conserved state they belong to -- which an in-place trial would have overwritten.)
    const bool trial_out_of_place = g0 != 0.0 || (direct && stage > 1);
    if (s->fused && pkg.first_order_flux_correct && (g0 == 0.0 || (pkg.nscalars == 0 && !s->amr)) &&
        !pkg.glmmhd_source_extended && s->mesh.ndim >= 2 && pkg.riemann != APK_RS_NONE && pkg.riemann != APK_RS_LLF) {
      if (trial_out_of_place) SIM_TRY(s, ensure_trial_cons(s));
      // FirstOrderFluxCorrect tests the UNfloored trial update (hydro.cpp:1283-1306; floors only act
      // in the ConsToPrim that follows the stage)
      // (the finishing sweep tests the update it holds in registers, before its own ConsToPrim floors
      // it; with passive scalars the stored state is tested after the stage, so nothing may floor it)
      const bool test_in_kernel = pkg.nscalars == 0;
      const bool fill = !(s->fmft && stage == s->nstages) && !s->amr && (test_in_kernel || ghost_c2p_fusable(s));
      if (fill) SIM_TRY(s, ensure_spare_prim(s));
      apk_stage_args a{};
      a.cfg = cfg;
      a.eos = pkg.eos;
      a.c_h = pkg.c_h;
      a.gam0 = g0;
      a.gam1 = g1;
      a.beta_dt = beta_dt;
      a.dedner = (pkg.fluid == APK_FLUID_GLMMHD) ? 1 : 0;
      a.glmmhd_alpha = pkg.glmmhd_alpha;
      a.mindx = pkg.mindx;
      a.fill_derived = fill ? 2 : 0;
      a.estimate_dt = (fill && stage == s->nstages && pkg.calc_dt_hyp) ? 1 : 0;
      a.face_neighbor = direct ? s->d_face_nbr : nullptr;
      a.trial = 1;  // its ConsToPrim latches flags into the trial word: kept or dropped below
      // the finishing sweep applies FirstOrderFluxCorrect's test to the update it has in registers
      // (passive scalars ride a separate kernel: there the stored state is tested afterwards)
      a.count_unphysical = test_in_kernel ? 1 : 0;
      const int outbuf = trial_out_of_place ? s->freebuf() : s->cur;
      a.cons_out_delta = s->d_cons2[outbuf] - s->d_cons2[s->cur];
      SIM_TRY(s, apk_stage_fused(s->ctx, s->mu0(), s->mu1(), &a, s->stream));
      long long bad = 0;
      if (a.count_unphysical) SIM_TRY(s, apk_stage_unphysical_read(s->ctx, &bad, s->stream));
      else SIM_TRY(s, apk_count_unphysical(s->ctx, s->mu0(), pkg.fluid, &bad, s->stream));
      done = bad == 0;
      if (fill) SIM_TRY(s, apk_trial_flags(s->ctx, done ? 1 : 0, s->stream));
      if (done) {
        s->cur = outbuf;  // (its ghost zones are filled by the exchange below)
        if (fill) {
          s->pcur = 1 - s->pcur;
          fused_fill = true;
        }
        s->stage_dt_pending = a.estimate_dt != 0;
        if (s->amr) {
          SIM_TRY(s, ensure_flux_arrays(s));
          const double psi_factor = a.dedner != 0 ? std::exp(-pkg.glmmhd_alpha * pkg.c_h * beta_dt / pkg.mindx) : 1.0;
          SIM_TRY(s, amr_flux_fix(s, cfg, beta_dt, psi_factor));
        }
      } else {
        s->fofc_fallback_stages += 1;
      }
    }
    if (!done) {
    // (the flux arrays' sweeps and FirstOrderFluxCorrect read the primitives of ghost cells: after stages that followed
    // the face table those are stale -- fill them, from the conserved buffer the stored primitives belong to: the
    // register u1 in stage 1, whose buffers have swapped roles above, the current state otherwise)
    if (s->local_ghosts_stale) SIM_TRY(s, materialize_local_ghosts(s, stage == 1 ? s->u1buf : s->cur));
    SIM_TRY(s, ensure_flux_arrays(s));
    // (faces of interior cells only: nothing downstream reads the reference's extra transverse rows)
    SIM_TRY(s, apk_calculate_fluxes_tight(s->ctx, s->mu0(), cfg, &pkg.eos, pkg.c_h, s->stream));
    if (pkg.first_order_flux_correct) {
      long long nfix = 0;
      SIM_TRY(s, apk_first_order_flux_correct(s->ctx, s->mu0(), s->mu1(), pkg.fluid, &pkg.eos, pkg.c_h, g0, g1,
                                              beta_dt, &nfix, s->stream));
      s->fofc_total += nfix;
    }
    if (s->amr) SIM_TRY(s, amr_flux_correction(s));
    SIM_TRY(s, apk_update_with_flux_divergence(s->ctx, s->mu0(), s->mu1(), g0, g1, beta_dt, s->stream));
    if (pkg.fluid == APK_FLUID_GLMMHD) {
      SIM_TRY(s, apk_dedner_source(s->ctx, s->mu0(), pkg.glmmhd_source_extended ? 1 : 0, pkg.glmmhd_alpha,
                                   pkg.c_h, pkg.mindx, beta_dt, s->stream));
    }
    }
  }
  if (s->fmft && stage == s->nstages) {
    const bool kick_fills = !fused_fill && !s->amr;
    // (a cycle whose stages store no primitives: the kick leaves them stale too, the next stage 1 reads the conserved state)
    const bool kick_no_prim = kick_fills && s->prim_stale && rk_prim_free_cycle(s);
    SIM_TRY(s, turbulence_driving(s, s->dt, kick_fills, kick_no_prim));
    if (kick_fills && !kick_no_prim) s->prim_stale = false;  // (the kick wrote the primitives of every cell it touched)
    if (kick_fills) {  // as after a stage whose finishing sweep did FillDerived and the dt estimate
      fused_fill = true;
      s->stage_dt_pending = pkg.calc_dt_hyp;
    }
  }
  // (a last stage that stored no primitives: the ghost zones get none either -- the next predictor reads the conserved
  // state there as everywhere)
  const bool ghost_prims = !s->prim_stale;
  const int c2p_in_copy = !(fused_fill && ghost_c2p_fusable(s) && ghost_prims)
                              ? GHOST_COPY
                              : (ghost_cons_dead ? GHOST_PRIM_COPY : GHOST_C2P);
  if (fused_fill && can_overlap_next(s, stage < s->nstages ? stage + 1 : 1)) {
    // post the messages and leave them in flight: the next stage (of this or of the next cycle)
    // completes the exchange
    SIM_TRY(s, exchange_begin(s, true, c2p_in_copy, direct, stage == s->nstages && thin_exchange_cycle(s)));
  } else if (s->amr && amr_faces_only(s) && !(stage == s->nstages && regrid_check_follows(s))) {
    // refined meshes: nothing in the stage loop reads a ghost cell behind an edge or a corner of a block -- the
    // exchange skips those boxes (37 % of the ghost cells of a 16^3 block with nghost = 4) and ConsToPrim the cells
    // (nor, with amr_direct, the ghost zones behind faces the stages cross by the face table)
    const bool dir = amr_direct(s);
    SIM_TRY(s, amr_exchange(s, s->cur, dir ? AMR_XCHG_DIRECT : AMR_XCHG_FACES));
    if (dir) s->skipped_local_exchanges += 1;
    if (stage == s->nstages && pkg.calc_dt_hyp && amr_pf) {
      // (the next predictor reads the conserved state: the estimate alone, no primitive stored)
      SIM_TRY(s, apk_cons_to_prim_dt_select(s->ctx, s->mu0(), pkg.fluid, &pkg.eos, 0, nullptr, 0u, s->stream));
      s->stage_dt_pending = true;
      s->prim_stale = true;
      s->amr_c2p_passes_skipped += 1;
    } else if (stage == s->nstages && pkg.calc_dt_hyp) {  // (the time-step estimate on the way, as below)
      SIM_TRY(s, apk_cons_to_prim_faces_dt(s->ctx, s->mu0(), pkg.fluid, &pkg.eos, dir ? s->d_face_nbr : nullptr, s->stream));
      s->stage_dt_pending = true;
    } else if (stage == 1 && amr_pf) {
      // (the corrector converts what it loads: no pass over the blocks here)
      s->amr_c2p_passes_skipped += 1;
    } else if (dir) {
      SIM_TRY(s, apk_cons_to_prim_faces_skip(s->ctx, s->mu0(), pkg.fluid, &pkg.eos, s->d_face_nbr, s->stream));
    } else {
      SIM_TRY(s, apk_cons_to_prim_faces(s->ctx, s->mu0(), pkg.fluid, &pkg.eos, s->stream));
    }
  } else if (s->amr && stage == s->nstages && amr_shell_before_check(s)) {
    // the last exchange of a cycle that ends with a refinement check: every ghost zone, but only as deep as the tagging
    // criteria and the first stage of the next cycle read; ConsToPrim of that shell with the time-step estimate
    // (with the face table: nor the zones behind same-level same-rank faces, 59 % of the shell's cells -- the tag kernel,
    // ConsToPrim and the next cycle's predictor all follow the table there)
    // (periodic boxes only: a physical-boundary phase copies the edge cells next to the boundary out of ghost zones
    // filled before it, and the tagging criteria read those edges)
    bool dir = amr_direct(s);
    for (int d = 0; d < 3; ++d)
      if (s->mesh.Active(d) && (s->mesh.bc_in[d] != BC_PERIODIC || s->mesh.bc_out[d] != BC_PERIODIC)) dir = false;
    SIM_TRY(s, amr_exchange(s, s->cur, dir ? AMR_XCHG_SHELL_DIRECT : AMR_XCHG_SHELL));
    if (dir) s->skipped_local_exchanges += 1;
    int crit = -1;
    double crit_p0 = 0.0, crit_p1 = 0.0;
    if (amr_pf && pkg.calc_dt_hyp) SIM_TRY(s, refinement_criterion(s, &crit, &crit_p0, &crit_p1));
    static const int fused_tag = std::getenv("APK_AMR_FUSED_TAG") ? std::atoi(std::getenv("APK_AMR_FUSED_TAG")) : 1;  // A/B switch
    bool tags_done = false;
    if (crit == APK_TAG_PRESSURE_GRADIENT && fused_tag && s->mesh.ndim == 3) {
      // (... and for the pressure gradient not even that: the criterion is reduced in the same pass, its pressures in LDS)
      int pending = 0;
      const int rc_tag = apk_tag_blocks_dt_from_cons(s->ctx, s->mu0(), pkg.fluid, &pkg.eos, dir ? s->d_face_nbr : nullptr, &pending, s->stream);
      if (rc_tag == APK_OK) {
        tags_done = true;
        s->amr_tags_posted = true;
        s->amr_posted_criterion = crit, s->amr_posted_pending = pending, s->amr_posted_p0 = crit_p0, s->amr_posted_p1 = crit_p1;
        s->prim_stale = true;
        s->amr_tag_vars_stored = false;
        s->amr_c2p_passes_skipped += 1;
      } else if (rc_tag != APK_ERR_UNSUPPORTED) {  // (blocks too wide for the pressure tile: the two passes below)
        return fail(s, rc_tag, std::string("apk_tag_blocks_dt_from_cons: ") + apk_last_error(s->ctx));
      }
    }
    if (tags_done) {
    } else if (crit >= 0) {
      // (the next predictor reads the conserved state: of the primitives only what the refinement criterion reads)
      // (the reference's order of the primitives: IDN = 0, IV1 .. IV3 = 1 .. 3, IPR = 4)
      const unsigned vars = crit == APK_TAG_PRESSURE_GRADIENT ? (1u << 4) : (crit == APK_TAG_VELOCITY_GRADIENT ? ((1u << 1) | (1u << 2)) : (1u << 0));
      SIM_TRY(s, apk_cons_to_prim_dt_select(s->ctx, s->mu0(), pkg.fluid, &pkg.eos, AMR_SHELL_DEPTH, dir ? s->d_face_nbr : nullptr, vars, s->stream));
      s->prim_stale = true;
      s->amr_tag_vars_stored = true;
      s->amr_c2p_passes_skipped += 1;
    } else {
      SIM_TRY(s, apk_cons_to_prim_dt_skip(s->ctx, s->mu0(), pkg.fluid, &pkg.eos, AMR_SHELL_DEPTH, dir ? s->d_face_nbr : nullptr, s->stream));
    }
    s->stage_dt_pending = true;
  } else {
    // (without a fused FillDerived the full-block ConsToPrim below reads every ghost zone)
    SIM_TRY(s, exchange_ghosts(s, c2p_in_copy, direct && fused_fill, fused_fill && stage == s->nstages && thin_exchange_cycle(s)));
    if (fused_fill) {
      // (not after an exchange that filled nothing: direct addressing on a mesh whose faces the table covers)
      const bool filled_none = direct && table_covers_all_faces(s);
      if (!c2p_in_copy && ghost_prims && !filled_none) SIM_TRY(s, apk_cons_to_prim_ghosts(s->ctx, s->mu0(), pkg.fluid, &pkg.eos, s->stream));
    } else if (stage == s->nstages && pkg.calc_dt_hyp) {
      // the last FillDerived of the cycle and the time-step estimate that follows it (hydro_driver.cpp:571-603) in
      // one pass: the interior cells' primitives are in registers anyway (refined meshes, flux-array stages)
      SIM_TRY(s, apk_cons_to_prim_dt(s->ctx, s->mu0(), pkg.fluid, &pkg.eos, -1, s->stream));
      s->stage_dt_pending = true;
      s->prim_stale = false;  // (every cell of every block)
    } else {
      SIM_TRY(s, fill_derived(s));
      s->prim_stale = false;
    }
  }
  if (stage == s->nstages && pkg.calc_c_h) {  // hydro_driver.cpp:589-603
    pkg.mindx = kHuge;
    pkg.dt_hyp = kHuge;
    s->dt_hyp_is_global = false;
  }
  return APK_OK;
}

int create_common(const char *deck, const char *const *overrides, int noverrides, int rank, int nranks,
                  apk_sim **out, char *errbuf, size_t errlen) {
  if (!out || !deck) return APK_ERR_INVALID;
  *out = nullptr;
  apk_sim *s = new (std::nothrow) apk_sim();
  if (!s) return APK_ERR_INVALID;
  s->rank = rank;
  s->nranks = nranks;
  try {
    s->pin.LoadFromString(deck);
    for (int i = 0; i < noverrides; ++i) s->pin.ApplyOverride(overrides[i]);
    s->problem_id = s->pin.GetString("job", "problem_id");
    hydro_initialize(s);
    mesh_initialize(s);
    if (s->problem_id == "linear_wave") lw_setup(s);
    else if (s->problem_id == "linear_wave_mhd") lwm_setup(s);
    else if (s->problem_id == "cpaw") cpaw_setup(s);
    else if (s->problem_id == "field_loop") field_loop_setup(s);
    else if (s->problem_id == "kh") kh_setup(s);
    else if (s->problem_id == "advection") {
      // advection::InitUserMeshData (src/pgen/advection.cpp:34-59): tlim counts box diagonals / |v|
      const double vx = s->pin.GetOrAddReal("problem/advection", "vx", 0.0), vy = s->pin.GetOrAddReal("problem/advection", "vy", 0.0),
                   vz = s->pin.GetOrAddReal("problem/advection", "vz", 0.0);
      const double L[3] = {s->xmax[0] - s->xmin[0], s->xmax[1] - s->xmin[1], s->xmax[2] - s->xmin[2]};
      const double vmag = std::sqrt(vx * vx + vy * vy + vz * vz) + 1.0e-20;  // TINY_NUMBER
      const double diag = std::sqrt(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
      s->tlim = diag / vmag * s->tlim;
    }
    else if (s->problem_id == "lw_implode" && s->pkg.fluid == APK_FLUID_GLMMHD)
      throw std::runtime_error("Only hydro runs are supported for LW implosion problem generator.");
    else if (s->problem_id == "turbulence") turbulence_setup(s);
    else if (s->problem_id != "sod" && s->problem_id != "orszag_tang" && s->problem_id != "synthetic" &&
             s->problem_id != "blast" && s->problem_id != "lw_implode" && s->problem_id != "cpaw" &&
             s->problem_id != "advection" && s->problem_id != "field_loop" && s->problem_id != "kh" &&
             s->problem_id != "linear_wave" && s->problem_id != "linear_wave_mhd")
      throw std::runtime_error("unknown job/problem_id: " + s->problem_id);
    // src/bvals/boundary_conditions_apk.hpp:47-50 (raised when the wall is first applied, i.e. after
    // the problem generator's own checks): the wall only mirrors the normal momentum
    for (int d = 0; d < 3; ++d)
      if ((s->mesh.bc_in[d] == BC_REFLECT || s->mesh.bc_out[d] == BC_REFLECT) && s->pkg.fluid != APK_FLUID_EULER)
        throw std::runtime_error("Reflecting boundary conditions for MHD need special treatment.");
  } catch (const std::exception &e) {
    if (errbuf && errlen) std::snprintf(errbuf, errlen, "%s", e.what());
    delete s;
    return APK_ERR_INVALID;
  }
  *out = s;
  return APK_OK;
}


}  // namespace host
}  // namespace apk

using namespace apk::host;

extern "C" {

int apk_sim_create_host_only(const char *deck, const char *const *overrides, int noverrides, int rank,
                             int nranks, apk_sim **out, char *errbuf, size_t errlen) {
  int rc = create_common(deck, overrides, noverrides, rank, nranks, out, errbuf, errlen);
  if (rc == APK_OK) (*out)->host_only = true;
  return rc;
}

int apk_sim_create(const char *deck, const char *const *overrides, int noverrides, int rank, int nranks,
                   const apk_allocator *allocator, const apk_comm_ops *comm, apk_stream_t stream,
                   apk_sim **out, char *errbuf, size_t errlen) {
  int rc = create_common(deck, overrides, noverrides, rank, nranks, out, errbuf, errlen);
  if (rc != APK_OK) return rc;
  apk_sim *s = *out;
  auto bail = [&](int code) {
    if (errbuf && errlen) std::snprintf(errbuf, errlen, "%s", s->err.c_str());
    apk_sim_destroy(s);
    *out = nullptr;
    return code;
  };
  s->stream = stream;
  if (allocator && allocator->alloc) {
    s->alloc = *allocator;
    s->have_alloc = true;
  }
  if (comm) {
    s->comm = *comm;
    s->have_comm = true;
  }
  // (comm == NULL with nranks > 1: the native RCCL transport is attached by apk_sim_comm_rccl
  // before apk_sim_initialize, which checks)
  if (nranks > 1 && comm && !(comm->exchange && comm->allreduce_min && comm->allreduce_sum)) {
    s->err = "nranks > 1 requires comm ops";
    return bail(APK_ERR_INVALID);
  }
  if (s->mesh.rehearse && comm) {
    s->err = "apk_amd/rehearse_remote_faces brings its own (loopback) transport";
    return bail(APK_ERR_INVALID);
  }
  rc = apk_create(&s->ctx);
  if (rc != APK_OK) {
    s->err = "apk_create failed: no usable gfx950 device (there is no CPU fallback)";
    return bail(rc);
  }
  if (s->amr) {
    if ((rc = amr_allocate(s, s->mesh.local_gids.size(), s->d_cons2, &s->d_prim2[0], s->d_flux, &s->d_coarse)) != APK_OK) return bail(rc);
    if ((rc = amr_rebuild(s)) != APK_OK) return bail(rc);
    if ((rc = build_copy_plans(s)) != APK_OK) return bail(rc);  // (empty: the uniform-mesh plans are unused)
    return APK_OK;
  }
  const size_t nlb = s->mesh.local_gids.size();
  const size_t bytes = (size_t)s->nper * nlb * sizeof(double);
  if ((rc = dev_alloc(s, "cons", bytes, &s->d_cons2[0])) != APK_OK) return bail(rc);
  if ((rc = dev_alloc(s, "prim", bytes, &s->d_prim2[0])) != APK_OK) return bail(rc);
  if ((rc = dev_alloc(s, "u1", bytes, &s->d_cons2[1])) != APK_OK) return bail(rc);
  if (hipMemset(s->d_cons2[0], 0, bytes) != hipSuccess || hipMemset(s->d_prim2[0], 0, bytes) != hipSuccess ||
      hipMemset(s->d_cons2[1], 0, bytes) != hipSuccess) {
    s->err = "hipMemset failed";
    return bail(APK_ERR_DEVICE);
  }
  for (size_t p = 0; p < s->mesh.peers.size(); ++p) {
    double *sb = nullptr, *rb = nullptr;
    const std::string st = "send:" + std::to_string(s->mesh.peers[p].rank);
    const std::string rt = "recv:" + std::to_string(s->mesh.peers[p].rank);
    if ((rc = dev_alloc(s, st.c_str(), s->mesh.peers[p].send_count * sizeof(double), &sb)) != APK_OK) return bail(rc);
    if ((rc = dev_alloc(s, rt.c_str(), s->mesh.peers[p].recv_count * sizeof(double), &rb)) != APK_OK) return bail(rc);
    s->send_buf.push_back(sb);
    s->recv_buf.push_back(rb);
  }
  if (!stage_can_fuse(s)) {
    if ((rc = ensure_flux_arrays(s)) != APK_OK) return bail(rc);
  } else if ((rc = build_packs(s)) != APK_OK) {
    return bail(rc);
  }
  if ((rc = build_copy_plans(s)) != APK_OK) return bail(rc);
  // (Same-rank ghost copies on a second stream, overlapped with the part of the next stage that needs no ghost zone, were
  // measured in round 2 on 8 x 128^3 PPM+HLLD VL2: 5.73 ms per cycle against 5.24 -- the thin slab launches next to every
  // face and the copy kernel's share of the memory system cost more than the overlap hid; direct neighbour addressing
  // then removed the copies altogether.)
  if (s->mesh.rehearse && (rc = comm_loopback_attach(s)) != APK_OK) return bail(rc);
  if (!s->mesh.peers.empty() && (rc = build_windows(s)) != APK_OK) return bail(rc);
  if ((rc = build_x1_tables(s)) != APK_OK) return bail(rc);
  if (s->mesh.ndim == 3 && (rc = build_face_table(s)) != APK_OK) return bail(rc);
  if (s->fmft && (rc = turbulence_device_setup(s)) != APK_OK) return bail(rc);
  return APK_OK;
}

void apk_sim_destroy(apk_sim *s) {
  if (!s) return;
  if (!s->host_only) {
    if (s->exchange_pending && s->comm.exchange_end) (void)s->comm.exchange_end(s->comm.user);  // drain
    (void)hipDeviceSynchronize();
    rccl_transport_destroy(s->rccl);
    s->rccl = nullptr;
    for (auto &pp : s->pplans_of)
      for (auto &pl : pp)
        if (pl) apk_copy_plan_destroy(pl);
    for (auto &pp : s->plans_of)
      for (auto &p : pp) apk_copy_plan_destroy(p);
    for (int p = 0; p < 3; ++p)
      for (int w = 0; w < 2; ++w) {
        apk_pack_destroy(s->mu0_of[p][w]);
        apk_pack_destroy(s->mu1_of[p][w]);
      }
    apk_fmft_destroy(s->fm_dev);
    if (s->side_stream) (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(s->side_stream));
    if (s->ev_fork) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(s->ev_fork));
    if (s->ev_join) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(s->ev_join));
    if (s->amr) amr_destroy_device_plans(s);
    amr_free_buffers(s, s->amr_halo);
    amr_free_buffers(s, s->amr_halo_faces);
    amr_free_buffers(s, s->amr_halo_shell);
    amr_free_buffers(s, s->amr_fluxmsg);
    amr_free_buffers(s, s->amr_move);
    dev_free(s, s->d_coarse);
    for (auto &t : s->x1win) dev_free(s, reinterpret_cast<double *>(t.d));
    for (auto &t : s->dcwin) dev_free(s, reinterpret_cast<double *>(t.d));
    for (auto &t : s->k3win) dev_free(s, reinterpret_cast<double *>(t.d));
    dev_free(s, reinterpret_cast<double *>(s->d_late_regions));
    dev_free(s, reinterpret_cast<double *>(s->d_face_nbr));
    for (auto &t : s->d_x1_tab) dev_free(s, static_cast<double *>(t));
    dev_free(s, s->d_acc);
    dev_free(s, s->d_phases);
    dev_free(s, s->d_cons2[0]);
    dev_free(s, s->d_prim2[0]);
    dev_free(s, s->d_prim2[1]);
    dev_free(s, s->d_cons2[1]);
    dev_free(s, s->d_cons2[2]);
    for (auto *f : s->d_flux) dev_free(s, f);
    for (auto *b : s->send_buf) dev_free(s, b);
    for (auto *b : s->recv_buf) dev_free(s, b);
    apk_destroy(s->ctx);
  }
  delete s;
}

const char *apk_sim_last_error(const apk_sim *s) { return s ? s->err.c_str() : "null sim"; }

int apk_sim_set_fused(apk_sim *s, int fused) {
  if (!s) return APK_ERR_INVALID;
  if (!s->host_only) SIM_TRY(s, sync_ghosts(s));
  s->fused = fused != 0;
  if (!s->host_only && !stage_can_fuse(s)) return ensure_flux_arrays(s);
  return APK_OK;
}

int apk_sim_set_overlap(apk_sim *s, int overlap) {
  if (!s) return APK_ERR_INVALID;
  if (!s->host_only) SIM_TRY(s, sync_ghosts(s));
  s->overlap = overlap != 0;
  return APK_OK;
}

long long apk_sim_overlapped_exchanges(const apk_sim *s) { return s ? s->overlapped : 0; }
long long apk_sim_skipped_local_exchanges(const apk_sim *s) { return s ? s->skipped_local_exchanges : 0; }
long long apk_sim_amr_c2p_passes_skipped(const apk_sim *s) { return s ? s->amr_c2p_passes_skipped : 0; }
int apk_sim_set_direct_neighbors(apk_sim *s, int on) {
  if (!s) return APK_ERR_INVALID;
  if (!s->host_only) SIM_TRY(s, sync_ghosts(s));
  s->direct_on = on != 0;
  return APK_OK;
}
int apk_sim_set_prim_free(apk_sim *s, int on) {
  if (!s) return APK_ERR_INVALID;
  if (!s->host_only) SIM_TRY(s, sync_ghosts(s));
  s->prim_free_on = on != 0;
  return APK_OK;
}
int apk_sim_set_thin_exchange(apk_sim *s, int on) {
  if (!s) return APK_ERR_INVALID;
  s->thin_on = on != 0;
  return APK_OK;
}
long long apk_sim_thin_exchanges(const apk_sim *s) { return s ? s->thin_exchanges : 0; }
int apk_sim_set_x1_direct(apk_sim *s, int on) {
  if (!s) return APK_ERR_INVALID;
  s->x1_on = on != 0;
  return APK_OK;
}
long long apk_sim_x1_direct_exchanges(const apk_sim *s) { return s ? s->x1_direct_exchanges : 0; }
int apk_sim_prim_is_stale(const apk_sim *s) { return (s && s->prim_stale) ? 1 : 0; }
long long apk_sim_turb_dt_kicks(const apk_sim *s) { return s ? s->turb_dt_kicks : 0; }
int apk_sim_set_amr_full_exchange(apk_sim *s, int on) {
  if (!s) return APK_ERR_INVALID;
  if (!s->host_only) SIM_TRY(s, sync_ghosts(s));
  s->amr_full_exchange = on != 0;
  return APK_OK;
}
double apk_sim_loop_seconds(const apk_sim *s) { return s ? s->loop_seconds : 0.0; }
int apk_sim_loop_cycles(const apk_sim *s) { return s ? s->perf_cycles : 0; }
long long apk_sim_loop_zone_cycles(const apk_sim *s) { return s ? s->zone_cycles - s->perf_zone_mark : 0; }

int apk_sim_initialize(apk_sim *s) {
  if (!s || s->host_only) return APK_ERR_INVALID;
  s->err.clear();
  if (s->nranks > 1 && !(s->have_comm && s->comm.exchange && s->comm.allreduce_min && s->comm.allreduce_sum))
    return fail(s, APK_ERR_INVALID, "nranks > 1 requires comm ops (apk_sim_create) or the native transport (apk_sim_comm_rccl)");
  SIM_TRY(s, sync_ghosts(s));
  const int nlb = (int)s->mesh.local_gids.size();
  std::vector<double> host((size_t)s->nper);
  try {
    if (s->problem_id == "turbulence") {
      std::vector<std::vector<double>> blocks;
      SIM_TRY(s, pgen_turbulence(s, blocks));
      for (int lb = 0; lb < nlb; ++lb)
        SIM_HIP(s, hipMemcpy(s->blk(s->d_cons(), lb), blocks[lb].data(), sizeof(double) * s->nblk, hipMemcpyHostToDevice));
    } else {
      for (int lb = 0; lb < nlb; ++lb) {
        pgen_block(s, lb, host);
        SIM_HIP(s, hipMemcpy(s->blk(s->d_cons(), lb), host.data(), sizeof(double) * s->nblk, hipMemcpyHostToDevice));
      }
    }
  } catch (const std::exception &e) {
    return fail(s, APK_ERR_INVALID, e.what());
  }
  s->time = 0.0;
  s->ncycle = 0;
  s->dt = kHuge;
  s->fofc_total = 0;
  s->fofc_fallback_stages = 0;
  s->zone_cycles = 0;
  s->pkg.mindx = kHuge;
  s->pkg.dt_hyp = kHuge;
  SIM_TRY(s, exchange_ghosts(s));
  SIM_TRY(s, fill_derived(s));
  // adaptive meshes: tag the initial condition, refine, and evaluate the problem generator again on
  // the new blocks (not a prolongation), level by level
  for (int pass = 0; s->amr && s->amr_adaptive && pass < s->amr->max_level; ++pass) {
    std::vector<int> tags;
    SIM_TRY(s, amr_global_tags(s, tags));
    try {
      if (!amr_update_tree(s, tags, false)) break;
    } catch (const std::exception &e) {
      return fail(s, APK_ERR_INVALID, e.what());
    }
    SIM_TRY(s, amr_reallocate(s));
    try {
      for (int lb = 0; lb < (int)s->mesh.local_gids.size(); ++lb) {
        pgen_block(s, lb, host);
        SIM_HIP(s, hipMemcpy(s->blk(s->d_cons(), lb), host.data(), sizeof(double) * s->nblk, hipMemcpyHostToDevice));
      }
    } catch (const std::exception &e) {
      return fail(s, APK_ERR_INVALID, e.what());
    }
    SIM_TRY(s, exchange_ghosts(s));
    SIM_TRY(s, fill_derived(s));
  }
  double est = kHuge;
  SIM_TRY(s, estimate_timestep(s, &est));
  set_global_dt(s, est);
  return APK_OK;
}

int apk_sim_step(apk_sim *s) {
  if (!s || s->host_only) return APK_ERR_INVALID;
  s->err.clear();
  if (s->time < s->tlim && (s->tlim - s->time) < s->dt) s->dt = s->tlim - s->time;
  SIM_TRY(s, pre_step(s));
  for (int stage = 1; stage <= s->nstages; ++stage) SIM_TRY(s, do_stage(s, stage));
  s->time += s->dt;
  s->ncycle += 1;
  s->zone_cycles += (long long)s->mesh.mb[0] * s->mesh.mb[1] * s->mesh.mb[2] * (long long)s->mesh.nblocks_total;
  double est = kHuge;
  if (s->amr && s->amr_adaptive && s->amr_check_interval > 0 && s->ncycle % s->amr_check_interval == 0) {
    // Mesh::LoadBalancingAndAdaptiveMeshRefinement, then the new time step (the reference's order).  The tag
    // reduction and the time-step reduction are enqueued back to back and read in ONE host round trip; the
    // estimate is only kept if the mesh stays as it is (every cycle but a few), else it is redone on the new one.
    AmrTagRequest req;
    SIM_TRY(s, amr_tags_begin(s, &req));
    DtEstimate e;
    SIM_TRY(s, estimate_timestep_read(s, &e));
    bool changed = false;
    SIM_TRY(s, amr_regrid(s, &changed, &req));
    if (changed) {  // measure again on the new mesh; flags raised during the cycle stay raised
      e.dt_hyp_local = kHuge;
      SIM_TRY(s, estimate_timestep_read(s, &e));
    }
    SIM_TRY(s, estimate_timestep_commit(s, e, &est));
  } else {
    SIM_TRY(s, estimate_timestep(s, &est));
  }
  set_global_dt(s, est);
  return APK_OK;
}

int apk_sim_run(apk_sim *s, int nlim, int *ncycles) {
  if (!s) return APK_ERR_INVALID;
  int n = 0;
  while (s->time < s->tlim && (nlim < 0 || n < nlim)) {
    int rc = apk_sim_step(s);
    if (rc != APK_OK) return rc;
    ++n;
  }
  if (ncycles) *ncycles = n;
  return APK_OK;
}

double apk_sim_time(const apk_sim *s) { return s->time; }
double apk_sim_dt(const apk_sim *s) { return s->dt; }
double apk_sim_tlim(const apk_sim *s) { return s->tlim; }
double apk_sim_c_h(const apk_sim *s) { return s->pkg.c_h; }
int apk_sim_ncycle(const apk_sim *s) { return s->ncycle; }
long long apk_sim_fofc_count(const apk_sim *s) { return s->fofc_total; }

int apk_sim_get_info(const apk_sim *s, apk_sim_info *o) {
  if (!s || !o) return APK_ERR_INVALID;
  std::memset(o, 0, sizeof(*o));
  o->fluid = s->pkg.fluid;
  o->recon = s->pkg.recon;
  o->riemann = s->pkg.riemann;
  o->integrator = s->pkg.integrator;
  for (int d = 0; d < 3; ++d) {
    o->nx[d] = s->mesh.nx[d];
    o->mb[d] = s->mesh.mb[d];
    o->xmin[d] = s->xmin[d];
    o->xmax[d] = s->xmax[d];
    o->dx[d] = s->dx[d];
  }
  o->ng = s->mesh.ng;
  o->nhydro = s->pkg.nhydro;
  o->nscalars = s->pkg.nscalars;
  o->ndim = s->mesh.ndim;
  o->nblocks_total = s->mesh.nblocks_total;
  o->nblocks_local = (int)s->mesh.local_gids.size();
  o->first_gid = s->mesh.local_gids.empty() ? -1 : s->mesh.local_gids[0];
  o->rank = s->rank;
  o->nranks = s->nranks;
  o->npeers = (int)s->mesh.peers.size();
  o->fofc = s->pkg.first_order_flux_correct;
  o->dedner_extended = s->pkg.glmmhd_source_extended;
  o->fused = stage_can_fuse(s);
  o->cfl = s->pkg.cfl;
  o->gamma = s->pkg.eos.gamma;
  o->glmmhd_alpha = s->pkg.glmmhd_alpha;
  o->cells_per_block = s->mesh.sn;
  o->zones_local = (int64_t)s->mesh.mb[0] * s->mesh.mb[1] * s->mesh.mb[2] * (int64_t)s->mesh.local_gids.size();
  o->zones_total = s->amr ? o->zones_local : (int64_t)s->mesh.nx[0] * s->mesh.nx[1] * s->mesh.nx[2];
  return APK_OK;
}

int apk_sim_block_location(const apk_sim *s, int lb, int *gid, int loc[3]) {
  if (!s || lb < 0 || lb >= (int)s->mesh.local_gids.size()) return APK_ERR_INVALID;
  if (gid) *gid = s->mesh.local_gids[lb];
  if (loc && s->amr) {
    for (int d = 0; d < 3; ++d) loc[d] = amr_leaf(s, lb).lx[d];
  } else if (loc) {
    s->mesh.Loc(s->mesh.local_gids[lb], loc);
  }
  return APK_OK;
}

// refinement level of a local block (0 on uniform meshes); with apk_sim_block_location's logical
// location at that level this places the block: x_min = mesh x_min + loc * nx_block * dx / 2^level
int apk_sim_block_level(const apk_sim *s, int lb) {
  if (!s || lb < 0 || lb >= (int)s->mesh.local_gids.size()) return -1;
  return block_level(s, lb);
}

// stages with first_order_flux_correct that had to fall back from the optimistic fused stage to the
// flux-array sequence because a cell failed the admissibility test
long long apk_sim_fofc_fallback_stages(const apk_sim *s) { return s ? s->fofc_fallback_stages : 0; }

int apk_sim_amr_stats(const apk_sim *s, long long *refined, long long *derefined, int *max_level, long long *zone_cycles) {
  if (!s) return APK_ERR_INVALID;
  if (refined) *refined = s->amr_refined;
  if (derefined) *derefined = s->amr_derefined;
  if (max_level) *max_level = s->amr ? s->amr->max_level : 0;
  if (zone_cycles) *zone_cycles = s->zone_cycles;
  return APK_OK;
}

// A regridding pass for GIVEN tags (+1 refine / -1 derefine / 0 per block of the forest, global
// numbering; every rank passes the same array): forest update, new distribution, new plans and --
// on a device sim -- the transfer of the state, ghost exchange and ConsToPrim on the new mesh
int apk_sim_amr_apply_tags(apk_sim *s, const int *tags, int ntags, int *changed) {
  if (!s || !s->amr || !tags || ntags != (int)s->amr->leaves.size()) return APK_ERR_INVALID;
  const std::vector<AmrLeaf> old = s->amr->leaves;
  const AmrPartition old_part = s->amr_part;
  bool ch = false;
  try {
    ch = amr_update_tree(s, std::vector<int>(tags, tags + ntags), true);
    if (s->host_only) {
      amr_sync_mesh(s);
      amr_localize(s);
    }
  } catch (const std::exception &e) {
    return fail(s, APK_ERR_INVALID, e.what());
  }
  if (!s->host_only && ch) {  // move the state as a regridding pass of the run does
    SIM_TRY(s, amr_transfer(s, old, old_part));
    SIM_TRY(s, exchange_ghosts(s));
    SIM_TRY(s, fill_derived(s));
  }
  if (changed) *changed = ch ? 1 : 0;
  return APK_OK;
}

// one regridding pass on demand (adaptive meshes do this every check_refine_interval cycles)
int apk_sim_regrid(apk_sim *s, int *changed) {
  if (!s || s->host_only || !s->amr) return APK_ERR_INVALID;
  bool ch = false;
  SIM_TRY(s, amr_regrid(s, &ch));
  if (changed) *changed = ch ? 1 : 0;
  return APK_OK;
}

void *apk_sim_block_ptr(const apk_sim *s, int lb, int field) {
  if (!s || s->host_only || lb < 0 || lb >= (int)s->mesh.local_gids.size()) return nullptr;
  double *base = field == 0 ? s->d_cons() : (field == 1 ? s->d_prim() : (field == 2 ? s->d_cons2[s->u1buf] : nullptr));
  return base ? s->blk(base, lb) : nullptr;
}

// The accessors hand blocks over in the natural layout, [nvar][Nk][Nj][Ni], whatever the row pitch on the device
namespace {
void unpad_block(const apk_sim *s, const double *padded, double *natural) {
  const Mesh &m = s->mesh;
  for (int n = 0; n < m.nvar; ++n)
    for (int k = 0; k < m.nk; ++k)
      for (int j = 0; j < m.nj; ++j)
        std::memcpy(natural + (((int64_t)n * m.nk + k) * m.nj + j) * m.ni, padded + n * m.sn + k * m.sk + j * m.sj, sizeof(double) * m.ni);
}
void pad_block(const apk_sim *s, const double *natural, double *padded) {
  const Mesh &m = s->mesh;
  for (int n = 0; n < m.nvar; ++n)
    for (int k = 0; k < m.nk; ++k)
      for (int j = 0; j < m.nj; ++j)
        std::memcpy(padded + n * m.sn + k * m.sk + j * m.sj, natural + (((int64_t)n * m.nk + k) * m.nj + j) * m.ni, sizeof(double) * m.ni);
}
}  // namespace

// a block as it lies on the device (rows at the mesh's pitch; the block's nblk doubles): the host-side readers of this
// file index it through mesh.sj / sk / sn
static int read_block_device_layout(apk_sim *s, int lb, int field, double *host_out) {
  if (s && !s->host_only) SIM_TRY(s, sync_ghosts(s));
  void *p = apk_sim_block_ptr(s, lb, field);
  if (!p || !host_out) return APK_ERR_INVALID;
  SIM_HIP(s, hipStreamSynchronize(hs(s)));
  SIM_HIP(s, hipMemcpy(host_out, p, sizeof(double) * s->nblk, hipMemcpyDeviceToHost));
  return APK_OK;
}

int apk_sim_read_block(apk_sim *s, int lb, int field, double *host_out) {
  if (!s || s->host_only || s->mesh.pitch == 0) return read_block_device_layout(s, lb, field, host_out);
  std::vector<double> tmp((size_t)s->nblk);
  SIM_TRY(s, read_block_device_layout(s, lb, field, tmp.data()));
  unpad_block(s, tmp.data(), host_out);
  return APK_OK;
}

int apk_sim_write_block(apk_sim *s, int lb, int field, const double *host_in) {
  if (s && !s->host_only) SIM_TRY(s, sync_ghosts(s));
  void *p = apk_sim_block_ptr(s, lb, field);
  if (!p || !host_in) return APK_ERR_INVALID;
  SIM_HIP(s, hipStreamSynchronize(hs(s)));
  if (s->mesh.pitch > 0) {
    std::vector<double> tmp((size_t)s->nblk);
    SIM_HIP(s, hipMemcpy(tmp.data(), p, sizeof(double) * s->nblk, hipMemcpyDeviceToHost));  // (the padding keeps what it held)
    pad_block(s, host_in, tmp.data());
    SIM_HIP(s, hipMemcpy(p, tmp.data(), sizeof(double) * s->nblk, hipMemcpyHostToDevice));
    return APK_OK;
  }
  SIM_HIP(s, hipMemcpy(p, host_in, sizeof(double) * s->nblk, hipMemcpyHostToDevice));
  return APK_OK;
}

int apk_sim_gather(apk_sim *s, int field, double *out) {
  if (!s || s->host_only || !out) return APK_ERR_INVALID;
  if (s->amr) return fail(s, APK_ERR_UNSUPPORTED, "apk_sim_gather needs a uniform mesh: read refined meshes block by block");
  const Mesh &m = s->mesh;
  std::vector<double> host((size_t)s->nper);
  const int64_t NX = m.nx[0], NY = m.nx[1], NZ = m.nx[2];
  for (int lb = 0; lb < (int)m.local_gids.size(); ++lb) {
    int rc = read_block_device_layout(s, lb, field, host.data());
    if (rc != APK_OK) return rc;
    int bc[3];
    m.Loc(m.local_gids[lb], bc);
    for (int n = 0; n < m.nvar; ++n)
      for (int k = m.ks; k <= m.ke; ++k)
        for (int j = m.js; j <= m.je; ++j)
          for (int i = m.is; i <= m.ie; ++i) {
            const int64_t gi = (int64_t)bc[0] * m.mb[0] + (i - m.is), gj = (int64_t)bc[1] * m.mb[1] + (j - m.js),
                          gk = (int64_t)bc[2] * m.mb[2] + (k - m.ks);
            out[((n * NZ + gk) * NY + gj) * NX + gi] = host[n * m.sn + k * m.sk + j * m.sj + i];
          }
  }
  return APK_OK;
}

int apk_sim_history(apk_sim *s, double *out8) {
  if (!s || s->host_only || !out8) return APK_ERR_INVALID;
  SIM_TRY(s, apk_history(s->ctx, s->mu0(), s->pkg.fluid, out8, s->stream));
  if (s->have_comm && s->nranks > 1) {
    if (s->comm.allreduce_sum(s->comm.user, out8, 8) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_sum failed");
  }
  return APK_OK;
}

// TurbulenceHst<Ms|Ma|pb> (src/pgen/turbulence.cpp:47-101), summed over ranks like Parthenon's
// UserHistoryOperation::sum
int apk_sim_turbulence_history(apk_sim *s, double *out3) {
  if (!s || s->host_only || !out3) return APK_ERR_INVALID;
  if (s->prim_stale) SIM_TRY(s, sync_ghosts(s));  // (the Mach numbers are sums over the PRIMITIVES of the interior)
  SIM_TRY(s, apk_turbulence_history(s->ctx, s->mu0(), s->pkg.fluid, s->pkg.eos.gamma, out3, s->stream));
  if (s->have_comm && s->nranks > 1) {
    if (s->comm.allreduce_sum(s->comm.user, out3, 3) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_sum failed");
  }
  return APK_OK;
}

// field_loop::RelDivBHst (src/pgen/field_loop.cpp:60-95), registered as "UserRelDivB" (:97-103)
int apk_sim_user_reldivb(apk_sim *s, double *out) {
  if (!s || s->host_only || !out || s->problem_id != "field_loop") return APK_ERR_INVALID;
  SIM_TRY(s, sync_ghosts(s));  // (div B differences reach into the ghost zones)
  SIM_TRY(s, apk_history_user_reldivb(s->ctx, s->mu0(), s->floop.amp, out, s->stream));
  if (s->have_comm && s->nranks > 1) {
    if (s->comm.allreduce_sum(s->comm.user, out, 1) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_sum failed");
  }
  return APK_OK;
}

int apk_sim_fmft_num_modes(const apk_sim *s) { return (s && s->fmft) ? s->fmft->num_modes() : 0; }

int apk_sim_fmft_var_hat(const apk_sim *s, double *out) {
  if (!s || !s->fmft || !out) return APK_ERR_INVALID;
  const auto &vh = s->fmft->var_hat();
  for (size_t q = 0; q < vh.size(); ++q) {
    out[2 * q] = vh[q].real();
    out[2 * q + 1] = vh[q].imag();
  }
  return APK_OK;
}

int apk_sim_fmft_evolve(apk_sim *s, double dt) {
  if (!s || !s->fmft) return APK_ERR_INVALID;
  s->fmft->Evolve(dt);
  return APK_OK;
}

int apk_sim_fmft_phases(const apk_sim *s, int axis, int n, int g0, double *out) {
  if (!s || !s->fmft || !out || axis < 0 || axis > 2 || n <= 0) return APK_ERR_INVALID;
  s->fmft->Phases(axis, n, g0, s->mesh.nx[axis], out);
  return APK_OK;
}

int apk_sim_read_acc(apk_sim *s, int lb, double *host_out) {
  if (!s || s->host_only || !s->d_acc || !host_out || lb < 0 || lb >= (int)s->mesh.local_gids.size()) return APK_ERR_INVALID;
  SIM_HIP(s, hipStreamSynchronize(hs(s)));
  SIM_HIP(s, hipMemcpy(host_out, s->d_acc + 3 * (int64_t)s->mesh.sn * lb, sizeof(double) * 3 * s->mesh.sn, hipMemcpyDeviceToHost));
  return APK_OK;
}

// pkg->CheckRefinementBlock as configured by <refinement> (src/hydro/hydro.cpp:788-816), evaluated
// for every local block: tags[lb] = +1 refine / 0 same / -1 derefine.  The mesh itself stays uniform
// here; this is the tagging half of the AMR loop.
int apk_sim_check_refinement(apk_sim *s, int *tags, double *crit) {
  if (!s || s->host_only || !tags) return APK_ERR_INVALID;
  SIM_TRY(s, sync_ghosts(s));
  int criterion;
  double p0, p1;
  SIM_TRY(s, refinement_criterion(s, &criterion, &p0, &p1));
  SIM_TRY(s, apk_tag_blocks(s->ctx, s->mu0(), criterion, p0, p1, tags, crit, s->stream));
  return APK_OK;
}

int apk_sim_history_labels(const apk_sim *s, char *buf, size_t len) {
  if (!s || !buf || !len) return APK_ERR_INVALID;
  std::string l = "mass 1-mom 2-mom 3-mom KE tot-E";
  const bool mhd = s->pkg.fluid == APK_FLUID_GLMMHD;
  if (mhd) l += " ME relDivB";
  if (s->fmft) l += mhd ? " Ms Ma plasma_beta" : " Ms";
  if (s->problem_id == "field_loop") l += " UserRelDivB";
  std::snprintf(buf, len, "%s", l.c_str());
  return APK_OK;
}

int apk_sim_write_history(apk_sim *s, const char *path) {
  if (!s || s->host_only || !path) return APK_ERR_INVALID;
  const bool mhd = s->pkg.fluid == APK_FLUID_GLMMHD;
  double h[8], t3[3] = {0, 0, 0};
  int rc = apk_sim_history(s, h);
  if (rc != APK_OK) return rc;
  if (s->fmft && (rc = apk_sim_turbulence_history(s, t3)) != APK_OK) return rc;
  const bool floop = s->problem_id == "field_loop";
  double urdb = 0.0;
  if (floop && (rc = apk_sim_user_reldivb(s, &urdb)) != APK_OK) return rc;
  if (s->rank != 0) return APK_OK;
  std::vector<double> row(h, h + (mhd ? 8 : 6));
  if (s->fmft) row.insert(row.end(), t3, t3 + (mhd ? 3 : 1));
  if (floop) row.push_back(urdb);
  FILE *f = std::fopen(path, "r");
  const bool fresh = (f == nullptr);
  if (f) std::fclose(f);
  f = std::fopen(path, "a");
  if (!f) return fail(s, APK_ERR_INVALID, std::string("history file could not be opened: ") + path);
  if (fresh) {
    char labels[256];
    apk_sim_history_labels(s, labels, sizeof(labels));
    int col = 1;
    std::fprintf(f, "#  History data\n");
    std::fprintf(f, "# [%d]=time     ", col++);
    std::fprintf(f, "[%d]=dt       ", col++);
    std::fprintf(f, "[%d]=cycle    ", col++);
    std::fprintf(f, "[%d]=nbtotal  ", col++);
    std::istringstream iss(labels);
    std::string lab;
    while (iss >> lab) std::fprintf(f, "[%d]=%-8s", col++, lab.c_str());
    std::fprintf(f, "\n");
  }
  const std::string fmt = " " + s->pin.GetOrAddString("parthenon/output_defaults", "data_format", "%12.5e");
  std::fprintf(f, fmt.c_str(), s->time);
  std::fprintf(f, fmt.c_str(), s->dt);
  std::fprintf(f, " %d %d", s->ncycle, s->mesh.nblocks_total);
  for (double v : row) std::fprintf(f, fmt.c_str(), v);
  std::fprintf(f, "\n");
  std::fclose(f);
  return APK_OK;
}

int apk_sim_write_linear_wave_errors(apk_sim *s, const char *path) {
  if (!s || !path) return APK_ERR_INVALID;
  const bool mhd_wave = s->problem_id == "linear_wave_mhd";
  const int ncol = mhd_wave ? 8 : 5;
  double rms = 0.0, l1[8], mx[8];
  int rc = mhd_wave ? apk_sim_linear_wave_mhd_errors(s, &rms, l1, mx) : apk_sim_linear_wave_errors(s, &rms, l1, mx);
  if (rc != APK_OK) return rc;
  if (s->rank != 0) return APK_OK;
  double max_max_over_l1 = 0.0;
  for (int n = 0; n < ncol; ++n) max_max_over_l1 = std::fmax(max_max_over_l1, mx[n] / l1[n]);
  FILE *f = std::fopen(path, "r");
  const bool fresh = (f == nullptr);
  if (f) std::fclose(f);
  f = std::fopen(path, "a");
  if (!f) return fail(s, APK_ERR_INVALID, "Error output file could not be opened");
  if (fresh) {  // linear_wave.cpp:315-320 / linear_wave_mhd.cpp:318-326
    std::fprintf(f, "# Nx1  Nx2  Nx3  Ncycle  ");
    std::fprintf(f, "RMS-L1-Error  d_L1  M1_L1  M2_L1  M3_L1  E_L1 ");
    if (mhd_wave) std::fprintf(f, "  B1c_L1  B2c_L1  B3c_L1");
    std::fprintf(f, "  Largest-Max/L1  d_max  M1_max  M2_max  M3_max  E_max ");
    if (mhd_wave) std::fprintf(f, "  B1c_max  B2c_max  B3c_max");
    std::fprintf(f, "\n");
  }
  // the hydro file's column 3 repeats Nx2 (that is what linear_wave.cpp:323-324 prints); the MHD file's holds Nx3
  std::fprintf(f, "%d  %d", s->mesh.nx[0], s->mesh.nx[1]);
  std::fprintf(f, "  %d  %d", mhd_wave ? s->mesh.nx[2] : s->mesh.nx[1], s->ncycle);
  std::fprintf(f, "  %e  %e", rms, l1[0]);
  std::fprintf(f, "  %e  %e  %e", l1[1], l1[2], l1[3]);
  std::fprintf(f, "  %e", l1[4]);
  for (int n = 5; n < ncol; ++n) std::fprintf(f, "  %e", l1[n]);
  std::fprintf(f, "  %e  %e  ", max_max_over_l1, mx[0]);
  std::fprintf(f, "%e  %e  %e", mx[1], mx[2], mx[3]);
  std::fprintf(f, "  %e", mx[4]);
  for (int n = 5; n < ncol; ++n) std::fprintf(f, "  %e", mx[n]);
  std::fprintf(f, "\n");
  std::fclose(f);
  return APK_OK;
}

int apk_sim_execute(apk_sim *s, const char *outdir, int *ncycles) {
  if (!s || s->host_only || !outdir) return APK_ERR_INVALID;
  struct HstOut {
    std::string path;
    double dt, next;
  };
  std::vector<HstOut> outs;
  try {
    const std::string base = s->pin.GetOrAddString("parthenon/job", "problem_id", "parthenon");
    for (const std::string &blk : s->pin.BlocksWithPrefix("parthenon/output")) {
      if (blk == "parthenon/output_defaults" || !s->pin.DoesParameterExist(blk, "file_type")) continue;
      if (s->pin.GetString(blk, "file_type") != "hst") continue;  // hdf5 / rst outputs are out of scope
      const std::string num = blk.substr(std::string("parthenon/output").size());
      const double out_dt = s->pin.GetReal(blk, "dt");
      if (!(out_dt > 0.0)) throw std::runtime_error("<" + blk + ">: dt must be positive for hst outputs");
      outs.push_back({std::string(outdir) + "/" + base + ".out" + num + ".hst", out_dt, 0.0});
      if (s->pin.DoesParameterExist(blk, "data_format"))
        s->pin.ApplyOverride("parthenon/output_defaults/data_format=" + s->pin.GetString(blk, "data_format"));
    }
  } catch (const std::exception &e) {
    return fail(s, APK_ERR_INVALID, e.what());
  }
  SIM_TRY(s, apk_sim_initialize(s));
  for (auto &o : outs) {
    if (s->rank == 0) std::remove(o.path.c_str());
    SIM_TRY(s, apk_sim_write_history(s, o.path.c_str()));
    o.next = o.dt;
  }
  int n = 0;
  // parthenon/time/perf_cycle_offset: cycles left out of the performance figure (first-touch
  // allocations, e.g. the flux-difference workspace and the spare prim buffer, happen there)
  const int perf_offset = s->pin.GetOrAddInteger("parthenon/time", "perf_cycle_offset", 0);
  SIM_HIP(s, hipStreamSynchronize(hs(s)));
  auto t0 = std::chrono::steady_clock::now();
  s->perf_cycles = 0;
  s->perf_zone_mark = s->zone_cycles;
  while (s->time < s->tlim && (s->nlim < 0 || n < s->nlim)) {
    if (n == perf_offset && n > 0) {
      SIM_HIP(s, hipStreamSynchronize(hs(s)));
      t0 = std::chrono::steady_clock::now();
      s->perf_zone_mark = s->zone_cycles;
    }
    SIM_TRY(s, apk_sim_step(s));
    ++n;
    if (n > perf_offset) s->perf_cycles += 1;
    const bool last = !(s->time < s->tlim && (s->nlim < 0 || n < s->nlim));
    for (auto &o : outs)
      if (s->time >= o.next || last) {
        SIM_TRY(s, apk_sim_write_history(s, o.path.c_str()));
        while (o.next <= s->time) o.next += o.dt;
      }
  }
  SIM_TRY(s, sync_ghosts(s));
  SIM_HIP(s, hipStreamSynchronize(hs(s)));
  s->loop_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if ((s->problem_id == "linear_wave" || s->problem_id == "linear_wave_mhd") && s->lw.compute_error)
    SIM_TRY(s, apk_sim_write_linear_wave_errors(s, (std::string(outdir) + "/linearwave-errors.dat").c_str()));
  if (s->problem_id == "cpaw" && s->cpaw.compute_error)
    SIM_TRY(s, apk_sim_write_cpaw_errors(s, (std::string(outdir) + "/cpaw-errors.dat").c_str()));
  if (ncycles) *ncycles = n;
  return APK_OK;
}

// src/pgen/linear_wave.cpp:183-335 (5 columns: d, M1, M2, M3, E) and src/pgen/linear_wave_mhd.cpp:177-276 (8 columns:
// + B1, B2, B3 against the ANALYTIC field; psi is not part of the norm): volume-weighted L1 and max errors of the
// interior cells against the wave at the initial phase, L1 normalised by the domain volume, RMS over the columns
static int linear_wave_errors_n(apk_sim *s, int ncol, double *rms, double *l1, double *mx) {
  const Mesh &m = s->mesh;
  const bool mhd_wave = ncol == 8;
  std::vector<double> host((size_t)s->nper);
  double acc[16] = {0};
  for (int lb = 0; lb < (int)m.local_gids.size(); ++lb) {
    int rc = read_block_device_layout(s, lb, 0, host.data());
    if (rc != APK_OK) return rc;
    LevelDxScope level_dx_scope(s, lb);
    const double cellvol = s->dx[0] * s->dx[1] * s->dx[2];
    double x0[3];
    block_origin(s, lb, x0);
    for (int k = m.ks; k <= m.ke; ++k)
      for (int j = m.js; j <= m.je; ++j)
        for (int i = m.is; i <= m.ie; ++i) {
          double u[8];
          if (mhd_wave) lwm_state(s, xc(s, x0, 0, i), xc(s, x0, 1, j), xc(s, x0, 2, k), u);
          else lw_state(s->lw, xc(s, x0, 0, i), xc(s, x0, 1, j), xc(s, x0, 2, k), u);
          for (int n = 0; n < ncol; ++n) {
            const double e = std::abs(u[n] - host[n * m.sn + k * m.sk + j * m.sj + i]);
            acc[n] += e * cellvol;
            if (e > acc[8 + n]) acc[8 + n] = e;
          }
        }
  }
  if (s->have_comm && s->nranks > 1) {
    if (s->comm.allreduce_sum(s->comm.user, acc, ncol) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_sum failed");
    // MAX of non-negative values through the MIN callback
    double neg[8];
    for (int n = 0; n < ncol; ++n) neg[n] = -acc[8 + n];
    if (s->comm.allreduce_min(s->comm.user, neg, ncol) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_min failed");
    for (int n = 0; n < ncol; ++n) acc[8 + n] = -neg[n];
  }
  const double vol = (s->xmax[0] - s->xmin[0]) * (s->xmax[1] - s->xmin[1]) * (s->xmax[2] - s->xmin[2]);
  double r = 0.0;
  for (int n = 0; n < ncol; ++n) {
    l1[n] = acc[n] / vol;
    mx[n] = acc[8 + n];
    r += l1[n] * l1[n];
  }
  *rms = std::sqrt(r);
  return APK_OK;
}

int apk_sim_linear_wave_errors(apk_sim *s, double *rms, double *l1, double *mx) {
  if (!s || s->host_only || s->problem_id != "linear_wave" || !rms || !l1 || !mx) return APK_ERR_INVALID;
  return linear_wave_errors_n(s, 5, rms, l1, mx);
}

int apk_sim_linear_wave_mhd_errors(apk_sim *s, double *rms, double *l1, double *mx) {
  if (!s || s->host_only || s->problem_id != "linear_wave_mhd" || !rms || !l1 || !mx) return APK_ERR_INVALID;
  return linear_wave_errors_n(s, 8, rms, l1, mx);
}

// cpaw::UserWorkAfterLoop (src/pgen/cpaw.cpp:127-221): L1 errors against the initial state, err8 in
// the order d, M1, M2, M3, E, B1, B2, B3
int apk_sim_cpaw_errors(apk_sim *s, double *rms, double *err8) {
  if (!s || s->host_only || s->problem_id != "cpaw" || !rms || !err8) return APK_ERR_INVALID;
  const Mesh &m = s->mesh;
  const CpawState &c = s->cpaw;
  std::vector<double> host((size_t)s->nper);
  double err[8] = {0};
  for (int lb = 0; lb < (int)m.local_gids.size(); ++lb) {
    int rc = read_block_device_layout(s, lb, 0, host.data());
    if (rc != APK_OK) return rc;
    double x0[3];
    block_origin(s, lb, x0);
    auto at = [&](int n, int k, int j, int i) { return host[n * m.sn + k * m.sk + j * m.sj + i]; };
    for (int k = m.ks; k <= m.ke; ++k)
      for (int j = m.js; j <= m.je; ++j)
        for (int i = m.is; i <= m.ie; ++i) {
          double mom[3], b[3];
          cpaw_state(c, xc(s, x0, 0, i), xc(s, x0, 1, j), xc(s, x0, 2, k), mom, b);
          err[0] += std::abs(c.den - at(0, k, j, i));
          for (int d = 0; d < 3; ++d) err[1 + d] += std::abs(mom[d] - at(1 + d, k, j, i));
          for (int d = 0; d < 3; ++d) err[5 + d] += std::abs(b[d] - at(5 + d, k, j, i));
          const double e0 = c.pres / c.gm1 + 0.5 * (mom[0] * mom[0] + mom[1] * mom[1] + mom[2] * mom[2]) / c.den +
                            0.5 * (b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
          err[4] += std::abs(e0 - at(4, k, j, i));
        }
  }
  if (s->have_comm && s->nranks > 1) {
    if (s->comm.allreduce_sum(s->comm.user, err, 8) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_sum failed");
  }
  const double ncells = (double)m.nx[0] * m.nx[1] * m.nx[2];
  double r = 0.0;
  for (int n = 0; n < 8; ++n) {
    err8[n] = err[n] / ncells;
    r += err8[n] * err8[n];
  }
  *rms = std::sqrt(r);
  return APK_OK;
}

// "cpaw-errors.dat" (cpaw.cpp:188-220): header when the file is new, otherwise append
int apk_sim_write_cpaw_errors(apk_sim *s, const char *path) {
  if (!s || !path) return APK_ERR_INVALID;
  double rms = 0.0, err[8];
  int rc = apk_sim_cpaw_errors(s, &rms, err);
  if (rc != APK_OK) return rc;
  if (s->rank != 0) return APK_OK;
  FILE *f = std::fopen(path, "r");
  const bool fresh = (f == nullptr);
  if (f) std::fclose(f);
  f = std::fopen(path, "a");
  if (!f) return fail(s, APK_ERR_INVALID, "Error output file could not be opened");
  if (fresh) {
    std::fprintf(f, "# Nx1  Nx2  Nx3  Ncycle  RMS-Error  d  M1  M2  M3");
    std::fprintf(f, "  E");
    std::fprintf(f, "  B1c  B2c  B3c");
    std::fprintf(f, "\n");
  }
  std::fprintf(f, "%d  %d", s->mesh.nx[0], s->mesh.nx[1]);
  std::fprintf(f, "  %d  %d  %e", s->mesh.nx[2], s->ncycle, rms);
  std::fprintf(f, "  %e  %e  %e  %e", err[0], err[1], err[2], err[3]);
  std::fprintf(f, "  %e", err[4]);
  std::fprintf(f, "  %e  %e  %e", err[5], err[6], err[7]);
  std::fprintf(f, "\n");
  std::fclose(f);
  return APK_OK;
}

int apk_sim_exchange_ghosts(apk_sim *s) {
  if (!s || s->host_only) return APK_ERR_INVALID;
  SIM_TRY(s, sync_ghosts(s));
  return exchange_ghosts(s);
}
int apk_sim_fill_derived(apk_sim *s) {
  if (!s || s->host_only) return APK_ERR_INVALID;
  SIM_TRY(s, sync_ghosts(s));
  return fill_derived(s);
}
int apk_sim_estimate_timestep(apk_sim *s, double *dt) {
  if (!s || s->host_only || !dt) return APK_ERR_INVALID;
  return estimate_timestep(s, dt);
}

// after the caller replaced the state (apk_sim_write_block + apk_sim_exchange_ghosts + apk_sim_fill_derived): the time
// step as apk_sim_initialize derives it from a problem generator's state -- no growth limit from an earlier step
int apk_sim_reset_time_step(apk_sim *s) {
  if (!s || s->host_only) return APK_ERR_INVALID;
  s->err.clear();
  SIM_TRY(s, sync_ghosts(s));
  s->dt = kHuge;
  s->pkg.mindx = kHuge;
  s->pkg.dt_hyp = kHuge;
  s->dt_hyp_is_global = false;
  s->stage_dt_pending = false;
  double est = kHuge;
  SIM_TRY(s, estimate_timestep(s, &est));
  set_global_dt(s, est);
  return APK_OK;
}

int apk_sim_kernel_timing_enable(apk_sim *s, int on) {
  if (!s || s->host_only) return APK_ERR_INVALID;
  return apk_kernel_timing_enable(s->ctx, on);
}
int apk_sim_kernel_timing_read(apk_sim *s, int slot, double *total_ms, long long *launches) {
  if (!s || s->host_only) return APK_ERR_INVALID;
  return apk_kernel_timing_read(s->ctx, slot, total_ms, launches);
}

// the message set the next comm.exchange moves: the uniform mesh's halo buffers, or -- on refined
// meshes -- whichever of halo / flux-correction / regridding messages the driver has made current
int apk_sim_num_peers(const apk_sim *s) {
  if (!s) return APK_ERR_INVALID;
  return s->active_msgs ? (int)s->active_msgs->plan.peers.size() : (int)s->mesh.peers.size();
}
long long apk_sim_message_generation(const apk_sim *s) { return s ? s->msg_generation : 0; }
// plan introspection: make the halo (1) or flux-correction (2) message set of a refined mesh the one
// apk_sim_peer reports (0: back to the uniform mesh's)
int apk_sim_select_messages(apk_sim *s, int which) {
  if (!s || which < 0 || which > 5 || (which > 0 && which < 5 && !s->amr) || (which == 5 && s->amr)) return APK_ERR_INVALID;
  const apk_sim::MsgSet *sets[6] = {nullptr, &s->amr_halo, &s->amr_fluxmsg, &s->amr_halo_faces, &s->amr_halo_shell, nullptr};
  s->active_msgs = sets[which];
  s->thin_msgs = which == 5;  // (the one-layer set of a uniform mesh; every exchange selects the set it moves anyway)
  s->msg_generation += 1;
  return APK_OK;
}

int apk_sim_peer(const apk_sim *s, int p, apk_peer_info *o) {
  if (!s || !o || p < 0 || p >= apk_sim_num_peers(s)) return APK_ERR_INVALID;
  if (s->active_msgs) {
    const apk_sim::MsgSet &m = *s->active_msgs;
    o->rank = m.plan.peers[p].rank;
    o->send_count = m.plan.peers[p].send_count;
    o->recv_count = m.plan.peers[p].recv_count;
    o->send_buf = (p < (int)m.send.size()) ? m.send[p] : nullptr;
    o->recv_buf = (p < (int)m.recv.size()) ? m.recv[p] : nullptr;
    return APK_OK;
  }
  o->rank = s->mesh.peers[p].rank;
  o->send_count = s->thin_msgs ? s->mesh.peers[p].send_count_thin : s->mesh.peers[p].send_count;
  o->recv_count = s->thin_msgs ? s->mesh.peers[p].recv_count_thin : s->mesh.peers[p].recv_count;
  o->send_buf = (p < (int)s->send_buf.size()) ? s->send_buf[p] : nullptr;
  o->recv_buf = (p < (int)s->recv_buf.size()) ? s->recv_buf[p] : nullptr;
  return APK_OK;
}

// phases 0..5: the uniform-mesh plan; 10: multilevel fill copies, 11..13: coarse-buffer boundaries
// x1..x3, 14..16: block boundaries x1..x3, 17..19: flux-correction copies x1..x3
const std::vector<BoxRegion> *plan_of_phase(const apk_sim *s, int phase) {
  // (0..7: the uniform mesh's plans PH_LOCAL .. PH_UNPACK_THIN; 60..63: the same pack / unpack plans without the x1 faces,
  // PH_PACK_NOX1 .. PH_UNPACK_THIN_NOX1 -- the numbers 10..49 below belong to refined meshes)
  if (phase >= 0 && phase < PH_PACK_NOX1) return &s->mesh.plan[phase];
  if (phase >= 60 && phase < 60 + (PH_COUNT - PH_PACK_NOX1)) return &s->mesh.plan[PH_PACK_NOX1 + (phase - 60)];
  if (!s->amr) return nullptr;
  // this rank's share (local block numbers; kinds 1 / 2 = message buffers of the halo set for 20..24,
  // of the flux-correction set for 25..33)
  const auto &l = s->amr_local;
  if (phase == 20) return &l.fill;
  if (phase == 21) return &l.fill_pack;
  if (phase == 22) return &l.fill_unpack;
  if (phase >= 25 && phase <= 27) return &l.flux_copy[phase - 25];
  if (phase >= 28 && phase <= 30) return &l.flux_pack[phase - 28];
  if (phase >= 31 && phase <= 33) return &l.flux_unpack[phase - 31];
  if (phase >= 34 && phase <= 36) return &l.coarse_bc[phase - 34];
  if (phase >= 37 && phase <= 39) return &l.fine_bc[phase - 37];
  // the exchanges of the stage loop: 40..42 faces only (fill copies, packs, unpacks), 43 the fill copies of 40 without
  // the same-level same-rank ones (direct neighbour addressing), 44..46 the shell (fill copies, packs, unpacks),
  // 47..49 its block boundaries
  if (phase == 40) return &l.fill_faces;
  if (phase == 41) return &l.fill_pack_faces;
  if (phase == 42) return &l.fill_unpack_faces;
  if (phase == 43) return &l.fill_direct;
  if (phase == 44) return &l.fill_shell;
  if (phase == 45) return &l.fill_pack_shell;
  if (phase == 46) return &l.fill_unpack_shell;
  if (phase >= 47 && phase <= 49) return &l.fine_bc_shell[phase - 47];
  if (phase == 10) return &s->amr_plans.fill;
  if (phase >= 11 && phase <= 13) return &s->amr_plans.coarse_bc[phase - 11];
  if (phase >= 14 && phase <= 16) return &s->amr_plans.fine_bc[phase - 14];
  if (phase >= 17 && phase <= 19) return &s->amr_plans.flux_copy[phase - 17];
  return nullptr;
}

int apk_sim_plan_size(const apk_sim *s, int phase) {
  const std::vector<BoxRegion> *p = s ? plan_of_phase(s, phase) : nullptr;
  return p ? (int)p->size() : APK_ERR_INVALID;
}

int apk_sim_plan_region(const apk_sim *s, int phase, int r, apk_region_info *o) {
  const std::vector<BoxRegion> *p = (s && o) ? plan_of_phase(s, phase) : nullptr;
  if (!p || r < 0 || r >= (int)p->size()) return APK_ERR_INVALID;
  const BoxRegion &b = (*p)[r];
  o->src_kind = b.src_kind;
  o->src_block = b.src_block;
  o->dst_kind = b.dst_kind;
  o->dst_block = b.dst_block;
  o->src_off = b.src_off;
  o->dst_off = b.dst_off;
  o->nvar = b.nvar;
  o->flip_var = b.flip_var;
  for (int q = 0; q < 3; ++q) o->ext[q] = b.ext[q];
  for (int q = 0; q < 4; ++q) {
    o->src_stride[q] = b.src_stride[q];
    o->dst_stride[q] = b.dst_stride[q];
  }
  return APK_OK;
}

// operator lists of the multilevel plans: 0 restrict-own, 1 prolongate, 2..4 flux restriction x1..x3
const std::vector<AmrRefOp> *ops_of(const apk_sim *s, int which) {
  if (!s->amr) return nullptr;
  // 10..14: this rank's share of 0..4 (local block numbers)
  if (which == 10) return &s->amr_local.restrict_own;
  if (which == 11) return &s->amr_local.prolongate;
  if (which >= 12 && which <= 14) return &s->amr_local.flux_restrict[which - 12];
  if (which == 15) return &s->amr_local.prolongate_faces;  // (of the faces-only exchange / of the shell exchange)
  if (which == 16) return &s->amr_local.prolongate_shell;
  if (which == 0) return &s->amr_plans.restrict_own;
  if (which == 1) return &s->amr_plans.prolongate;
  if (which >= 2 && which <= 4) return &s->amr_plans.flux_restrict[which - 2];
  return nullptr;
}

int apk_sim_amr_ops_size(const apk_sim *s, int which) {
  const std::vector<AmrRefOp> *p = s ? ops_of(s, which) : nullptr;
  return p ? (int)p->size() : APK_ERR_INVALID;
}

int apk_sim_amr_op(const apk_sim *s, int which, int n, apk_amr_op_info *o) {
  const std::vector<AmrRefOp> *p = (s && o) ? ops_of(s, which) : nullptr;
  if (!p || n < 0 || n >= (int)p->size()) return APK_ERR_INVALID;
  const AmrRefOp &a = (*p)[n];
  o->kind = a.kind;
  o->level = a.level;
  o->src_kind = a.src_kind;
  o->src_block = a.src_block;
  o->dst_kind = a.dst_kind;
  o->dst_block = a.dst_block;
  for (int q = 0; q < 3; ++q) {
    o->lo[q] = a.lo[q];
    o->hi[q] = a.hi[q];
    o->dx[q] = level_dx(s, a.level, q);
    o->xmin[q] = s->xmin[q] + (double)s->amr->leaves[a.geom_block].lx[q] * s->mesh.mb[q] * o->dx[q];
  }
  o->cng = s->amr_geom.cng;
  o->coarse_doubles = s->amr_geom.coarse_doubles;
  return APK_OK;
}

}  // extern "C"
