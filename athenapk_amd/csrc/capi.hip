// capi.hip -- the extern "C" boundary (include/apk_amd.h).  Validates arguments on the host
// (the reference aborts with PARTHENON_FAIL on unknown options, src/hydro/hydro.cpp:299,338,
// 366; here they become error codes), then enqueues kernels on the caller's stream.
#include <cmath>
#include <cstring>
#include <limits>
#include <new>

#include "apk_internal.hpp"
#include "fused_kernel.hpp"

using namespace apk;

namespace {

hipStream_t as_stream(apk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

bool valid_eos(const apk_eos *e) { return e && e->gamma > 1.0; }

// registry of compiled-in flux functions: src/hydro/hydro.cpp:386-416
bool in_registry(const apk_flux_cfg &c) {
  const bool recon_ok = c.recon >= APK_RC_DC && c.recon <= APK_RC_LIMO3;
  if (!recon_ok) return false;
  if (c.riemann == APK_RS_NONE || c.riemann == APK_RS_LLF) return c.recon == APK_RC_DC;
  if (c.fluid == APK_FLUID_EULER) return c.riemann == APK_RS_HLLE || c.riemann == APK_RS_HLLC;
  if (c.fluid == APK_FLUID_GLMMHD) return c.riemann == APK_RS_HLLE || c.riemann == APK_RS_HLLD;
  return false;
}

int need_nghost(int recon) {  // src/hydro/hydro.cpp:316-339
  switch (recon) {
  case APK_RC_DC: return 1;
  case APK_RC_PPM:
  case APK_RC_WENOZ: return 3;
  default: return 2;
  }
}

int check_cfg(apk_ctx *ctx, const apk_pack *md, const apk_flux_cfg &cfg) {
  if (!in_registry(cfg)) return set_err(ctx, APK_ERR_UNSUPPORTED, "flux function not in registry");
  const int nh = (cfg.fluid == APK_FLUID_EULER) ? 5 : 9;
  if (md->view.nhydro != nh) return set_err(ctx, APK_ERR_INVALID, "pack nhydro does not match fluid");
  if (md->view.ng < need_nghost(cfg.recon))
    return set_err(ctx, APK_ERR_NGHOST, "need more ghost zones for chosen reconstruction");
  return APK_OK;
}

bool same_shape(const apk_pack *a, const apk_pack *b) {
  const PackView &x = a->view, &y = b->view;
  return x.nblocks == y.nblocks && x.nvar == y.nvar && x.ni == y.ni && x.nj == y.nj &&
         x.nk == y.nk && x.ng == y.ng && x.sj == y.sj && x.sk == y.sk && x.sn == y.sn;
}

int ensure_partial(apk_ctx *ctx, size_t n) {
  if (ctx->partial_cap >= n) return APK_OK;
  if (ctx->d_partial) (void)hipFree(ctx->d_partial);
  ctx->d_partial = nullptr;
  ctx->partial_cap = 0;
  APK_HIP_TRY(ctx, hipMalloc(&ctx->d_partial, n * sizeof(double)));
  ctx->partial_cap = n;
  return APK_OK;
}

int ensure_mark(apk_ctx *ctx, size_t n) {
  if (ctx->mark_cap >= n) return APK_OK;
  if (ctx->d_mark) (void)hipFree(ctx->d_mark);
  ctx->d_mark = nullptr;
  ctx->mark_cap = 0;
  APK_HIP_TRY(ctx, hipMalloc(&ctx->d_mark, n));
  ctx->mark_cap = n;
  return APK_OK;
}

}  // namespace

namespace apk {
namespace {
// word 4 = the stage's time-step minimum, word 5 = the two flag words; h = pinned host memory (device address)
__global__ void cycle_gather_kernel(unsigned long long *u64, unsigned long long *h, const unsigned long long *tag_words, double *h_tags,
                                    int ntags) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < ntags) {
    h_tags[t] = __longlong_as_double((long long)tag_words[t]);
    const_cast<unsigned long long *>(tag_words)[t] = 0ull;
  }
  if (t == 0) {
    h[4] = u64[4];
    h[5] = u64[5];
    u64[4] = u64[15];                                  // +max: ready for the next reduction
    reinterpret_cast<unsigned *>(u64 + 5)[0] = 0u;     // latched flags handed over ([1], the trial stage's, stays)
  }
}
}  // namespace

int prepare_dt_word(apk_ctx *ctx, hipStream_t s) {
  if (!ctx->dt_word_clean || ctx->clean_stream != s) {
    // +max (neutral element of the min), device to device so the launch path never blocks the host
    if (hipMemcpyAsync(ctx->d_u64 + 4, ctx->d_u64 + 15, sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return APK_ERR_DEVICE;
  }
  ctx->dt_word_clean = false;  // (about to be reduced into)
  ctx->last_stage_min_valid = false;
  return APK_OK;
}

int launch_cycle_gather(apk_ctx *ctx, hipStream_t s) {
  const int ntags = (ctx->tags_pending > 0 && ctx->h_partial_dev) ? ctx->tags_pending : 0;
  const int blocks = ntags > 0 ? (ntags + 255) / 256 : 1;
  hipLaunchKernelGGL(cycle_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ctx->d_u64, ctx->h_pinned_dev,
                     ctx->d_tagmax, ctx->h_partial_dev, ntags);
  if (hipGetLastError() != hipSuccess) return APK_ERR_DEVICE;
  ctx->dt_word_clean = true;
  ctx->clean_stream = s;
  if (ntags > 0) {
    ctx->tag_words_clean = ntags;
    ctx->tags_pending = 0;
  }
  return APK_OK;
}
}  // namespace apk

extern "C" {

int apk_version(void) { return APK_AMD_VERSION; }

int apk_fp_strict(void) {
#ifdef APK_FP_STRICT
  return 1;
#else
  return 0;
#endif
}

int apk_create(apk_ctx **out) {
  if (!out) return APK_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return APK_ERR_NO_DEVICE;
  apk_ctx *ctx = new (std::nothrow) apk_ctx();
  if (!ctx) return APK_ERR_INVALID;
  if (hipGetDevice(&ctx->device) != hipSuccess) {
    delete ctx;
    return APK_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess ||
      std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    // kernels are compiled for gfx950 only; anything else cannot run them
    delete ctx;
    return APK_ERR_NO_DEVICE;
  }
  if (hipMalloc(&ctx->d_u64, 16 * sizeof(unsigned long long)) != hipSuccess ||
      hipHostMalloc(&ctx->h_pinned, 256, hipHostMallocDefault) != hipSuccess) {
    apk_destroy(ctx);
    return APK_ERR_DEVICE;
  }
  // the flag words ([0] latched, [1] trial stage) are word 5 of d_u64, next to the stage's time-step word 4: one
  // 16-byte read-back fetches both at the end of a cycle (apk_stage_dt_flags_read)
  ctx->d_flags = reinterpret_cast<unsigned *>(ctx->d_u64 + 5);
  (void)hipMemset(ctx->d_u64, 0, 16 * sizeof(unsigned long long));
  {
    const double huge = 1.7976931348623157e308;  // word 15: constant +max, the neutral element of the dt min
    (void)hipMemcpy(ctx->d_u64 + 15, &huge, sizeof(double), hipMemcpyHostToDevice);
  }
  {
    void *dev = nullptr;  // (pinned host memory is mapped into the device's address space by default; if not: copies)
    if (hipHostGetDevicePointer(&dev, ctx->h_pinned, 0) == hipSuccess) ctx->h_pinned_dev = static_cast<unsigned long long *>(dev);
    else (void)hipGetLastError();
  }
  *out = ctx;
  return APK_OK;
}

void apk_destroy(apk_ctx *ctx) {
  if (!ctx) return;
  if (ctx->d_u64) (void)hipFree(ctx->d_u64);  // (d_flags points into it)
  if (ctx->d_mflux) (void)hipFree(ctx->d_mflux);
  if (ctx->d_partial) (void)hipFree(ctx->d_partial);
  if (ctx->d_tagmax) (void)hipFree(ctx->d_tagmax);
  if (ctx->d_mark) (void)hipFree(ctx->d_mark);
  if (ctx->d_du) (void)hipFree(ctx->d_du);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  if (ctx->h_partial) (void)hipHostFree(ctx->h_partial);
  for (auto &sp : ctx->spans) {
    (void)hipEventDestroy(sp.start);
    (void)hipEventDestroy(sp.stop);
  }
  for (auto e : ctx->free_events) (void)hipEventDestroy(e);
  delete ctx;
}

const char *apk_last_error(const apk_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int apk_pack_create(apk_ctx *ctx, const apk_pack_desc *desc, apk_pack **out) {
  if (!ctx || !desc || !out) return APK_ERR_INVALID;
  *out = nullptr;
  if (desc->nblocks <= 0 || !desc->blocks) return set_err(ctx, APK_ERR_INVALID, "empty pack");
  if (desc->nhydro != 5 && desc->nhydro != 9) return set_err(ctx, APK_ERR_INVALID, "nhydro must be 5 or 9");
  if (desc->nscalars < 0 || desc->ng < 1 || desc->nx[0] < 1 || desc->nx[1] < 1 || desc->nx[2] < 1)
    return set_err(ctx, APK_ERR_INVALID, "bad pack geometry");
  if (desc->nx[1] == 1 && desc->nx[2] > 1) return set_err(ctx, APK_ERR_INVALID, "nx2 == 1 requires nx3 == 1");
  {
    const PackView t = make_view(*desc, nullptr);  // (explicit strides: no array may overlap the next one)
    if (desc->stride[0] < 0 || desc->stride[1] < 0 || desc->stride[2] < 0 || t.sj < t.ni || t.sk < t.sj * t.nj || t.sn < t.sk * t.nk)
      return set_err(ctx, APK_ERR_INVALID, "pack strides smaller than the extents they step over");
  }
  apk_pack *p = new (std::nothrow) apk_pack();
  if (!p) return APK_ERR_INVALID;
  p->h_blocks.assign(desc->blocks, desc->blocks + desc->nblocks);
  p->desc = *desc;
  p->desc.blocks = p->h_blocks.data();
  for (int d = 0; d < 3; ++d) {
    p->have_flux[d] = true;
    for (const auto &b : p->h_blocks) p->have_flux[d] = p->have_flux[d] && (b.flux[d] != nullptr);
  }
  for (const auto &b : p->h_blocks) {
    if (!b.cons) {
      delete p;
      return set_err(ctx, APK_ERR_INVALID, "block without cons pointer");
    }
  }
  const size_t bytes = sizeof(apk_block_desc) * desc->nblocks;
  hipError_t e = hipMalloc(&p->d_blocks, bytes);
  if (e == hipSuccess) e = hipMemcpy(p->d_blocks, p->h_blocks.data(), bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (p->d_blocks) (void)hipFree(p->d_blocks);
    delete p;
    return set_err(ctx, APK_ERR_DEVICE, "apk_pack_create", e);
  }
  p->view = make_view(p->desc, p->d_blocks);
  *out = p;
  return APK_OK;
}

void apk_pack_destroy(apk_pack *pack) {
  if (!pack) return;
  if (pack->d_blocks) (void)hipFree(pack->d_blocks);
  delete pack;
}

namespace {
int calculate_fluxes(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg, const apk_eos *eos, double c_h, int faces,
                     apk_stream_t stream, const int *face_list = nullptr, int nlist = 0, const FluxConsInput *from_cons = nullptr) {
  if (!ctx || !md || !valid_eos(eos)) return set_err(ctx, APK_ERR_INVALID, "apk_calculate_fluxes: bad argument");
  int rc = check_cfg(ctx, md, cfg);
  if (rc != APK_OK) return rc;
  for (int d = 0; d < md->view.ndim; ++d)
    if (!md->have_flux[d]) return set_err(ctx, APK_ERR_INVALID, "pack has no flux arrays");
  if (!from_cons)
    for (const auto &b : md->h_blocks)
      if (!b.prim) return set_err(ctx, APK_ERR_INVALID, "block without prim pointer");
  hipStream_t s = as_stream(stream);
  const PackView &pv = md->view;
  ScopedTiming timing(ctx, APK_T_FLUXES, s);
  if ((cfg.riemann == APK_RS_NONE || cfg.riemann == APK_RS_LLF) && faces == 2)
    return set_err(ctx, APK_ERR_UNSUPPORTED, "boundary-plane fluxes are not offered for the none / llf entries");
  if (cfg.riemann == APK_RS_NONE || cfg.riemann == APK_RS_LLF)
    rc = launch_fluxes_misc(pv, cfg.fluid, cfg.riemann, eos->gamma, c_h, s);
  else if (cfg.fluid == APK_FLUID_EULER)
    rc = (cfg.riemann == APK_RS_HLLE) ? launch_fluxes_euler_hlle(pv, cfg.recon, eos->gamma, c_h, s, faces, face_list, nlist, from_cons)
                                      : launch_fluxes_euler_hllc(pv, cfg.recon, eos->gamma, c_h, s, faces, face_list, nlist, from_cons);
  else
    rc = (cfg.riemann == APK_RS_HLLE) ? launch_fluxes_mhd_hlle(pv, cfg.recon, eos->gamma, c_h, s, faces, face_list, nlist, from_cons)
                                      : launch_fluxes_mhd_hlld(pv, cfg.recon, eos->gamma, c_h, s, faces, face_list, nlist, from_cons);
  if (rc != APK_OK) return set_err(ctx, rc, "flux kernel launch failed", hipGetLastError());
  return APK_OK;
}
}  // namespace

int apk_calculate_fluxes(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg, const apk_eos *eos,
                         double c_h, apk_stream_t stream) {
  return calculate_fluxes(ctx, md, cfg, eos, c_h, 0, stream);
}

int apk_calculate_fluxes_tight(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg, const apk_eos *eos,
                               double c_h, apk_stream_t stream) {
  return calculate_fluxes(ctx, md, cfg, eos, c_h, 1, stream);
}

int apk_calculate_fluxes_boundary(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg, const apk_eos *eos,
                                  double c_h, apk_stream_t stream) {
  return calculate_fluxes(ctx, md, cfg, eos, c_h, 2, stream);
}

int apk_calculate_fluxes_boundary_list(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg, const apk_eos *eos,
                                       double c_h, const int *faces, int nfaces, apk_stream_t stream) {
  if (!faces || nfaces < 0) return set_err(ctx, APK_ERR_INVALID, "apk_calculate_fluxes_boundary_list: bad argument");
  return calculate_fluxes(ctx, md, cfg, eos, c_h, 2, stream, faces, nfaces);
}

int apk_calculate_fluxes_boundary_list_from_cons(apk_ctx *ctx, const apk_pack *md, apk_flux_cfg cfg, const apk_eos *eos,
                                                 double c_h, const int *faces, int nfaces, long long cons_delta,
                                                 apk_stream_t stream) {
  if (!faces || nfaces < 0 || !valid_eos(eos))
    return set_err(ctx, APK_ERR_INVALID, "apk_calculate_fluxes_boundary_list_from_cons: bad argument");
  // (the stencil cells are converted by the lean ConsToPrim, and nothing writes a floored value back)
  if (!eos_is_lean(*eos) || eos->dfloor > 0.0 || eos->efloor > 0.0)
    return set_err(ctx, APK_ERR_UNSUPPORTED, "apk_calculate_fluxes_boundary_list_from_cons: floors / ceilings need the stored primitives");
  if (cfg.riemann == APK_RS_NONE || cfg.riemann == APK_RS_LLF)
    return set_err(ctx, APK_ERR_UNSUPPORTED, "boundary-plane fluxes are not offered for the none / llf entries");
  const StageConsts k = make_stage_consts(eos->gamma, c_h, *eos);
  const FluxConsInput ci{(int64_t)cons_delta, *eos, k.eos_gm1, k.vceil_sq, k.pfloor_over_gm1};
  const int rc = calculate_fluxes(ctx, md, cfg, eos, c_h, 2, stream, faces, nfaces, &ci);
  return rc;
}

int apk_update_with_flux_divergence(apk_ctx *ctx, const apk_pack *u0, const apk_pack *u1,
                                    double gam0, double gam1, double beta_dt,
                                    apk_stream_t stream) {
  if (!ctx || !u0 || !u1 || !same_shape(u0, u1))
    return set_err(ctx, APK_ERR_INVALID, "apk_update_with_flux_divergence: bad argument");
  for (int d = 0; d < u0->view.ndim; ++d)
    if (!u0->have_flux[d]) return set_err(ctx, APK_ERR_INVALID, "u0 pack has no flux arrays");
  ScopedTiming timing(ctx, APK_T_UPDATE, as_stream(stream));
  int rc = launch_update_flux_div(u0->view, u1->view, gam0, gam1, beta_dt, as_stream(stream));
  if (rc != APK_OK) return set_err(ctx, rc, "update kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_dedner_source(apk_ctx *ctx, const apk_pack *md, int extended, double alpha, double c_h,
                      double mindx, double beta_dt, apk_stream_t stream) {
  if (!ctx || !md || md->view.nhydro != 9 || !(mindx > 0.0))
    return set_err(ctx, APK_ERR_INVALID, "apk_dedner_source: bad argument");
  // Mignone & Tzeferacos 2010 (27): dedner_source.cpp:32
  const double coeff = std::exp(-alpha * c_h * beta_dt / mindx);
  ScopedTiming timing(ctx, APK_T_DEDNER, as_stream(stream));
  int rc = launch_dedner(md->view, extended, coeff, beta_dt, as_stream(stream));
  if (rc != APK_OK) return set_err(ctx, rc, "dedner kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_stage_fused(apk_ctx *ctx, const apk_pack *u0, const apk_pack *u1,
                    const apk_stage_args *a, apk_stream_t stream) {
  if (!ctx || !u0 || !u1 || !a || !same_shape(u0, u1) || !valid_eos(&a->eos))
    return set_err(ctx, APK_ERR_INVALID, "apk_stage_fused: bad argument");
  int rc = check_cfg(ctx, u0, a->cfg);
  if (rc != APK_OK) return rc;
  if (a->cfg.riemann == APK_RS_NONE || a->cfg.riemann == APK_RS_LLF)
    return set_err(ctx, APK_ERR_UNSUPPORTED, "fused stage: none/llf solvers use the flux-array path");
  if (a->dedner != 0 && (a->cfg.fluid != APK_FLUID_GLMMHD || !(a->mindx > 0.0)))
    return set_err(ctx, APK_ERR_INVALID, "fused stage: Dedner source needs glmmhd and mindx > 0");
  if (a->phase < 0 || a->phase > 2 || (a->phase == 1) != (a->window != nullptr) ||
      (a->phase == 1 && (a->window_rl < 3 || a->window_rl > u0->view.ni || a->window_rows < 1 ||
                         a->window_rows > u0->view.nx2)))
    return set_err(ctx, APK_ERR_INVALID, "fused stage: phase / window mismatch");
  if (a->phase == 1 && a->estimate_dt && a->cfg.recon == APK_RC_DC && u0->view.ndim == 3)
    return set_err(ctx, APK_ERR_UNSUPPORTED, "fused stage: estimate_dt is not available in a split 3-D donor-cell stage");
  if (a->fill_derived < 0 || a->fill_derived > 3) return set_err(ctx, APK_ERR_INVALID, "fused stage: fill_derived must be 0, 1, 2 or 3");
  if (a->fill_derived == 3 && !a->estimate_dt)
    return set_err(ctx, APK_ERR_INVALID, "fused stage: fill_derived = 3 (primitives for the time-step estimate only) needs estimate_dt");
  if (a->prim_from_cons < 0 || a->prim_from_cons > 2) return set_err(ctx, APK_ERR_INVALID, "fused stage: prim_from_cons must be 0, 1 or 2");
  if (a->prim_from_cons == 2 && a->cons_out_delta == 0)
    return set_err(ctx, APK_ERR_INVALID, "fused stage: prim_from_cons = 2 (input = u0.cons) needs an out-of-place result (cons_out_delta)");
  if (a->prim_from_cons) {
    for (const apk_block_desc &b : u1->h_blocks)
      if (!b.cons) return set_err(ctx, APK_ERR_INVALID, "fused stage: prim_from_cons needs u1.cons");
  }
  if (a->fill_derived == 2) {
    for (const apk_block_desc &b : u1->h_blocks)
      if (!b.prim) return set_err(ctx, APK_ERR_INVALID, "fused stage: fill_derived = 2 needs prim arrays in u1");
    for (size_t b = 0; b < u0->h_blocks.size(); ++b)
      if (u0->h_blocks[b].prim == u1->h_blocks[b].prim)
        return set_err(ctx, APK_ERR_INVALID, "fused stage: fill_derived = 2 needs u1.prim distinct from u0.prim");
  }
  double coeff = 1.0;
  if (a->dedner != 0) coeff = std::exp(-a->glmmhd_alpha * a->c_h * a->beta_dt / a->mindx);
  rc = launch_stage_fused(ctx, u0->view, u1->view, *a, coeff, as_stream(stream));
  if (rc == APK_ERR_UNSUPPORTED) return set_err(ctx, rc, "fused stage: option combination not supported (scalars, 1-D/extended-Dedner fill_derived, split 1-D or 3-D donor-cell stage, fill_derived = 3 / prim_from_cons outside the lean two-kernel / donor-cell stage)");
  if (rc != APK_OK) return set_err(ctx, rc, "fused stage kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_stage_unphysical_read(apk_ctx *ctx, long long *count, apk_stream_t stream) {
  if (!ctx || !count) return APK_ERR_INVALID;
  hipStream_t s = as_stream(stream);
  auto *h = static_cast<unsigned long long *>(ctx->h_pinned);
  APK_HIP_TRY(ctx, hipMemcpyAsync(h + 6, ctx->d_u64 + 6, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  APK_HIP_TRY(ctx, hipStreamSynchronize(s));
  *count = (long long)h[6];
  return APK_OK;
}

int apk_stage_split_axis(const apk_pack *u0, const apk_flux_cfg *cfg, int fill_derived) {
  if (!u0 || !cfg) return 0;
  apk::StageParams sp{};
  sp.prim_to_u1 = (fill_derived >= 2) ? 1 : 0;
  const int extra = fill_derived ? apk::EXTRA_C2P : apk::EXTRA_NONE;
  if (u0->view.ndim == 3 && cfg->recon == APK_RC_DC) return 0;  // single-kernel stage: 3-D index windows
  return apk::two_kernel_stage_applies(u0->view, cfg->recon, extra, sp) ? 3 : 1;
}

int apk_stage_x1_halo(const apk_pack *u0, const apk_flux_cfg *cfg, const apk_eos *eos, int fill_derived, int dedner, int prim_from_cons) {
  if (!u0 || !cfg || !eos) return 0;
  if (cfg->riemann == APK_RS_NONE || cfg->riemann == APK_RS_LLF || u0->view.nvar != u0->view.nhydro) return 0;
  apk::StageParams sp{};
  sp.eos = *eos;
  sp.dedner = dedner;
  sp.prim_to_u1 = (fill_derived >= 2) ? 1 : 0;
  sp.no_prim_store = (fill_derived == 3) ? 1 : 0;
  sp.prim_from_cons = prim_from_cons;
  sp.out_delta = (prim_from_cons == 2) ? 1 : 0;  // (such a stage writes its result elsewhere: the caller's business)
  const int extra = fill_derived ? (fill_derived == 3 ? apk::EXTRA_C2P_DT : apk::EXTRA_C2P) : apk::EXTRA_NONE;
  if (!apk::x1_halo_stage_ok(u0->view, cfg->recon, extra, sp)) return 0;
  if (cfg->recon != APK_RC_DC && prim_from_cons && apk_stage_single_march(u0, cfg)) return 0;
  return 1;
}

int apk_stage_single_march(const apk_pack *u0, const apk_flux_cfg *cfg) {
  if (!u0 || !cfg) return 0;
  apk::StageParams sp{};  // a whole-block lean stage whose input is a conserved state
  sp.prim_from_cons = 1;
  sp.eos.vceil = sp.eos.eceil = __builtin_inf();
  sp.eos.pfloor = sp.eos.dfloor = sp.eos.efloor = -1.0;
  if (!apk::two_kernel_stage_applies(u0->view, cfg->recon, apk::EXTRA_NONE, sp)) return 0;
  if (cfg->fluid == APK_FLUID_EULER && cfg->recon == APK_RC_PLM)
    return apk::single_march_stage_applies<APK_FLUID_EULER, APK_RC_PLM>(u0->view, apk::EXTRA_NONE, sp) ? 1 : 0;
  return 0;
}

int apk_cons_to_prim(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos,
                     apk_stream_t stream) {
  if (!ctx || !md || !valid_eos(eos) || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_cons_to_prim: bad argument");
  ScopedTiming timing(ctx, APK_T_C2P, as_stream(stream));
  int rc = launch_cons_to_prim(md->view, fluid, *eos, ctx->d_flags, as_stream(stream));
  if (rc != APK_OK) return set_err(ctx, rc, "cons_to_prim kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_cons_to_prim_dt(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, int ghost_depth, apk_stream_t stream) {
  return apk_cons_to_prim_dt_skip(ctx, md, fluid, eos, ghost_depth, nullptr, stream);
}

int apk_cons_to_prim_dt_skip(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, int ghost_depth, const int *face_neighbor,
                             apk_stream_t stream) {
  if (!ctx || !md || !valid_eos(eos) || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_cons_to_prim_dt: bad argument");
  hipStream_t s = as_stream(stream);
  unsigned long long *dt_bits = ctx->d_u64 + 4;  // the stage's word: apk_stage_dt_read / apk_stage_dt_flags_read
  if (apk::prepare_dt_word(ctx, s) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "time-step word reset", hipGetLastError());
  ScopedTiming timing(ctx, APK_T_C2P, s);
  int rc = launch_cons_to_prim(md->view, fluid, *eos, ctx->d_flags, s, false, nullptr, 0, false, face_neighbor, dt_bits, ghost_depth);
  if (rc != APK_OK) return set_err(ctx, rc, "cons_to_prim kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_cons_to_prim_dt_select(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, int ghost_depth, const int *face_neighbor,
                               unsigned store_vars, apk_stream_t stream) {
  if (!ctx || !md || !valid_eos(eos) || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9) || ghost_depth < 0 || ghost_depth > md->view.ng)
    return set_err(ctx, APK_ERR_INVALID, "apk_cons_to_prim_dt_select: bad argument");
  if (md->view.nvar != md->view.nhydro || (store_vars >> md->view.nhydro) != 0u)
    return set_err(ctx, APK_ERR_INVALID, "apk_cons_to_prim_dt_select: store_vars names the hydro primitives of a pack without passive scalars");
  // (a floor or ceiling writes conserved values back that belong with ALL primitives of the cell)
  if (!eos_is_lean(*eos) || eos->dfloor > 0.0 || eos->efloor > 0.0)
    return set_err(ctx, APK_ERR_UNSUPPORTED, "apk_cons_to_prim_dt_select: floors / ceilings need the full ConsToPrim");
  hipStream_t s = as_stream(stream);
  unsigned long long *dt_bits = ctx->d_u64 + 4;  // the stage's word: apk_stage_dt_read / apk_stage_dt_flags_read
  if (apk::prepare_dt_word(ctx, s) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "time-step word reset", hipGetLastError());
  ScopedTiming timing(ctx, APK_T_C2P, s);
  // (store_vars = every primitive is the pass apk_cons_to_prim_dt_skip runs; ~0u is the kernels' word for it)
  const unsigned all = (1u << md->view.nhydro) - 1u;
  int rc = launch_cons_to_prim(md->view, fluid, *eos, ctx->d_flags, s, false, nullptr, 0, false, face_neighbor, dt_bits, ghost_depth,
                               store_vars == all ? ~0u : store_vars);
  if (rc != APK_OK) return set_err(ctx, rc, "cons_to_prim kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_cons_to_prim_faces(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, apk_stream_t stream) {
  if (!ctx || !md || !valid_eos(eos) || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_cons_to_prim_faces: bad argument");
  ScopedTiming timing(ctx, APK_T_C2P, as_stream(stream));
  int rc = launch_cons_to_prim(md->view, fluid, *eos, ctx->d_flags, as_stream(stream), false, nullptr, 0, true);
  if (rc != APK_OK) return set_err(ctx, rc, "cons_to_prim kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_cons_to_prim_faces_skip(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, const int *face_neighbor,
                                apk_stream_t stream) {
  if (!ctx || !md || !valid_eos(eos) || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9) || !face_neighbor)
    return set_err(ctx, APK_ERR_INVALID, "apk_cons_to_prim_faces_skip: bad argument");
  ScopedTiming timing(ctx, APK_T_C2P, as_stream(stream));
  int rc = launch_cons_to_prim(md->view, fluid, *eos, ctx->d_flags, as_stream(stream), false, nullptr, 0, true, face_neighbor);
  if (rc != APK_OK) return set_err(ctx, rc, "cons_to_prim kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_cons_to_prim_faces_dt(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, const int *face_neighbor,
                              apk_stream_t stream) {
  if (!ctx || !md || !valid_eos(eos) || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_cons_to_prim_faces_dt: bad argument");
  hipStream_t s = as_stream(stream);
  unsigned long long *dt_bits = ctx->d_u64 + 4;  // the stage's word: apk_stage_dt_read / apk_stage_dt_flags_read
  if (apk::prepare_dt_word(ctx, s) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "time-step word reset", hipGetLastError());
  ScopedTiming timing(ctx, APK_T_C2P, s);
  int rc = launch_cons_to_prim(md->view, fluid, *eos, ctx->d_flags, s, false, nullptr, 0, true, face_neighbor, dt_bits);
  if (rc != APK_OK) return set_err(ctx, rc, "cons_to_prim kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_cons_to_prim_ghosts(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos,
                            apk_stream_t stream) {
  if (!ctx || !md || !valid_eos(eos) || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_cons_to_prim_ghosts: bad argument");
  ScopedTiming timing(ctx, APK_T_C2P, as_stream(stream));
  int rc = launch_cons_to_prim(md->view, fluid, *eos, ctx->d_flags, as_stream(stream), true);
  if (rc != APK_OK) return set_err(ctx, rc, "cons_to_prim kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_cons_to_prim_ghosts_split(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos,
                                  const unsigned *late_regions, int part, apk_stream_t stream) {
  if (!ctx || !md || !valid_eos(eos) || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9) || !late_regions || (part != 1 && part != 2))
    return set_err(ctx, APK_ERR_INVALID, "apk_cons_to_prim_ghosts_split: bad argument");
  ScopedTiming timing(ctx, APK_T_C2P, as_stream(stream));
  int rc = launch_cons_to_prim(md->view, fluid, *eos, ctx->d_flags, as_stream(stream), true, late_regions, part);
  if (rc != APK_OK) return set_err(ctx, rc, "cons_to_prim kernel launch failed", hipGetLastError());
  return APK_OK;
}

int apk_stage_dt_read(apk_ctx *ctx, double cfl, double *dt_out, apk_stream_t stream) {
  if (!ctx || !dt_out) return APK_ERR_INVALID;
  if (ctx->last_stage_min_valid) {  // (apk_stage_dt_flags_read has taken the minimum out of the device word)
    *dt_out = cfl * ctx->last_stage_min;
    return APK_OK;
  }
  hipStream_t s = as_stream(stream);
  auto *h = static_cast<unsigned long long *>(ctx->h_pinned);
  APK_HIP_TRY(ctx, hipMemcpyAsync(h + 4, ctx->d_u64 + 4, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  APK_HIP_TRY(ctx, hipStreamSynchronize(s));
  double m;
  std::memcpy(&m, h + 4, sizeof(m));
  *dt_out = cfl * m;  // hydro.cpp:909
  return APK_OK;
}

int apk_estimate_timestep(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos,
                          double cfl, double *dt_out, apk_stream_t stream) {
  if (!ctx || !md || !valid_eos(eos) || !dt_out ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_estimate_timestep: bad argument");
  hipStream_t s = as_stream(stream);
  const double huge = std::numeric_limits<double>::max();
  unsigned long long bits;
  std::memcpy(&bits, &huge, sizeof(bits));
  auto *h = static_cast<unsigned long long *>(ctx->h_pinned);
  h[0] = bits;
  APK_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_u64, h, sizeof(bits), hipMemcpyHostToDevice, s));
  int rc;
  {
    ScopedTiming timing(ctx, APK_T_MIN_DT, s);
    rc = launch_min_dt(md->view, fluid, eos->gamma, ctx->d_u64, s);
  }
  if (rc != APK_OK) return set_err(ctx, rc, "min_dt kernel launch failed", hipGetLastError());
  APK_HIP_TRY(ctx, hipMemcpyAsync(h + 1, ctx->d_u64, sizeof(bits), hipMemcpyDeviceToHost, s));
  APK_HIP_TRY(ctx, hipStreamSynchronize(s));
  double m;
  std::memcpy(&m, h + 1, sizeof(m));
  *dt_out = cfl * m;  // hydro.cpp:909
  return APK_OK;
}

int apk_first_order_flux_correct(apk_ctx *ctx, const apk_pack *u0, const apk_pack *u1, int fluid,
                                 const apk_eos *eos, double c_h, double gam0, double gam1,
                                 double beta_dt, long long *num_corrected, apk_stream_t stream) {
  if (!ctx || !u0 || !u1 || !same_shape(u0, u1) || !valid_eos(eos) ||
      u0->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_first_order_flux_correct: bad argument");
  for (int d = 0; d < u0->view.ndim; ++d)
    if (!u0->have_flux[d]) return set_err(ctx, APK_ERR_INVALID, "u0 pack has no flux arrays");
  hipStream_t s = as_stream(stream);
  int rc = ensure_mark(ctx, (size_t)u0->view.nblocks * (size_t)u0->view.sn);
  if (rc != APK_OK) return rc;
  auto *h = static_cast<unsigned long long *>(ctx->h_pinned);
  long long total = 0;
  unsigned long long corrected = 0;
  int attempts = 0;
  do {  // hydro.cpp:1265-1339: <= 4 attempts, one host sync per attempt
    APK_HIP_TRY(ctx, hipMemsetAsync(ctx->d_u64 + 2, 0, sizeof(unsigned long long), s));
    rc = launch_fofc_mark(u0->view, u1->view, fluid, gam0, gam1, beta_dt, attempts, ctx->d_mark,
                          ctx->d_u64 + 2, s);
    if (rc != APK_OK) return set_err(ctx, rc, "fofc mark kernel launch failed", hipGetLastError());
    APK_HIP_TRY(ctx, hipMemcpyAsync(h + 2, ctx->d_u64 + 2, sizeof(unsigned long long),
                                    hipMemcpyDeviceToHost, s));
    APK_HIP_TRY(ctx, hipStreamSynchronize(s));
    corrected = h[2];
    if (corrected > 0) {
      rc = launch_fofc_fix(u0->view, fluid, eos->gamma, c_h, ctx->d_mark, s);
      if (rc != APK_OK) return set_err(ctx, rc, "fofc fix kernel launch failed", hipGetLastError());
    }
    total += (long long)corrected;
    attempts += 1;
  } while (corrected > 0 && attempts < 4);
  if (num_corrected) *num_corrected = total;
  return APK_OK;
}

int apk_count_unphysical(apk_ctx *ctx, const apk_pack *md, int fluid, long long *count, apk_stream_t stream) {
  if (!ctx || !md || !count || md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_count_unphysical: bad argument");
  hipStream_t s = as_stream(stream);
  auto *h = static_cast<unsigned long long *>(ctx->h_pinned);
  APK_HIP_TRY(ctx, hipMemsetAsync(ctx->d_u64 + 2, 0, sizeof(unsigned long long), s));
  int rc = launch_count_unphysical(md->view, fluid, ctx->d_u64 + 2, s);
  if (rc != APK_OK) return set_err(ctx, rc, "count_unphysical launch failed", hipGetLastError());
  APK_HIP_TRY(ctx, hipMemcpyAsync(h + 2, ctx->d_u64 + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  APK_HIP_TRY(ctx, hipStreamSynchronize(s));
  *count = (long long)h[2];
  return APK_OK;
}

int apk_history(apk_ctx *ctx, const apk_pack *md, int fluid, double *out8, apk_stream_t stream) {
  if (!ctx || !md || !out8 || md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_history: bad argument");
  hipStream_t s = as_stream(stream);
  int nwg = 0;
  launch_history(md->view, fluid, nullptr, &nwg, nullptr, s);
  int rc = ensure_partial(ctx, (size_t)nwg * 8 + 8);
  if (rc != APK_OK) return rc;
  double *d_out = ctx->d_partial + (size_t)nwg * 8;
  rc = launch_history(md->view, fluid, ctx->d_partial, &nwg, d_out, s);
  if (rc != APK_OK) return set_err(ctx, rc, "history kernel launch failed", hipGetLastError());
  auto *h = static_cast<double *>(ctx->h_pinned) + 8;
  APK_HIP_TRY(ctx, hipMemcpyAsync(h, d_out, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
  APK_HIP_TRY(ctx, hipStreamSynchronize(s));
  std::memcpy(out8, h, 8 * sizeof(double));
  return APK_OK;
}

int apk_stage_dt_flags_read(apk_ctx *ctx, double cfl, double *dt_out, unsigned *flags, apk_stream_t stream) {
  if (!ctx || !dt_out || !flags) return APK_ERR_INVALID;
  hipStream_t s = as_stream(stream);
  auto *h = static_cast<unsigned long long *>(ctx->h_pinned);
  if (ctx->last_stage_min_valid) {  // (already consumed: the word on the device has been reset)
    *dt_out = cfl * ctx->last_stage_min;
    return apk_poll_device_flags(ctx, flags, stream);
  }
  if (ctx->h_pinned_dev) {
    // one small kernel hands words 4 / 5 (and the tag criteria an apk_tag_blocks_begin left pending) to the host and
    // leaves the device words ready for the next cycle
    if (apk::launch_cycle_gather(ctx, s) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "cycle gather launch", hipGetLastError());
  } else {
    // words 4 (the stage's minimum) and 5 (the flag words) in one copy
    APK_HIP_TRY(ctx, hipMemcpyAsync(h + 4, ctx->d_u64 + 4, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    APK_HIP_TRY(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof(unsigned), s));
    // (the same consume-on-read semantics as the gather kernel's: word 4 back to +max, ready for the next reduction)
    APK_HIP_TRY(ctx, hipMemcpyAsync(ctx->d_u64 + 4, ctx->d_u64 + 15, sizeof(double), hipMemcpyDeviceToDevice, s));
    ctx->dt_word_clean = true;
    ctx->clean_stream = s;
  }
  APK_HIP_TRY(ctx, hipStreamSynchronize(s));
  double m;
  std::memcpy(&m, h + 4, sizeof(m));
  ctx->last_stage_min = m;
  ctx->last_stage_min_valid = true;
  *dt_out = cfl * m;  // hydro.cpp:909
  unsigned f[2];
  std::memcpy(f, h + 5, sizeof(f));
  *flags = f[0];
  return APK_OK;
}

namespace {
__global__ void commit_trial_flags_kernel(unsigned *f) {
  if (f[1]) atomicOr(f, f[1]);
  f[1] = 0u;
}
}  // namespace

int apk_trial_flags(apk_ctx *ctx, int keep, apk_stream_t stream) {
  if (!ctx) return APK_ERR_INVALID;
  hipStream_t s = as_stream(stream);
  if (keep) {
    hipLaunchKernelGGL(commit_trial_flags_kernel, dim3(1), dim3(1), 0, s, ctx->d_flags);
    if (hipGetLastError() != hipSuccess) return set_err(ctx, APK_ERR_DEVICE, "apk_trial_flags: launch failed");
  } else {
    APK_HIP_TRY(ctx, hipMemsetAsync(ctx->d_flags + 1, 0, sizeof(unsigned), s));
  }
  return APK_OK;
}

int apk_poll_device_flags(apk_ctx *ctx, unsigned *flags, apk_stream_t stream) {
  if (!ctx || !flags) return APK_ERR_INVALID;
  hipStream_t s = as_stream(stream);
  auto *h = reinterpret_cast<unsigned *>(static_cast<char *>(ctx->h_pinned) + 192);
  APK_HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_flags, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  APK_HIP_TRY(ctx, hipMemsetAsync(ctx->d_flags, 0, sizeof(unsigned), s));
  APK_HIP_TRY(ctx, hipStreamSynchronize(s));
  *flags = *h;
  return APK_OK;
}

int apk_copy_plan_create(apk_ctx *ctx, const apk_copy_region *regions, int n,
                         apk_copy_plan **out) {
  if (!ctx || !out || n < 0 || (n > 0 && !regions)) return APK_ERR_INVALID;
  apk_copy_plan *p = new (std::nothrow) apk_copy_plan();
  if (!p) return APK_ERR_INVALID;
  p->n = n;
  for (int r = 0; r < n; ++r) {
    const int64_t c = (int64_t)regions[r].ext[0] * regions[r].ext[1] * regions[r].ext[2];
    if (c > p->max_cells) p->max_cells = c;
    if (c * regions[r].nvar > p->max_items) p->max_items = c * regions[r].nvar;
  }
  if (n > 0) {
    std::vector<apk_copy_chunk> by_items, by_cells;
    for (int r = 0; r < n; ++r) {
      const int64_t c = (int64_t)regions[r].ext[0] * regions[r].ext[1] * regions[r].ext[2];
      const int64_t items = c * regions[r].nvar;
      if (items >= (int64_t)1 << 31) {
        delete p;
        return set_err(ctx, APK_ERR_INVALID, "apk_copy_plan_create: box too large");
      }
      for (int64_t f = 0; f < items; f += kCopyChunkItems) by_items.push_back({r, (int)f});
      for (int64_t f = 0; f < c; f += kCopyChunkCells) by_cells.push_back({r, (int)f});
    }
    p->nchunks_items = (int)by_items.size();
    p->nchunks_cells = (int)by_cells.size();
    hipError_t e = hipMalloc(&p->d_regions, sizeof(apk_copy_region) * n);
    if (e == hipSuccess)
      e = hipMemcpy(p->d_regions, regions, sizeof(apk_copy_region) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess && !by_items.empty()) {
      e = hipMalloc(&p->d_chunks_items, sizeof(apk_copy_chunk) * by_items.size());
      if (e == hipSuccess)
        e = hipMemcpy(p->d_chunks_items, by_items.data(), sizeof(apk_copy_chunk) * by_items.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && !by_cells.empty()) {
      e = hipMalloc(&p->d_chunks_cells, sizeof(apk_copy_chunk) * by_cells.size());
      if (e == hipSuccess)
        e = hipMemcpy(p->d_chunks_cells, by_cells.data(), sizeof(apk_copy_chunk) * by_cells.size(), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
      apk_copy_plan_destroy(p);
      return set_err(ctx, APK_ERR_DEVICE, "apk_copy_plan_create", e);
    }
  }
  *out = p;
  return APK_OK;
}

void apk_copy_plan_destroy(apk_copy_plan *plan) {
  if (!plan) return;
  if (plan->d_regions) (void)hipFree(plan->d_regions);
  if (plan->d_chunks_items) (void)hipFree(plan->d_chunks_items);
  if (plan->d_chunks_cells) (void)hipFree(plan->d_chunks_cells);
  delete plan;
}

int apk_copy_plan_run(apk_ctx *ctx, const apk_copy_plan *plan, apk_stream_t stream) {
  if (!ctx || !plan) return APK_ERR_INVALID;
  if (plan->n <= 0) return APK_OK;
  ScopedTiming timing(ctx, APK_T_COPY, as_stream(stream));
  int rc = launch_copy_regions(*plan, as_stream(stream));
  if (rc != APK_OK) return set_err(ctx, rc, "copy kernel launch failed", hipGetLastError());
  return APK_OK;
}

namespace {
int copy_plan_run_c2p(apk_ctx *ctx, const apk_copy_plan *plan, int fluid, const apk_eos *eos, int64_t prim_delta, int latch_flags,
                      bool prim_only, apk_stream_t stream) {
  if (!ctx || !plan || !valid_eos(eos) || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD))
    return set_err(ctx, APK_ERR_INVALID, "apk_copy_plan_run_c2p: bad argument");
  if (eos->dfloor > 0.0 || eos->pfloor > 0.0 || eos->efloor > 0.0 || eos->vceil < 1.0e300 || eos->eceil < 1.0e300)
    return set_err(ctx, APK_ERR_UNSUPPORTED, "apk_copy_plan_run_c2p: floors / ceilings are active; copy, then apk_cons_to_prim_ghosts");
  if (plan->n <= 0) return APK_OK;
  ScopedTiming timing(ctx, APK_T_COPY, as_stream(stream));
  int rc = launch_copy_regions(*plan, as_stream(stream), fluid, eos, latch_flags ? ctx->d_flags : nullptr, prim_delta, prim_only);
  if (rc != APK_OK) return set_err(ctx, rc, "copy kernel launch failed", hipGetLastError());
  return APK_OK;
}
}  // namespace

int apk_copy_plan_run_c2p(apk_ctx *ctx, const apk_copy_plan *plan, int fluid, const apk_eos *eos, int64_t prim_delta,
                          int latch_flags, apk_stream_t stream) {
  return copy_plan_run_c2p(ctx, plan, fluid, eos, prim_delta, latch_flags, false, stream);
}

int apk_copy_plan_run_c2p_prim_only(apk_ctx *ctx, const apk_copy_plan *plan, int fluid, const apk_eos *eos, int64_t prim_delta,
                                    int latch_flags, apk_stream_t stream) {
  return copy_plan_run_c2p(ctx, plan, fluid, eos, prim_delta, latch_flags, true, stream);
}

int apk_kernel_timing_enable(apk_ctx *ctx, int on) {
  if (!ctx) return APK_ERR_INVALID;
  ctx->timing_on = on != 0;
  return APK_OK;
}

int apk_kernel_timing_read(apk_ctx *ctx, int slot, double *total_ms, long long *launches) {
  if (!ctx || slot < 0 || slot >= APK_T_COUNT) return APK_ERR_INVALID;
  // fold every completed span into the per-slot totals, then hand the events back
  for (auto &sp : ctx->spans) {
    float ms = 0.0f;
    hipError_t e = hipEventSynchronize(sp.stop);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, sp.start, sp.stop);
    if (e != hipSuccess) return set_err(ctx, APK_ERR_DEVICE, "apk_kernel_timing_read", e);
    ctx->timing_ms[sp.slot] += (double)ms;
    ctx->timing_n[sp.slot] += 1;
    ctx->free_events.push_back(sp.start);
    ctx->free_events.push_back(sp.stop);
  }
  ctx->spans.clear();
  if (total_ms) *total_ms = ctx->timing_ms[slot];
  if (launches) *launches = ctx->timing_n[slot];
  ctx->timing_ms[slot] = 0.0;
  ctx->timing_n[slot] = 0;
  return APK_OK;
}

}  // extern "C"
