// fused_dispatch.hip -- entry of the fused stage path: sizes the du workspace of the handle
// and picks the (fluid, riemann) family.
#include "fused_kernel.hpp"

namespace apk {

int launch_stage_fused(apk_ctx *ctx, const PackView &u0, const PackView &u1,
                       const apk_stage_args &a, double dedner_coeff, hipStream_t s) {
  StageParams sp;
  sp.mflux = nullptr;
  if (u0.nvar != u0.nhydro) {
    // passive scalars: the sweeps leave the mass fluxes in a workspace of the handle
    const size_t need = 3 * (size_t)u0.nblocks * (size_t)u0.sn;
    if (need > ctx->mflux_cap) {
      if (ctx->d_mflux) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(ctx->d_mflux);
      }
      ctx->d_mflux = nullptr;
      ctx->mflux_cap = 0;
      if (hipMalloc(&ctx->d_mflux, need * sizeof(double)) != hipSuccess) return APK_ERR_DEVICE;
      ctx->mflux_cap = need;
    }
    sp.mflux = ctx->d_mflux;
  }
  sp.gamma = a.eos.gamma;
  sp.c_h = a.c_h;
  sp.k = make_stage_consts(a.eos.gamma, a.c_h, a.eos);
  sp.gam0 = a.gam0;
  sp.gam1 = a.gam1;
  sp.beta_dt = a.beta_dt;
  sp.dedner = a.dedner;
  sp.dedner_coeff = dedner_coeff;
  sp.du = nullptr;
  sp.du_first = 0;
  sp.du_pitch = 0;
  sp.ctx = ctx;
  sp.eos = a.eos;
  sp.flags = ctx->d_flags + (a.trial ? 1 : 0);
  sp.dt_bits = ctx->d_u64 + 4;  // word 4: min of the finishing sweep (apk_stage_dt_read)
  sp.bad_count = nullptr;
  sp.out_delta = a.cons_out_delta;
  sp.face_nbr = a.face_neighbor;
  sp.x1_blocks = nullptr;
  sp.x1_recv_depth = sp.x1_send_depth = sp.x1_send_field = 0;
  if (a.x1_halo) {
    if (!a.x1_halo->blocks || a.x1_halo->send_field < 0 || a.x1_halo->send_field > 1) return APK_ERR_INVALID;
    sp.x1_blocks = a.x1_halo->blocks;
    sp.x1_recv_depth = a.x1_halo->recv_depth;
    sp.x1_send_depth = a.x1_halo->send_depth;
    sp.x1_send_field = a.x1_halo->send_field;
  }
  if (a.cons_store < 0 || a.cons_store > 2) return APK_ERR_INVALID;
  // (only the stage whose primitives go out of place may drop its conserved result; a windowed phase keeps everything)
  sp.cons_store = (a.fill_derived == 2 && !a.trial && a.cons_out_delta == 0) ? a.cons_store : 0;
  if (a.cons_out_delta != 0 && u0.nvar != u0.nhydro) return APK_ERR_UNSUPPORTED;  // (the scalar kernels update in place)
  if (a.count_unphysical) {     // word 6: cells failing FirstOrderFluxCorrect's test (apk_stage_unphysical_read)
    if (u0.nvar != u0.nhydro) return APK_ERR_UNSUPPORTED;  // (scalars are updated by their own kernel)
    sp.bad_count = ctx->d_u64 + 6;
    if (a.phase != 2 && hipMemsetAsync(sp.bad_count, 0, sizeof(unsigned long long), s) != hipSuccess) return APK_ERR_DEVICE;
  }
  int extra = EXTRA_NONE;
  if (a.fill_derived < 0 || a.fill_derived > 3 || (a.fill_derived == 3 && !a.estimate_dt)) return APK_ERR_INVALID;
  sp.prim_to_u1 = (a.fill_derived >= 2) ? 1 : 0;
  sp.no_prim_store = (a.fill_derived == 3) ? 1 : 0;  // (the stage forms that cannot honour it refuse: launch_fused_stage)
  sp.prim_from_cons = a.prim_from_cons;  // (0, 1 or 2: apk_stage_fused has checked)
  sp.phase = a.phase;
  sp.window = (a.phase == 1) ? a.window : nullptr;
  sp.window_rl = a.window_rl;
  sp.window_rows = a.window_rows;
  if (a.fill_derived) {
    // in-place prim replacement is only safe when the finishing sweep is a march (x2/x3) and
    // the extended Dedner source does not read neighbouring primitives (out of place it may)
    if (u0.ndim == 1 || (a.dedner == 2 && a.fill_derived != 2)) return APK_ERR_UNSUPPORTED;
    extra = a.estimate_dt ? EXTRA_C2P_DT : EXTRA_C2P;
  } else if (a.estimate_dt) {
    return APK_ERR_INVALID;
  }
  if (extra == EXTRA_C2P_DT && a.phase != 1) {  // (a split stage reduces dt in phase 2)
    if (prepare_dt_word(ctx, s) != APK_OK) return APK_ERR_DEVICE;
  }
  if (u0.ndim > 1) {
    const size_t need = (size_t)u0.nblocks * (size_t)u0.nvar * (size_t)u0.sn;
    if (need > ctx->du_cap) {  // workspace owned by the handle, grown on demand
      if (ctx->d_du) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(ctx->d_du);
      }
      ctx->d_du = nullptr;
      ctx->du_cap = 0;
      if (hipMalloc(&ctx->d_du, need * sizeof(double)) != hipSuccess) return APK_ERR_DEVICE;
      ctx->du_cap = need;
    }
    sp.du = ctx->d_du;
  }
  if (a.cfg.fluid == APK_FLUID_EULER) {
    if (a.cfg.riemann == APK_RS_HLLE) return launch_fused_euler_hlle(u0, u1, a.cfg.recon, sp, extra, s);
    if (a.cfg.riemann == APK_RS_HLLC) return launch_fused_euler_hllc(u0, u1, a.cfg.recon, sp, extra, s);
  } else if (a.cfg.fluid == APK_FLUID_GLMMHD) {
    if (a.cfg.riemann == APK_RS_HLLE) return launch_fused_mhd_hlle(u0, u1, a.cfg.recon, sp, extra, s);
    if (a.cfg.riemann == APK_RS_HLLD) return launch_fused_mhd_hlld(u0, u1, a.cfg.recon, sp, extra, s);
  }
  return APK_ERR_UNSUPPORTED;
}

}  // namespace apk
