// fused_dispatch.hip -- entry of the fused stage path: owns the du scratch array and picks
// the (fluid, riemann) family.
#include "fused_kernel.hpp"

namespace apk {

namespace {
double *g_du = nullptr;      // per-process scratch, grown on demand (never shrinks)
size_t g_du_cap = 0;         // in doubles
int g_du_device = -1;
}  // namespace

int launch_stage_fused(const PackView &u0, const PackView &u1, const apk_stage_args &a,
                       double dedner_coeff, hipStream_t s) {
  if (u0.nvar != u0.nhydro) return APK_ERR_UNSUPPORTED;  // passive scalars: flux-array path
  StageParams sp;
  sp.gamma = a.eos.gamma;
  sp.c_h = a.c_h;
  sp.gam0 = a.gam0;
  sp.gam1 = a.gam1;
  sp.beta_dt = a.beta_dt;
  sp.dedner = a.dedner;
  sp.dedner_coeff = dedner_coeff;
  sp.du = nullptr;
  if (u0.ndim > 1) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return APK_ERR_DEVICE;
    const size_t need = (size_t)u0.nblocks * (size_t)u0.nvar * (size_t)u0.sn;
    if (need > g_du_cap || dev != g_du_device) {
      if (g_du) {
        (void)hipDeviceSynchronize();
        (void)hipFree(g_du);
      }
      g_du = nullptr;
      g_du_cap = 0;
      if (hipMalloc(&g_du, need * sizeof(double)) != hipSuccess) return APK_ERR_DEVICE;
      g_du_cap = need;
      g_du_device = dev;
    }
    sp.du = g_du;
  }
  if (a.cfg.fluid == APK_FLUID_EULER) {
    if (a.cfg.riemann == APK_RS_HLLE) return launch_fused_euler_hlle(u0, u1, a.cfg.recon, sp, s);
    if (a.cfg.riemann == APK_RS_HLLC) return launch_fused_euler_hllc(u0, u1, a.cfg.recon, sp, s);
  } else if (a.cfg.fluid == APK_FLUID_GLMMHD) {
    if (a.cfg.riemann == APK_RS_HLLE) return launch_fused_mhd_hlle(u0, u1, a.cfg.recon, sp, s);
    if (a.cfg.riemann == APK_RS_HLLD) return launch_fused_mhd_hlld(u0, u1, a.cfg.recon, sp, s);
  }
  return APK_ERR_UNSUPPORTED;
}

}  // namespace apk
