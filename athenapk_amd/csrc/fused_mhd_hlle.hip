// fused_mhd_hlle.hip -- fused stage sweeps for the (APK_FLUID_GLMMHD, APK_RS_HLLE) family.
#include "fused_kernel.hpp"

namespace apk {
int launch_fused_mhd_hlle(const PackView &u0, const PackView &u1, int recon,
                         const StageParams &sp, int extra, hipStream_t s) {
  return launch_fused_family<APK_FLUID_GLMMHD, APK_RS_HLLE>(u0, u1, recon, sp, extra, s);
}
}  // namespace apk
