// fused_mhd_hlld.hip -- fused stage sweeps for the (APK_FLUID_GLMMHD, APK_RS_HLLD) family.
#include "fused_kernel.hpp"

namespace apk {
int launch_fused_mhd_hlld(const PackView &u0, const PackView &u1, int recon,
                         const StageParams &sp, int extra, hipStream_t s) {
  return launch_fused_family<APK_FLUID_GLMMHD, APK_RS_HLLD>(u0, u1, recon, sp, extra, s);
}
}  // namespace apk
