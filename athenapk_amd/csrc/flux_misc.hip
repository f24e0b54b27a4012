// flux_misc.hip -- the remaining registry entries: (dc, none) resets the fluxes
// (src/hydro/rsolvers/rsolvers.hpp:35-63) and (dc, llf) is CalculateFluxesTight
// (src/hydro/hydro.cpp:980-1022), for both fluids.
#include "flux_kernel.hpp"

namespace apk {
int launch_fluxes_misc(const PackView &pv, int fluid, int riemann, double gamma, double c_h,
                       hipStream_t s) {
  if (fluid == APK_FLUID_EULER) {
    if (riemann == APK_RS_NONE)
      return launch_flux_all_dirs<APK_FLUID_EULER, APK_RC_DC, APK_RS_NONE>(pv, gamma, c_h, s);
    if (riemann == APK_RS_LLF)
      return launch_flux_all_dirs<APK_FLUID_EULER, APK_RC_DC, APK_RS_LLF>(pv, gamma, c_h, s, FLUX_FACES_TIGHT);
  } else if (fluid == APK_FLUID_GLMMHD) {
    if (riemann == APK_RS_NONE)
      return launch_flux_all_dirs<APK_FLUID_GLMMHD, APK_RC_DC, APK_RS_NONE>(pv, gamma, c_h, s);
    if (riemann == APK_RS_LLF)
      return launch_flux_all_dirs<APK_FLUID_GLMMHD, APK_RC_DC, APK_RS_LLF>(pv, gamma, c_h, s, FLUX_FACES_TIGHT);
  }
  return APK_ERR_UNSUPPORTED;
}
}  // namespace apk
