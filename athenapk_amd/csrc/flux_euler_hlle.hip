// flux_euler_hlle.hip -- flux-array sweeps for the (APK_FLUID_EULER, APK_RS_HLLE) family, all
// reconstructions (registry of src/hydro/hydro.cpp:386-416).
#include "flux_kernel.hpp"

namespace apk {
int launch_fluxes_euler_hlle(const PackView &pv, int recon, double gamma, double c_h,
                          hipStream_t s, int faces, const int *face_list, int nlist, const FluxConsInput *from_cons) {
  return launch_flux_family<APK_FLUID_EULER, APK_RS_HLLE>(pv, recon, gamma, c_h, s, faces, face_list, nlist, from_cons);
}
}  // namespace apk
