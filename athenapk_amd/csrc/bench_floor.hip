// bench_floor.hip -- the empirical issue floor of the PPM + HLLD scheme (bench.py: roofline.general_stage.scheme_floor).
//
// What a GLM-MHD PPM+HLLD cell-stage costs when NOTHING but the scheme's own arithmetic is issued: per sweep direction and
// cell the nine PPM reconstructions (ppm_interface + ppm_cell of hydro_math.hpp, the very functions the stage kernels
// call) and one HLLD solve, at the stage kernels' occupancy of two waves per SIMD -- no global memory, no wave shifts, no
// update / Dedner source / ConsToPrim, and a state so smooth and monotone that no lane ever enters a limiter block.  The
// five-row stencil sits in a private LDS ring, as in the marches (a register-resident stencil would spill next to an HLLD
// solve); one cell-stage = three such sweep steps.  A stage kernel cannot run faster than this at its occupancy: what
// separates a measured stage from it is the marches' own work (loads, stores, wave shifts, limiter blocks, the finish).
#include "hydro_math.hpp"

namespace apk {
namespace {

__global__ void __launch_bounds__(64, 2) scheme_floor_kernel(double *out, int steps, StageConsts k, double seed) {
  constexpr int NV = NGLMMHD, NS = 4;
  __shared__ __attribute__((aligned(16))) double ring[NS * NV * 64];
  const int lane = threadIdx.x;
  // a smooth, monotone pencil per lane: row r of the stencil is base + r * slope (+ a curvature too small for an extremum)
  double base[NV], slope[NV];
  const double ph = seed + 1.0e-3 * lane + 1.0e-5 * blockIdx.x;
  base[IDN] = 1.0 + 0.1 * ph, slope[IDN] = 1.0e-3;
  base[IV1] = 0.3, slope[IV1] = 2.0e-3;
  base[IV2] = -0.2, slope[IV2] = 1.0e-3;
  base[IV3] = 0.1, slope[IV3] = -1.5e-3;
  base[IPR] = 1.0 + 0.05 * ph, slope[IPR] = 1.2e-3;
  base[IB1] = 0.5, slope[IB1] = 1.0e-3;
  base[IB2] = 0.4, slope[IB2] = -1.0e-3;
  base[IB3] = -0.3, slope[IB3] = 0.8e-3;
  base[IPS] = 0.01, slope[IPS] = 1.0e-4;
#pragma unroll
  for (int m = 0; m < NS; ++m)
#pragma unroll
    for (int n = 0; n < NV; ++n) ring[(m * NV + n) * 64 + lane] = base[n] + (m + 0.01 * m * m) * slope[n];
  double newest[NV], face_carry[NV], wl_prev[NV], acc[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    newest[n] = base[n] + (NS + 0.01 * NS * NS) * slope[n];
    face_carry[n] = ppm_interface(ring[(0 * NV + n) * 64 + lane], ring[(1 * NV + n) * 64 + lane], ring[(2 * NV + n) * 64 + lane],
                                  ring[(3 * NV + n) * 64 + lane]);
    wl_prev[n] = ring[(1 * NV + n) * 64 + lane];
    acc[n] = 0.0;
  }
  int slot0 = 0;
  for (int step = 0; step < steps; ++step) {
    double ql[NV], qr[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const double a0 = ring[(((slot0 + 0) & (NS - 1)) * NV + n) * 64 + lane], a1 = ring[(((slot0 + 1) & (NS - 1)) * NV + n) * 64 + lane],
                   a2 = ring[(((slot0 + 2) & (NS - 1)) * NV + n) * 64 + lane], a3 = ring[(((slot0 + 3) & (NS - 1)) * NV + n) * 64 + lane];
      const double face_p = ppm_interface(a1, a2, a3, newest[n]);
      ppm_cell(a0, a1, a2, a3, newest[n], face_carry[n], face_p, ql[n], qr[n]);
      face_carry[n] = face_p;
    }
    double f[NV];
    riemann<APK_FLUID_GLMMHD, APK_RS_HLLD>(wl_prev, qr, k, f);
    // the ring moves on by one row; the row that enters continues the pencil and depends on the flux just computed (so that
    // no step can be hoisted or shared), by an amount too small to bend it
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      wl_prev[n] = ql[n];
      acc[n] += f[n];
      ring[(slot0 * NV + n) * 64 + lane] = newest[n];
      newest[n] = fma(1.0e-13, f[n], newest[n] + slope[n] * (1.0 + 0.02 * (step & 7)));
    }
    slot0 = (slot0 + 1) & (NS - 1);
  }
  double sum = 0.0;
#pragma unroll
  for (int n = 0; n < NV; ++n) sum += acc[n] + wl_prev[n];
  if (sum == 1.2345e300) out[blockIdx.x] = sum;  // (keeps the loop alive; never true)
}

}  // namespace
}  // namespace apk

extern "C" int apk_bench_scheme_floor(int steps, int reps, double *ms_per_launch, long long *lane_steps_per_launch) {
  if (steps < 1 || reps < 1 || !ms_per_launch || !lane_steps_per_launch) return APK_ERR_INVALID;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return APK_ERR_NO_DEVICE;
  cus = prop.multiProcessorCount;
  const int waves = cus * 4 * 2;  // two per SIMD: the stage kernels' occupancy
  double *out = nullptr;
  if (hipMalloc(&out, sizeof(double) * waves) != hipSuccess) return APK_ERR_DEVICE;
  apk_eos eos{};
  eos.gamma = 5.0 / 3.0;
  eos.vceil = eos.eceil = __builtin_inf();
  const apk::StageConsts k = apk::make_stage_consts(5.0 / 3.0, 2.0, eos);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(apk::scheme_floor_kernel, dim3(waves), dim3(64), 0, nullptr, out, steps, k, 0.5);  // warm-up
  (void)hipEventRecord(e0, nullptr);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(apk::scheme_floor_kernel, dim3(waves), dim3(64), 0, nullptr, out, steps, k, 0.5 + r);
  (void)hipEventRecord(e1, nullptr);
  const bool ok = hipEventSynchronize(e1) == hipSuccess && hipGetLastError() == hipSuccess;
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  if (!ok) return APK_ERR_DEVICE;
  *ms_per_launch = (double)ms / reps;
  *lane_steps_per_launch = (long long)waves * 64 * steps;
  return APK_OK;
}
