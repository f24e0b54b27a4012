// hlld_resources.hip -- can the finishing march run three waves per SIMD (<= 168 VGPRs)?  (round-3 review, item 1, stage (i))
// The HLLD solve by itself, and the solve with L doubles live through it (what a march carries across its x2 solve: the
// prefetched row, the previous face's flux, the PPM interface values, the x1 flux differences, the next L state = 45
// doubles), each compiled for three waves per SIMD:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iathenapk_amd/csrc -fapprox-func -freciprocal-math -c \
//         tools/ubench/hlld_resources.hip -o /dev/null -Rpass-analysis=kernel-resource-usage
// Result (profiles/r04_three_waves_per_simd.txt): the solve alone needs 116 VGPRs -- not the ~170 the scheduler spreads
// it over when it has 256 --, so it fits; with 18 / 27 / 36 / 45 doubles carried through it the kernel spills 24 / 46 /
// 74 / 114 registers to scratch at the 168-register limit.  The march carries 45, and its LDS (160 KB / 12 waves =
// 13.3 KB) is taken by the stencil ring (3 slots x 9 variables x 512 B = 13.5 KB), so the carried state has nowhere to go.
#include "fused_kernel.hpp"
using namespace apk;
// HLLD alone: 18 doubles in, 9 out, nothing else live
template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) hlld_alone(const double *in, double *out, StageConsts k, long long sn) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  double wl[9], wr[9], f[9];
#pragma unroll
  for (int n = 0; n < 9; ++n) { wl[n] = in[n * sn + t]; wr[n] = in[(9 + n) * sn + t]; }
  glmmhd_hlld(wl, wr, k, f);
#pragma unroll
  for (int n = 0; n < 9; ++n) out[n * sn + t] = f[n];
}
// HLLD with L live-through doubles (the march's carried state across the x2 solve)
template <int WAVES, int L>
__global__ void __launch_bounds__(64, WAVES) hlld_carry(const double *in, double *out, StageConsts k, long long sn, int iters) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  double carry[L];
#pragma unroll
  for (int n = 0; n < L; ++n) carry[n] = in[(18 + n) * sn + t];
  for (int it = 0; it < iters; ++it) {
    double wl[9], wr[9], f[9];
#pragma unroll
    for (int n = 0; n < 9; ++n) { wl[n] = in[n * sn + t + it]; wr[n] = in[(9 + n) * sn + t + it]; }
    glmmhd_hlld(wl, wr, k, f);
#pragma unroll
    for (int n = 0; n < L; ++n) carry[n] = carry[n] * 0.999 + f[n % 9];
  }
#pragma unroll
  for (int n = 0; n < L; ++n) out[n * sn + t] = carry[n];
}
template __global__ void hlld_alone<2>(const double *, double *, StageConsts, long long);
template __global__ void hlld_alone<3>(const double *, double *, StageConsts, long long);
template __global__ void hlld_alone<4>(const double *, double *, StageConsts, long long);
template __global__ void hlld_carry<3, 18>(const double *, double *, StageConsts, long long, int);
template __global__ void hlld_carry<3, 27>(const double *, double *, StageConsts, long long, int);
template __global__ void hlld_carry<3, 36>(const double *, double *, StageConsts, long long, int);
template __global__ void hlld_carry<3, 45>(const double *, double *, StageConsts, long long, int);
