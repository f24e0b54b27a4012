// accuracy of the product build's fsqrt / fsqrt_rsqrt / frcp / fast_speed against IEEE evaluation on the host
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "../../athenapk_amd/csrc/hydro_math.hpp"
using namespace apk;
__global__ void k(const double *x, const double *y, double *o, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  double r, ir;
  fsqrt_rsqrt(x[t], r, ir);
  o[t] = fsqrt(x[t]);
  o[n + t] = r;
  o[2 * n + t] = ir;
  o[3 * n + t] = frcp(x[t]);
  o[4 * n + t] = fast_speed(5.0 / 3.0, x[t], y[t], 0.3 * y[t], 0.7 * x[t], 0.2);
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), y(n), o(5 * n);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> u(-6.0, 6.0);
  for (int i = 0; i < n; ++i) { x[i] = std::pow(10.0, u(g)); y[i] = std::pow(10.0, u(g)); }
  double *dx, *dy, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dout, 5 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, dout, n);
  hipMemcpy(o.data(), dout, 5 * n * 8, hipMemcpyDeviceToHost);
  double e[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double s = std::sqrt(x[i]);
    e[0] = std::fmax(e[0], std::fabs(o[i] - s) / s);
    e[1] = std::fmax(e[1], std::fabs(o[n + i] - s) / s);
    e[2] = std::fmax(e[2], std::fabs(o[2 * n + i] - 1.0 / s) * s);
    e[3] = std::fmax(e[3], std::fabs(o[3 * n + i] - 1.0 / x[i]) * x[i]);
    const double gam = 5.0 / 3.0, d = x[i], p = y[i], bx = 0.3 * y[i], by = 0.7 * x[i], bz = 0.2;
    const double asq = gam * p, ct2 = by * by + bz * bz, qsq = bx * bx + ct2 + asq, tmp = bx * bx + ct2 - asq;
    const double cf = std::sqrt(0.5 * (qsq + std::sqrt(tmp * tmp + 4.0 * asq * ct2)) / d);
    e[4] = std::fmax(e[4], std::fabs(o[4 * n + i] - cf) / cf);
  }
  std::printf("max rel err: fsqrt %.2e  fsqrt_rsqrt.root %.2e  .inv_root %.2e  frcp %.2e  fast_speed %.2e\n", e[0], e[1], e[2], e[3], e[4]);
  return 0;
}
