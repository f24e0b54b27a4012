// ubench_lat.hip -- how many INDEPENDENT dependency chains per SIMD does it take to keep the fp64 pipe
// full?  Register-only loops of one instruction kind with NCHAIN chains per lane (1, 2, 4) at 1, 2, 3, 4
// waves per SIMD; plus the accuracy of v_rcp_f64 / v_rsq_f64 (how many Newton steps the product build's
// frcp / fsqrt need).   hipcc --offload-arch=gfx950 -O3 -o ubench_lat ubench_lat.hip && ./ubench_lat
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int UNROLL = 64;
enum Kind { K_FMA = 0, K_ADD, K_MUL, K_RCP, K_RSQ, K_RCP_NEWTON, K_CMPSEL, K_MINMAX };
static const char *kName[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_rsq_f64", "rcp+3newton(7 inst)", "cmp+2cndmask(3 inst)", "v_max_f64"};
static const int kInst[] = {1, 1, 1, 1, 1, 7, 3, 1};

template <int KIND, int NCHAIN>
__global__ void __launch_bounds__(256) chain_kernel(double *out, int trips, double seed) {
  double a[NCHAIN];
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) a[c] = seed + 1e-3 * (threadIdx.x + c);
  const double b = 1.0 + 1e-9 * seed, d = 1e-9 * seed;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int c = 0; c < NCHAIN; ++c) {
        if constexpr (KIND == K_FMA) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d));
        else if constexpr (KIND == K_ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(d));
        else if constexpr (KIND == K_MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
        else if constexpr (KIND == K_MINMAX) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[c]) : "v"(d));
        else if constexpr (KIND == K_RCP) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[c]));
        else if constexpr (KIND == K_RSQ) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[c]));
        else if constexpr (KIND == K_RCP_NEWTON) {
          double x = a[c], y, e;
          asm volatile("v_rcp_f64 %0, %1" : "=v"(y) : "v"(x));
          asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(x), "v"(y));
          asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(y) : "v"(e));
          asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(x), "v"(y));
          asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(y) : "v"(e));
          asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(x), "v"(y));
          asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(y) : "v"(e));
          a[c] = y;
        } else if constexpr (KIND == K_CMPSEL) {
          int lo = __double2loint(a[c]), hi = __double2hiint(a[c]);
          asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %5, vcc"
                       : "+v"(lo), "+v"(hi) : "v"(a[c]), "v"(b), "v"(__double2loint(b)), "v"(__double2hiint(b)) : "vcc");
          a[c] = __hiloint2double(hi, lo);
        }
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) s += a[c];
  if (s == 12345.678) out[0] = s;
}

template <int KIND, int NCHAIN>
void run(int occ, int trips, double *d_out, int cus) {
  // 64-thread blocks so that any number of waves per SIMD can be asked for: occ * 4 blocks per CU
  const int blocks = cus * 4 * occ;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((chain_kernel<KIND, NCHAIN>), dim3(blocks), dim3(64), 0, 0, d_out, trips, 1.0);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((chain_kernel<KIND, NCHAIN>), dim3(blocks), dim3(64), 0, 0, d_out, trips, 1.0);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double inst = (double)blocks * trips * UNROLL * NCHAIN * kInst[KIND];
  const double per_simd = inst / (best * 1e-3) / (cus * 4.0);
  // per-wave time between dependent instructions of ONE chain, in ns
  const double ns_per_link = best * 1e6 / ((double)trips * UNROLL * kInst[KIND]);
  std::printf("{\"kind\": \"%s\", \"chains_per_lane\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"inst_per_simd_per_s\": %.4e, "
              "\"ns_per_dependent_link\": %.3f}\n", kName[KIND], NCHAIN, occ, best, per_simd, ns_per_link);
}

__global__ void accuracy_kernel(const double *x, double *rcp, double *rsq, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    rcp[i] = __builtin_amdgcn_rcp(x[i]);
    rsq[i] = __builtin_amdgcn_rsq(x[i]);
  }
}

int main(int argc, char **argv) {
  const int trips = argc > 1 ? std::atoi(argv[1]) : 400;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double *d_out;
  CHECK(hipMalloc(&d_out, 64));
  for (int occ : {1, 2, 3, 4}) {
    run<K_FMA, 1>(occ, trips, d_out, cus);
    run<K_FMA, 2>(occ, trips, d_out, cus);
    run<K_FMA, 4>(occ, trips, d_out, cus);
  }
  for (int occ : {1, 2}) {
    run<K_ADD, 1>(occ, trips, d_out, cus);
    run<K_MUL, 1>(occ, trips, d_out, cus);
    run<K_MINMAX, 1>(occ, trips, d_out, cus);
    run<K_CMPSEL, 1>(occ, trips, d_out, cus);
    run<K_CMPSEL, 2>(occ, trips, d_out, cus);
    run<K_RCP, 1>(occ, trips / 4, d_out, cus);
    run<K_RCP, 2>(occ, trips / 4, d_out, cus);
    run<K_RSQ, 1>(occ, trips / 4, d_out, cus);
    run<K_RCP_NEWTON, 1>(occ, trips / 4, d_out, cus);
    run<K_RCP_NEWTON, 2>(occ, trips / 4, d_out, cus);
  }
  // accuracy of the raw transcendental results
  const int n = 1 << 20;
  std::vector<double> h(n), hr(n), hs(n);
  unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    const double u = (st >> 11) * (1.0 / 9007199254740992.0);
    h[i] = std::exp((u * 2.0 - 1.0) * 27.6);  // 1e-12 .. 1e12
  }
  double *dx, *dr, *ds;
  CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&dr, n * 8)); CHECK(hipMalloc(&ds, n * 8));
  CHECK(hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(accuracy_kernel, dim3(n / 256), dim3(256), 0, 0, dx, dr, ds, n);
  CHECK(hipMemcpy(hr.data(), dr, n * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hs.data(), ds, n * 8, hipMemcpyDeviceToHost));
  double er = 0.0, es = 0.0;
  for (int i = 0; i < n; ++i) {
    er = std::fmax(er, std::fabs(hr[i] * h[i] - 1.0));
    es = std::fmax(es, std::fabs(hs[i] * std::sqrt(h[i]) - 1.0));
  }
  std::printf("{\"v_rcp_f64_max_rel_err\": %.3e, \"bits\": %.1f, \"v_rsq_f64_max_rel_err\": %.3e, \"rsq_bits\": %.1f}\n", er, -std::log2(er), es,
              -std::log2(es));
  return 0;
}
