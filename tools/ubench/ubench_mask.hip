// ubench_mask.hip -- what does a wave64 fp64 instruction cost when most of its lanes are switched off?
// A dependent v_fma_f64 loop (4 chains per lane, 2 waves per SIMD) under exec masks with 64, 32 (lower half), 16 (one
// row), 2 and 1 live lanes, and with the live lanes spread one per row.    hipcc --offload-arch=gfx950 -O3 -o ubench_mask ubench_mask.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void __launch_bounds__(64) masked_kernel(double *out, int trips, double seed, unsigned long long mask) {
  double a[4];
  for (int c = 0; c < 4; ++c) a[c] = seed + 1e-3 * (threadIdx.x + c);
  const double b = 1.0 + 1e-9 * seed, d = 1e-9 * seed;
  if ((mask >> threadIdx.x) & 1ull) {   // the loop runs with exec = mask
    for (int t = 0; t < trips; ++t) {
#pragma unroll
      for (int u = 0; u < 64; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d));
      }
    }
  }
  double s = a[0] + a[1] + a[2] + a[3];
  if (s == 12345.678) out[0] = s;
}

// 256-thread blocks (one wave per SIMD); blocks alternate between two exec masks, so with two blocks per CU every SIMD
// holds one wave of each kind: do few-lane instructions hold up the full-width wave next to them?
template <int OP>
__global__ void __launch_bounds__(256) mixed_kernel(double *out, int trips_even, int trips_odd, double seed, unsigned long long mask_even,
                                                    unsigned long long mask_odd) {
  double a[4];
  for (int c = 0; c < 4; ++c) a[c] = seed + 1e-3 * (threadIdx.x + c);
  const double b = 1.0 + 1e-9 * seed, d = 1e-9 * seed;
  const unsigned long long mask = (blockIdx.x & 1) ? mask_odd : mask_even;
  const int trips = (blockIdx.x & 1) ? trips_odd : trips_even;
  if ((mask >> (threadIdx.x & 63)) & 1ull) {
    for (int t = 0; t < trips; ++t) {
#pragma unroll
      for (int u = 0; u < 64; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if constexpr (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d));
          else if constexpr (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(d));
          else if constexpr (OP == 2) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[c]) : "v"(d));
          else asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
        }
      }
    }
  }
  double s = a[0] + a[1] + a[2] + a[3];
  if (s == 12345.678) out[0] = s;
}

template <int OP>
static float time_mixed(double *d_out, int cus, int te, int to, unsigned long long me, unsigned long long mo) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int blocks = cus * 2;
  hipLaunchKernelGGL(mixed_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, te, to, 1.0, me, mo);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mixed_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, te, to, 1.0, me, mo);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, trips = 400;
  double *d_out;
  CHECK(hipMalloc(&d_out, 64));
  struct { const char *name; unsigned long long mask; } cases[] = {
      {"64 lanes", ~0ull}, {"32 lanes (lower half)", 0xffffffffull}, {"16 lanes (row 0)", 0xffffull}, {"16 lanes (row 3)", 0xffffull << 48},
      {"2 lanes (0, 1)", 3ull}, {"1 lane (0)", 1ull}, {"1 lane (37)", 1ull << 37}, {"4 lanes, one per row", (1ull) | (1ull << 16) | (1ull << 32) | (1ull << 48)},
      {"2 lanes (rows 0 and 3)", 1ull | (1ull << 63)}};
  for (int occ : {1, 2}) {
    for (auto &cs : cases) {
      const int blocks = cus * 4 * occ;
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      hipLaunchKernelGGL(masked_kernel, dim3(blocks), dim3(64), 0, 0, d_out, trips, 1.0, cs.mask);
      CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(masked_kernel, dim3(blocks), dim3(64), 0, 0, d_out, trips, 1.0, cs.mask);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double inst = (double)trips * 64 * 4 * occ;  // wave-instructions per SIMD
      std::printf("{\"waves_per_simd\": %d, \"exec\": \"%s\", \"ms\": %.4f, \"ns_per_wave_inst_per_simd\": %.3f}\n", occ, cs.name, best,
                  best * 1e6 / inst);
    }
  }
  // lanes-live sweep (one wave per SIMD)
  for (int nl : {1, 2, 4, 5, 8, 12, 15, 16, 17, 24, 32, 48, 64}) {
    const unsigned long long m = nl == 64 ? ~0ull : ((1ull << nl) - 1ull);
    const float ms = time_mixed<0>(d_out, cus, 400, 0, m, 0ull);
    std::printf("{\"sweep\": \"v_fma_f64, lanes 0..n-1 live, one wave per SIMD\", \"lanes\": %d, \"ns_per_wave_inst\": %.3f}\n", nl, ms * 1e6 / (400.0 * 64 * 4));
  }
  const char *opn[4] = {"v_fma_f64", "v_add_f64", "v_max_f64", "v_mul_f64"};
  float t1[4], t64[4];
  t1[0] = time_mixed<0>(d_out, cus, 400, 0, 1ull, 0ull); t64[0] = time_mixed<0>(d_out, cus, 400, 0, ~0ull, 0ull);
  t1[1] = time_mixed<1>(d_out, cus, 400, 0, 1ull, 0ull); t64[1] = time_mixed<1>(d_out, cus, 400, 0, ~0ull, 0ull);
  t1[2] = time_mixed<2>(d_out, cus, 400, 0, 1ull, 0ull); t64[2] = time_mixed<2>(d_out, cus, 400, 0, ~0ull, 0ull);
  t1[3] = time_mixed<3>(d_out, cus, 400, 0, 1ull, 0ull); t64[3] = time_mixed<3>(d_out, cus, 400, 0, ~0ull, 0ull);
  for (int o = 0; o < 4; ++o)
    std::printf("{\"op\": \"%s\", \"ns_per_wave_inst_1_lane\": %.3f, \"ns_per_wave_inst_64_lanes\": %.3f}\n", opn[o], t1[o] * 1e6 / (400.0 * 64 * 4),
                t64[o] * 1e6 / (400.0 * 64 * 4));
  // a full-width wave and a one-lane wave on the same SIMD: alone / alone / together (same trips each)
  const float full = time_mixed<0>(d_out, cus, 400, 0, ~0ull, 0ull), one = time_mixed<0>(d_out, cus, 0, 100, 0ull, 1ull);
  const float both = time_mixed<0>(d_out, cus, 400, 100, ~0ull, 1ull);
  std::printf("{\"mixed\": \"full-width wave (400 trips) and one-lane wave (100 trips) per SIMD\", \"full_alone_ms\": %.4f, \"one_lane_alone_ms\": %.4f, \"together_ms\": %.4f}\n",
              full, one, both);
  return 0;
}
