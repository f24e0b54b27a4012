// ubench_fp64.hip -- achievable fp64 VALU issue rate on one MI355X, the ceiling the fused sweeps are
// priced against.  Each kernel is a register-only loop of one instruction
// kind with NCHAIN independent dependency chains per lane; the grid fills every SIMD with `occ`
// waves.  Prints wave-instructions per second and the implied cycles per wave64 instruction at the
// effective clock measured with s_memtime... (no: the clock comes from rocprofv3 GRBM_GUI_ACTIVE;
// here we print rates only, per kind, so that ratios between kinds are box-independent).
//
//   hipcc --offload-arch=gfx950 -O3 -o ubench_fp64 ubench_fp64.hip && ./ubench_fp64
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                         \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      std::exit(1);                                                                      \
    }                                                                                    \
  } while (0)

constexpr int NCHAIN = 8;
constexpr int UNROLL = 32;  // instructions per chain per loop trip

enum Kind { K_FMA64 = 0, K_MUL64, K_ADD64, K_MAX64, K_CNDMASK, K_CMP64_CND, K_MOV32, K_RCP64, K_RSQ64, K_SQRT64, K_FMA32, K_DPP, K_COUNT };
static const char *kKindName[K_COUNT] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_cndmask_b32",
                                         "v_cmp_lt_f64+v_cndmask_b32", "v_mov_b32", "v_rcp_f64", "v_rsq_f64",
                                         "v_sqrt_f64", "v_fma_f32", "v_mov_b32 dpp wave_shr:1"};
// VALU instructions per "op" of each kind (the cmp+cndmask pair counts as two)
static const int kInstPerOp[K_COUNT] = {1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1};

template <int KIND>
__global__ void __launch_bounds__(256) loop_kernel(double *out, int trips, double seed) {
  double a[NCHAIN];
  float af[NCHAIN];
  int ai[NCHAIN];
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) {
    a[c] = seed + 1e-3 * (threadIdx.x + c);
    af[c] = (float)a[c];
    ai[c] = threadIdx.x + c;
  }
  const double b = 1.0 + 1e-9 * seed, d = 1e-9 * seed;
  const float bf = (float)b, df = (float)d;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int c = 0; c < NCHAIN; ++c) {
        if constexpr (KIND == K_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d));
        else if constexpr (KIND == K_MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
        else if constexpr (KIND == K_ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(d));
        else if constexpr (KIND == K_MAX64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[c]) : "v"(d));
        else if constexpr (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ai[c]) : "v"(ai[(c + 1) % NCHAIN]));
        else if constexpr (KIND == K_CMP64_CND)
          asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(ai[c]) : "v"(a[c]), "v"(b), "v"(ai[(c + 1) % NCHAIN]) : "vcc");
        else if constexpr (KIND == K_MOV32) asm volatile("v_mov_b32 %0, %1" : "+v"(ai[c]) : "v"(ai[(c + 1) % NCHAIN]));
        else if constexpr (KIND == K_RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[c]));
        else if constexpr (KIND == K_RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[c]));
        else if constexpr (KIND == K_SQRT64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a[c]));
        else if constexpr (KIND == K_FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(af[c]) : "v"(bf), "v"(df));
        else if constexpr (KIND == K_DPP) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ai[c]));
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < NCHAIN; ++c) s += a[c] + (double)af[c] + (double)ai[c];
  if (s == 12345.678) out[0] = s;  // never true: keeps the chains alive
}

template <int KIND>
double run(int occ, int trips, double *d_out) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * occ;  // 256-thread blocks: 4 waves, one per SIMD; occ blocks per CU = occ waves/SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(loop_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, trips, 1.0);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(loop_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, trips, 1.0);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double ops = (double)blocks * 4.0 * trips * UNROLL * NCHAIN;  // wave-level ops
  const double rate = ops * kInstPerOp[KIND] / (best * 1e-3);        // wave-instructions / s
  std::printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_inst_per_s\": %.4e, "
              "\"per_simd_per_s\": %.4e, \"cycles_per_inst_at_2.4GHz\": %.3f}\n",
              kKindName[KIND], occ, best, rate, rate / (cus * 4.0), 2.4e9 / (rate / (cus * 4.0)));
  return rate;
}

int main(int argc, char **argv) {
  const int trips = argc > 1 ? std::atoi(argv[1]) : 2000;
  double *d_out;
  CHECK(hipMalloc(&d_out, 64));
  for (int occ : {1, 2, 4, 8}) {
    run<K_FMA64>(occ, trips, d_out);
  }
  for (int occ : {2, 4}) {
    run<K_MUL64>(occ, trips, d_out);
    run<K_ADD64>(occ, trips, d_out);
    run<K_MAX64>(occ, trips, d_out);
    run<K_CNDMASK>(occ, trips, d_out);
    run<K_CMP64_CND>(occ, trips, d_out);
    run<K_MOV32>(occ, trips, d_out);
    run<K_DPP>(occ, trips, d_out);
    run<K_FMA32>(occ, trips, d_out);
    run<K_RCP64>(occ, trips / 4, d_out);
    run<K_RSQ64>(occ, trips / 4, d_out);
    run<K_SQRT64>(occ, trips / 4, d_out);
  }
  // a long fp64 FMA run for the clock measurement (rocprofv3 --pmc GRBM_GUI_ACTIVE on this binary)
  run<K_FMA64>(4, trips * 10, d_out);
  CHECK(hipFree(d_out));
  return 0;
}
