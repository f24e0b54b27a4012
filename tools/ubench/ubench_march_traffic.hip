// ubench_march_traffic.hip -- what does the memory system deliver for the finishing march's ACCESS PATTERN with no arithmetic?
// 2048 waves (2 per SIMD, forced by an LDS allocation like the march's ring), each marching along j over 64 consecutive
// (k, i)-flattened columns of a 134^3 block: per row 9 + 9 + 9 loads of 512 B (prim, d3, u1) and 9 + 9 stores (u0, prim'),
// i.e. the finishing march's 45 streams per block.  Variants: fewer stores, and a row-interleaved layout [k][j][var][i]
// (one wave-row's nine variables contiguous) instead of [var][k][j][i].
//   hipcc --offload-arch=gfx950 -O3 -o ubench_march_traffic ubench_march_traffic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int NV = 9, N = 134, NI = 128;

template <int NLOAD_ARR, int NSTORE_ARR, bool INTERLEAVED, int MODE = 0, int SCHED = 0>
__global__ void __launch_bounds__(64, 2) march(const double *in, double *out, long long sn_block, int nwaves, long long total_rows, int wpb) {
  extern __shared__ double pad[];  // 18 KB: two waves per SIMD, as the march
  const int lane = threadIdx.x;
  const int w = (int)(blockIdx.x % 8u) * ((nwaves + 7) / 8) + (int)(blockIdx.x / 8u);
  if (w >= nwaves) return;
  // MODE 0: 58-cell chunks of the (k, i)-flattened run, every lane stores; 1: the same, lanes 3..60 store (the march);
  // 2: 64-cell chunks of interior cells only (two per row), row pitch 134 (interior starts 24 B into a line);
  // 3: the same with row pitch 160 and the interior starting on a 128-byte boundary
  constexpr int P = (MODE == 3) ? 160 : N, I0 = (MODE == 3) ? 16 : 3;
  const long long sj = INTERLEAVED ? (long long)NV * P : P, sk = sj * N, sn = INTERLEAVED ? P : (long long)P * N * N;
  long long r = total_rows * w / nwaves;
  long long r_end = total_rows * (w + 1) / nwaves;
  double acc = 0.0;
  // SCHED 1: wave w takes column w whole (all waves at the same row at the same time: the two waves that share a
  // partially written cache line write it within one iteration of each other), then a share of the leftover columns
  const int ncol = (int)(total_rows / NI);
  int piece = 0;
  const int left_cols = ncol - nwaves, seg = left_cols > 0 ? (left_cols * NI + nwaves - 1) / nwaves : 0;
  while (SCHED == 1 ? piece < 2 : r < r_end) {
    int item, j0, nrows;
    if (SCHED == 1) {
      if (piece == 0) {
        item = w, j0 = 0, nrows = NI;
      } else {
        if (left_cols <= 0) break;
        const int per_col = (NI + seg - 1) / seg;       // segments per leftover column
        const int sidx = w / left_cols, lc = w - sidx * left_cols;  // adjacent waves: adjacent columns, same rows
        if (sidx >= per_col) break;
        item = nwaves + lc, j0 = sidx * seg, nrows = (j0 + seg <= NI) ? seg : NI - j0;
      }
      ++piece;
    } else {
      item = (int)(r / NI);
      j0 = (int)(r - (long long)item * NI);
      const long long left = r_end - r;
      nrows = left < NI - j0 ? (int)left : NI - j0;
      r += nrows;
    }
    const int b = item / wpb, chunk = item - b * wpb;
    int k, i;
    if (MODE >= 2) {
      const long long t = (long long)chunk * 64 + lane;
      k = (int)(t / NI);
      i = I0 + (int)(t - (long long)k * NI);
    } else {
      long long t = (long long)chunk * 58 + lane - 3;
      if (t < 0) t = 0;
      if (t >= (long long)NI * N) t = (long long)NI * N - 1;
      k = (int)(t / N), i = (int)(t - (long long)k * N);
    }
    const bool st_lane = (MODE != 1) || (lane >= 3 && lane <= 60);
    const long long base = (long long)(3 + k) * sk + i;
    for (int j = 3 + j0; j < 3 + j0 + nrows; ++j) {
      double v[NLOAD_ARR][NV];
#pragma unroll
      for (int a = 0; a < NLOAD_ARR; ++a)
#pragma unroll
        for (int n = 0; n < NV; ++n) v[a][n] = in[((long long)a * 8 + b) * sn_block + n * sn + base + j * sj];
#pragma unroll
      for (int s = 0; s < NSTORE_ARR; ++s)
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          double x = v[0][n];
#pragma unroll
          for (int a = 1; a < NLOAD_ARR; ++a) x += v[a][n];
          if (st_lane) out[((long long)s * 8 + b) * sn_block + n * sn + base + j * sj] = x + s;
        }
      if (NSTORE_ARR == 0) {
#pragma unroll
        for (int a = 0; a < NLOAD_ARR; ++a)
#pragma unroll
          for (int n = 0; n < NV; ++n) acc += v[a][n];
      }
    }
  }
  if (acc == 12345.678) pad[lane] = acc, out[0] = pad[lane];
}

template <int L, int S, bool I, int MODE = 0, int SCHED = 0>
static void run(const char *name, const double *in, double *out, long long snb) {
  const int wpb = (MODE >= 2) ? NI * NI / 64 : (NI * N + 57) / 58, nwaves = 2048;
  const long long total_rows = 8LL * wpb * NI;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((march<L, S, I, MODE, SCHED>), dim3(((nwaves + 7) / 8) * 8), dim3(64), 18432, 0, in, out, snb, nwaves, total_rows, wpb);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double useful = 8.0 * NI * NI * NI * NV * 8.0 * (L + S);  // bytes of interior cells
  std::printf("{\"variant\": \"%s\", \"load_arrays\": %d, \"store_arrays\": %d, \"layout\": \"%s\", \"ms\": %.4f, \"interior_GB\": %.3f, \"TBps_interior\": %.3f}\n",
              name, L, S, I ? "[k][j][var][i]" : "[var][k][j][i]", best, useful / 1e9, useful / best / 1e9);
}

int main() {
  const long long snb = (long long)NV * 160 * N * N;
  double *in, *out;
  CHECK(hipMalloc(&in, sizeof(double) * snb * 8 * 3));
  CHECK(hipMalloc(&out, sizeof(double) * snb * 8 * 2));
  CHECK(hipMemset(in, 0, sizeof(double) * snb * 8 * 3));
  CHECK(hipMemset(out, 0, sizeof(double) * snb * 8 * 2));
  run<3, 2, false>("march: 27 load + 18 store streams", in, out, snb);
  run<3, 1, false>("27 load + 9 store", in, out, snb);
  run<3, 0, false>("27 load", in, out, snb);
  run<2, 1, false>("18 load + 9 store", in, out, snb);
  run<1, 1, false>("9 load + 9 store", in, out, snb);
  run<3, 2, false, 1>("march, lanes 3..60 store", in, out, snb);
  run<1, 1, false, 1>("9 + 9, lanes 3..60 store", in, out, snb);
  run<3, 2, false, 1, 1>("march, lanes 3..60 store, one column per wave in lockstep", in, out, snb);
  run<3, 1, false, 1, 0>("27 + 9, lanes 3..60 store, equal split", in, out, snb);
  run<3, 1, false, 1, 1>("27 + 9, lanes 3..60 store, lockstep", in, out, snb);
  run<3, 2, false, 2>("64-cell interior chunks, pitch 134", in, out, snb);
  run<1, 1, false, 2>("9 + 9, 64-cell interior chunks, pitch 134", in, out, snb);
  run<3, 2, false, 3>("64-cell interior chunks, pitch 160 aligned", in, out, snb);
  run<1, 1, false, 3>("9 + 9, 64-cell chunks, pitch 160 aligned", in, out, snb);
  run<3, 0, false, 3>("27 load, 64-cell chunks aligned", in, out, snb);
  run<3, 2, true>("march, interleaved rows", in, out, snb);
  run<3, 1, true>("27 + 9, interleaved rows", in, out, snb);
  run<3, 0, true>("27 load, interleaved rows", in, out, snb);
  run<1, 1, true>("9 + 9, interleaved rows", in, out, snb);
  return 0;
}
