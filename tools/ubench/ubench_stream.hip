// ubench_stream.hip -- does the width of a lane's access bound the streaming rate?  NS independent arrays ("variables")
// are read with 8 B per lane (global_load_dwordx2, what the fused kernels do: one double of one cell per lane) or with
// 16 B per lane (global_load_dwordx4), same bytes, same number of waves; plus a read + write (copy) pair.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_stream ubench_stream.hip && ./ubench_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int NS = 9;

// every wave walks `rows` consecutive 64-lane rows of each of the NS arrays (stride between arrays: elems)
template <int WIDTH>  // doubles per lane per load: 1 or 2
__global__ void __launch_bounds__(64) read_kernel(const double *__restrict__ a, double *out, long long elems, int rows) {
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * rows * 64 * WIDTH;
  double s = 0.0;
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int v = 0; v < NS; ++v) {
      const double *p = a + v * elems + base + (long long)r * 64 * WIDTH;
      if constexpr (WIDTH == 1) s += p[lane];
      else {
        const double2 q = reinterpret_cast<const double2 *>(p)[lane];
        s += q.x + q.y;
      }
    }
  }
  if (s == 12345.678) out[0] = s;
}

template <int WIDTH>
__global__ void __launch_bounds__(64) copy_kernel(const double *__restrict__ a, double *__restrict__ b, long long elems, int rows) {
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * rows * 64 * WIDTH;
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int v = 0; v < NS; ++v) {
      const long long off = v * elems + base + (long long)r * 64 * WIDTH;
      if constexpr (WIDTH == 1) b[off + lane] = a[off + lane] * 1.0000001;
      else {
        double2 q = reinterpret_cast<const double2 *>(a + off)[lane];
        q.x *= 1.0000001;
        q.y *= 1.0000001;
        reinterpret_cast<double2 *>(b + off)[lane] = q;
      }
    }
  }
}

int main() {
  const long long elems = 1ll << 25;  // 32 Mi doubles = 256 MiB per array, 9 arrays = 2.4 GB
  double *a, *b, *out;
  CHECK(hipMalloc(&a, sizeof(double) * elems * NS));
  CHECK(hipMalloc(&b, sizeof(double) * elems * NS));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(a, 0, sizeof(double) * elems * NS));
  CHECK(hipMemset(b, 0, sizeof(double) * elems * NS));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rows : {16, 128}) {
    for (int width : {1, 2}) {
      for (int kind : {0, 1}) {
        const long long per_wave = (long long)rows * 64 * width;
        const int blocks = (int)(elems / per_wave);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
          CHECK(hipEventRecord(e0, 0));
          if (kind == 0) {
            if (width == 1) hipLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(64), 0, 0, a, out, elems, rows);
            else hipLaunchKernelGGL(read_kernel<2>, dim3(blocks), dim3(64), 0, 0, a, out, elems, rows);
          } else {
            if (width == 1) hipLaunchKernelGGL(copy_kernel<1>, dim3(blocks), dim3(64), 0, 0, a, b, elems, rows);
            else hipLaunchKernelGGL(copy_kernel<2>, dim3(blocks), dim3(64), 0, 0, a, b, elems, rows);
          }
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          float ms;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = (double)elems * NS * 8.0 * (kind == 0 ? 1.0 : 2.0);
        std::printf("{\"kind\": \"%s\", \"bytes_per_lane\": %d, \"rows_per_wave\": %d, \"waves\": %d, \"ms\": %.3f, \"TB_per_s\": %.3f}\n",
                    kind == 0 ? "read 9 arrays" : "copy 9 arrays", 8 * width, rows, blocks, best, bytes / (best * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
