// kernels_turb.hip -- device side of the few-modes turbulence driver (BASELINE config 4):
// explicit inverse transform of <= ~100 modes per cell and the Perturb kernels.
// The inverse transform is ALU-bound (3 x num_modes complex MACs per cell): one lane per cell,
// x1 along the lanes so phases_i is a coalesced read, phases_j / phases_k and var_hat are
// wave-uniform (scalar loads).
#include <cstdlib>
#include <cstring>
#include <new>

#include "apk_internal.hpp"
#include "hydro_math.hpp"

struct apk_fmft {
  int nblocks = 0, num_modes = 0;
  apk_fmft_block *d_blocks = nullptr;
  double *d_var_hat = nullptr;  // [3][M][2]
};

namespace apk {
namespace {

APK_DEV bool interior_of(const PackView &pv, int &b, int &k, int &j, int &i) {
  int io, jo;
  const bool inside = rect_ij(pv.nx1, pv.nx2, io, jo);
  i = pv.is + io;
  j = pv.js + jo;
  b = blockIdx.z / pv.nx3;
  k = pv.ks + blockIdx.z % pv.nx3;
  return inside;
}
inline dim3 igrid(const PackView &pv) { return rect_grid(pv.nx1, pv.nx2, pv.nx3 * pv.nblocks); }

#ifndef APK_FP_STRICT
// Product build: the same sum with the products regrouped, acc_c = 2 Re sum_m (var_hat_c[m] phase_j[m] phase_k[m])
// phase_i[m] -- the bracket does not depend on i, so each wave (one (j, k) row of a block) forms it once per mode
// (lane m, kept in LDS) and every cell costs 6 FMAs, 2 coalesced loads and 3 broadcast LDS reads per mode
// instead of two complex products, three complex MACs and 8 loads (1.45 -> 0.5 ms on 256^3 with 30 modes).
// Differs from the reference's grouping (few_modes_ft.cpp:330-347, kept by the parity build below) in the last
// bits only.
constexpr int kMaxRowModes = 64;
// Who keeps what: the phases along x1 belong to the CELL COLUMN (mode, i), the coefficients A to the ROW (mode, j, k).
// Read per row, the table phases_i of a block -- 2 M nx1 doubles, 61 KB for 30 modes on 128-cell rows -- misses the L1
// and came out of the L2 once per wave: 4 GB per launch on 256^3, 0.57 ms at the L2's rate; staged in the LDS per
// workgroup it was the LDS's rate instead, 0.43 ms.  So a lane now keeps its column's phases in REGISTERS, kModeChunk
// modes at a time, and works through the kRowsPerWave rows of its wave with them: per mode and cell six FMAs and the
// row's three coefficient pairs as broadcast LDS reads, nothing else.  The rows' coefficients are formed once per wave
// (lane = mode) into a slice of the LDS of its own; modes beyond M are padded with zero coefficients.
constexpr int kModeChunk = 10;
constexpr int kRowsPerWave = 8;
__global__ void __launch_bounds__(256)
fmft_inverse_rows_kernel(PackView pv, const apk_fmft_block *blocks, const double *var_hat, int M, int Mpad, int groups_per_block, int xpasses) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x, wv = threadIdx.y;
  const int xp = blockIdx.x % xpasses, bg = blockIdx.x / xpasses;
  const int b = bg / groups_per_block, g = bg - b * groups_per_block;
  const apk_fmft_block blk = blocks[b];
  const int n1 = pv.nx1, n2 = pv.nx2, n3 = pv.nx3;
  const int rows = n2 * n3;
  const int r0 = (g * 4 + wv) * kRowsPerWave;
  double *A = lds + (size_t)wv * kRowsPerWave * Mpad * 6;  // [row][mode][6], this wave's
  for (int m = lane; m < Mpad; m += 64) {
    double vr[3] = {0.0, 0.0, 0.0}, vi[3] = {0.0, 0.0, 0.0};
    if (m < M) {
#pragma unroll
      for (int c = 0; c < 3; ++c) vr[c] = var_hat[(c * M + m) * 2], vi[c] = var_hat[(c * M + m) * 2 + 1];
    }
#pragma unroll
    for (int q = 0; q < kRowsPerWave; ++q) {
      const int r = (r0 + q < rows) ? r0 + q : rows - 1;
      const int ko = r / n2, jo = r - ko * n2;
      double pr = 0.0, pim = 0.0;
      if (m < M) {
        const double jr = blk.phases_j[m * n2 + jo], ji = blk.phases_j[(M + m) * n2 + jo];
        const double kr = blk.phases_k[m * n3 + ko], ki = blk.phases_k[(M + m) * n3 + ko];
        pr = jr * kr - ji * ki, pim = jr * ki + ji * kr;
      }
      double *a = A + ((size_t)q * Mpad + m) * 6;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        a[2 * c] = vr[c] * pr - vi[c] * pim;
        a[2 * c + 1] = vr[c] * pim + vi[c] * pr;
      }
    }
  }
  // (the slice is this wave's own: its LDS writes are complete before its reads below are issued)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // two cells per lane (io, io + 64): a broadcast read hands its 16 bytes to all 64 lanes -- 1 KB of LDS bandwidth per
  // instruction, which is what bounded the one-cell form (0.43 ms) -- and serves both
  const int io0 = xp * 128 + lane, io1 = io0 + 64;
  const int il0 = io0 < n1 ? io0 : n1 - 1, il1 = io1 < n1 ? io1 : n1 - 1;
  double acc[kRowsPerWave][3], bcc[kRowsPerWave][3];
#pragma unroll
  for (int q = 0; q < kRowsPerWave; ++q) acc[q][0] = acc[q][1] = acc[q][2] = bcc[q][0] = bcc[q][1] = bcc[q][2] = 0.0;
  for (int c0 = 0; c0 < Mpad; c0 += kModeChunk) {
    double ir[kModeChunk], ii[kModeChunk], jr[kModeChunk], ji[kModeChunk];
#pragma unroll
    for (int u = 0; u < kModeChunk; ++u) {
      const int m = (c0 + u < M) ? c0 + u : M - 1;  // (padding: its coefficients are zero)
      ir[u] = blk.phases_i[(int64_t)m * n1 + il0];
      ii[u] = blk.phases_i[(int64_t)(M + m) * n1 + il0];
      jr[u] = blk.phases_i[(int64_t)m * n1 + il1];
      ji[u] = blk.phases_i[(int64_t)(M + m) * n1 + il1];
    }
#pragma unroll
    for (int q = 0; q < kRowsPerWave; ++q) {
      const double *a = A + ((size_t)q * Mpad + c0) * 6;
#pragma unroll
      for (int u = 0; u < kModeChunk; ++u) {
        const double a0 = a[6 * u + 0], a1 = a[6 * u + 1], a2 = a[6 * u + 2], a3 = a[6 * u + 3], a4 = a[6 * u + 4], a5 = a[6 * u + 5];
        acc[q][0] += a0 * ir[u] - a1 * ii[u];
        acc[q][1] += a2 * ir[u] - a3 * ii[u];
        acc[q][2] += a4 * ir[u] - a5 * ii[u];
        bcc[q][0] += a0 * jr[u] - a1 * ji[u];
        bcc[q][1] += a2 * jr[u] - a3 * ji[u];
        bcc[q][2] += a4 * jr[u] - a5 * ji[u];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kRowsPerWave; ++q) {
    const int r = r0 + q;
    if (r < rows) {
      const int ko = r / n2, jo = r - ko * n2;
      const int64_t cell = (int64_t)(pv.ks + ko) * pv.sk + (int64_t)(pv.js + jo) * pv.sj + pv.is;
      if (io0 < n1) {
        blk.acc[0 * pv.sn + cell + io0] = 2. * acc[q][0];
        blk.acc[1 * pv.sn + cell + io0] = 2. * acc[q][1];
        blk.acc[2 * pv.sn + cell + io0] = 2. * acc[q][2];
      }
      if (io1 < n1) {
        blk.acc[0 * pv.sn + cell + io1] = 2. * bcc[q][0];
        blk.acc[1 * pv.sn + cell + io1] = 2. * bcc[q][1];
        blk.acc[2 * pv.sn + cell + io1] = 2. * bcc[q][2];
      }
    }
  }
}
#endif

// few_modes_ft.cpp:330-347
__global__ void __launch_bounds__(256)
fmft_inverse_kernel(PackView pv, const apk_fmft_block *blocks, const double *var_hat, int M) {
  int b, k, j, i;
  if (!interior_of(pv, b, k, j, i)) return;
  const apk_fmft_block blk = blocks[b];
  // phases(c, m, idx), idx fastest (the reference's variable shape {2, num_modes, nx}):
  // phases_i is a coalesced row per (c, m); phases_j / phases_k are wave-uniform
  const double *pi = blk.phases_i + (i - pv.is);
  const double *pj = blk.phases_j + (j - pv.js);
  const double *pk = blk.phases_k + (k - pv.ks);
  const int64_t n1 = pv.nx1, n2 = pv.nx2, n3 = pv.nx3;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int m = 0; m < M; ++m) {
    const double ir = pi[m * n1], ii = pi[(M + m) * n1];
    const double jr = pj[m * n2], ji = pj[(M + m) * n2];
    const double kr = pk[m * n3], ki = pk[(M + m) * n3];
    // phase = phase_i * phase_j * phase_k (complex products, left to right)
    const double pr = ir * jr - ii * ji, pim = ir * ji + ii * jr;
    const double qr = pr * kr - pim * ki, qi = pr * ki + pim * kr;
    s0 += 2. * (var_hat[(0 * M + m) * 2] * qr - var_hat[(0 * M + m) * 2 + 1] * qi);
    s1 += 2. * (var_hat[(1 * M + m) * 2] * qr - var_hat[(1 * M + m) * 2 + 1] * qi);
    s2 += 2. * (var_hat[(2 * M + m) * 2] * qr - var_hat[(2 * M + m) * 2 + 1] * qi);
  }
  const int64_t cell = k * pv.sk + j * pv.sj + i;
  blk.acc[0 * pv.sn + cell] = s0;
  blk.acc[1 * pv.sn + cell] = s1;
  blk.acc[2 * pv.sn + cell] = s2;
}

APK_DEV double wsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// generic "NQ sums per workgroup" epilogue: partial[wg][NQ]
template <int NQ>
APK_DEV void store_partials(const double (&h)[NQ], double *partial) {
  __shared__ double part[4][NQ];
  const int tid = threadIdx.y * 64 + threadIdx.x;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double s = wsum(h[q]);
    if ((tid & 63) == 0) part[tid >> 6][q] = s;
  }
  __syncthreads();
  if (tid < NQ) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[(int64_t)wg * NQ + tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
  }
}

template <int NQ>
__global__ void __launch_bounds__(256) final_sum_kernel(const double *partial, int nwg, double *out) {
  // NQ <= 8 quantities x 32 lanes each, fixed order => deterministic
  const int q = threadIdx.x / 32, lane = threadIdx.x % 32;
  if (q >= NQ) return;
  double s = 0.0;
  for (int w = lane; w < nwg; w += 32) s += partial[(int64_t)w * NQ + q];
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 32);
  if (lane == 0) out[q] = s;
}

// first level of the final sum when there are many partials (one per 256 cells: 65536 on 256^3, which
// the single workgroup above walks in 2048 dependent steps = 0.6 ms): kMidGroups workgroups each sum a
// contiguous share in a fixed order; the single workgroup then adds kMidGroups values per quantity
constexpr int kMidGroups = 256;
template <int NQ>
__global__ void __launch_bounds__(256) mid_sum_kernel(const double *partial, int nwg, double *out) {
  const int chunk = (nwg + kMidGroups - 1) / kMidGroups;
  const int lo = blockIdx.x * chunk, hi = (lo + chunk < nwg) ? lo + chunk : nwg;
  double h[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) h[q] = 0.0;
  for (int w = lo + (int)threadIdx.x; w < hi; w += 256) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) h[q] += partial[(int64_t)w * NQ + q];
  }
  __shared__ double part[4][NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double v = wsum(h[q]);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < NQ)
    out[(int64_t)blockIdx.x * NQ + threadIdx.x] =
        (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// turbulence.cpp:395-413
__global__ void __launch_bounds__(256)
turb_mean_momentum_kernel(PackView pv, const apk_fmft_block *blocks, double *partial) {
  int b, k, j, i;
  double h[4] = {0, 0, 0, 0};
  if (interior_of(pv, b, k, j, i)) {
    const apk_block_desc blk = pv.blocks[b];
    const double vol = blk.dx[0] * blk.dx[1] * blk.dx[2];
    const int64_t cell = k * pv.sk + j * pv.sj + i;
    const double den = blk.cons[IDN * pv.sn + cell];
    const double *acc = blocks[b].acc + cell;
    h[0] = den * vol;
    h[1] = den * acc[0 * pv.sn] * vol;
    h[2] = den * acc[1 * pv.sn] * vol;
    h[3] = den * acc[2 * pv.sn] * vol;
  }
  store_partials<4>(h, partial);
}

// turbulence.cpp:421-430
__global__ void __launch_bounds__(256)
turb_remove_mean_kernel(PackView pv, const apk_fmft_block *blocks, double m1, double m2, double m3,
                        double *partial) {
  int b, k, j, i;
  double h[1] = {0};
  if (interior_of(pv, b, k, j, i)) {
    const apk_block_desc blk = pv.blocks[b];
    const double vol = blk.dx[0] * blk.dx[1] * blk.dx[2];
    const int64_t cell = k * pv.sk + j * pv.sj + i;
    double *acc = blocks[b].acc + cell;
    const double a0 = acc[0 * pv.sn] - m1, a1 = acc[1 * pv.sn] - m2, a2 = acc[2 * pv.sn] - m3;
    acc[0 * pv.sn] = a0;
    acc[1 * pv.sn] = a1;
    acc[2 * pv.sn] = a2;
    h[0] = sqr(a0) * vol + sqr(a1) * vol + sqr(a2) * vol;
  }
  store_partials<1>(h, partial);
}

// turbulence.cpp:446-469
__global__ void __launch_bounds__(256)
turb_apply_kernel(PackView pv, const apk_fmft_block *blocks, double norm, double dt) {
  int b, k, j, i;
  if (!interior_of(pv, b, k, j, i)) return;
  const apk_block_desc blk = pv.blocks[b];
  const int64_t cell = k * pv.sk + j * pv.sj + i;
  double *acc = blocks[b].acc + cell;
  double *u = blk.cons + cell;
  const double a0 = acc[0 * pv.sn] * norm, a1 = acc[1 * pv.sn] * norm, a2 = acc[2 * pv.sn] * norm;
  acc[0 * pv.sn] = a0;
  acc[1 * pv.sn] = a1;
  acc[2 * pv.sn] = a2;
  const double den = u[IDN * pv.sn];
  const double qa = dt * den;
  const double m1 = u[IM1 * pv.sn], m2 = u[IM2 * pv.sn], m3 = u[IM3 * pv.sn];
  u[IEN * pv.sn] += (m1 * dt * a0 + m2 * dt * a1 + m3 * dt * a2 +
                     (sqr(a0) + sqr(a1) + sqr(a2)) * qa * qa / (2 * den));
  u[IM1 * pv.sn] = m1 + qa * a0;
  u[IM2 * pv.sn] = m2 + qa * a1;
  u[IM3 * pv.sn] = m3 + qa * a2;
}

// turbulence.cpp:47-101
template <int FLUID>
__global__ void __launch_bounds__(256) turb_history_kernel(PackView pv, double gamma, double *partial) {
  int b, k, j, i;
  double h[3] = {0, 0, 0};
  if (interior_of(pv, b, k, j, i)) {
    const apk_block_desc blk = pv.blocks[b];
    const double vol = blk.dx[0] * blk.dx[1] * blk.dx[2];
    const double *w = blk.prim + k * pv.sk + j * pv.sj + i;
    const double d = w[IDN * pv.sn], p = w[IPR * pv.sn];
    const double vel2 = (w[IV1 * pv.sn] * w[IV1 * pv.sn] + w[IV2 * pv.sn] * w[IV2 * pv.sn] +
                         w[IV3 * pv.sn] * w[IV3 * pv.sn]);
    const double c_s = sqrt(gamma * p / d);
    const double e_kin = 0.5 * d * vel2;
    h[0] = sqrt(vel2) / c_s * vol;
    if constexpr (FLUID == APK_FLUID_GLMMHD) {
      const double B2 = (w[IB1 * pv.sn] * w[IB1 * pv.sn] + w[IB2 * pv.sn] * w[IB2 * pv.sn] +
                         w[IB3 * pv.sn] * w[IB3 * pv.sn]);
      const double e_mag = 0.5 * B2;
      h[1] = sqrt(e_kin / e_mag) * vol;
      h[2] = p / e_mag * vol;
    }
  }
  store_partials<3>(h, partial);
}

// field_loop::RelDivBHst (src/pgen/field_loop.cpp:60-95): sum of 0.5 |dx| |div B| / B0 * volume with
// centred differences of the cell-centred field and a FIXED normalisation B0
__global__ void __launch_bounds__(256) user_reldivb_kernel(PackView pv, double B0, double *partial) {
  int b, k, j, i;
  double h[1] = {0};
  if (interior_of(pv, b, k, j, i)) {
    const apk_block_desc blk = pv.blocks[b];
    const double vol = blk.dx[0] * blk.dx[1] * blk.dx[2];
    const double *u = blk.cons + k * pv.sk + j * pv.sj + i;
    const double *b1 = u + IB1 * pv.sn, *b2 = u + IB2 * pv.sn, *b3 = u + IB3 * pv.sn;
    double divb = (b1[1] - b1[-1]) / blk.dx[0] + (b2[pv.sj] - b2[-pv.sj]) / blk.dx[1];
    if (pv.ndim == 3) divb += (b3[pv.sk] - b3[-pv.sk]) / blk.dx[2];
    h[0] = 0.5 * (sqrt(sqr(blk.dx[0]) + sqr(blk.dx[1]) + sqr(blk.dx[2]))) * fabs(divb) / B0 * vol;
  }
  store_partials<1>(h, partial);
}

int ensure_partial_cap(apk_ctx *ctx, size_t n) {
  if (ctx->partial_cap >= n) return APK_OK;
  if (ctx->d_partial) (void)hipFree(ctx->d_partial);
  ctx->d_partial = nullptr;
  ctx->partial_cap = 0;
  if (hipMalloc(&ctx->d_partial, n * sizeof(double)) != hipSuccess) return APK_ERR_DEVICE;
  ctx->partial_cap = n;
  return APK_OK;
}

// run a partial-sum kernel result through the final reduction and bring NQ doubles to the host
// The kick with the work of the two tasks that follow it on the same cell: FillDerived (ConsToPrim, floors
// included, prim in place: the kick is the last thing to touch the cell before the next stage reads it) and
// the time-step estimate -- one pass over cons instead of three.
// STORE_PRIM = false (apk_turb_apply_dt): the primitives feed the time-step estimate only -- the next stage derives its
// input from the conserved state (apk_stage_args.prim_from_cons).  Of the conserved variables only the kicked ones
// (momenta, energy) and whatever a floor changed are written back.
template <int FLUID, bool WITH_DT, bool STORE_PRIM = true>
__global__ void __launch_bounds__(256)
turb_apply_fill_kernel(PackView pv, const apk_fmft_block *blocks, double norm, double dt, apk_eos eos, unsigned *flags,
                       unsigned long long *dt_bits) {
  constexpr int NV = nvars<FLUID>();
  int b, k, j, i;
  double lane_min = 1.7976931348623157e308;
  if (interior_of(pv, b, k, j, i)) {
    const apk_block_desc blk = pv.blocks[b];
    const int64_t cell = k * pv.sk + j * pv.sj + i;
    double *acc = blocks[b].acc + cell;
    double *u = blk.cons + cell;
    const double a0 = acc[0 * pv.sn] * norm, a1 = acc[1 * pv.sn] * norm, a2 = acc[2 * pv.sn] * norm;
    acc[0 * pv.sn] = a0;
    acc[1 * pv.sn] = a1;
    acc[2 * pv.sn] = a2;
    double un[NV], was[NV], w[NV], di;
#pragma unroll
    for (int n = 0; n < NV; ++n) was[n] = un[n] = u[n * pv.sn];
    const double den = un[IDN];
    const double qa = dt * den;
    const double m1 = un[IM1], m2 = un[IM2], m3 = un[IM3];
    un[IEN] += (m1 * dt * a0 + m2 * dt * a1 + m3 * dt * a2 + (sqr(a0) + sqr(a1) + sqr(a2)) * qa * qa / (2 * den));
    un[IM1] = m1 + qa * a0;
    un[IM2] = m2 + qa * a1;
    un[IM3] = m3 + qa * a2;
    const unsigned fl = cons_to_prim_cell<FLUID>(eos, un, w, di);
    if (fl) atomicOr(flags, fl);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const bool kicked = (n == IM1 || n == IM2 || n == IM3 || n == IEN);
      if (kicked || un[n] != was[n]) u[n * pv.sn] = un[n];  // (the others: only where a floor acted)
    }
    if constexpr (STORE_PRIM) {
      double *p = blk.prim + cell;
#pragma unroll
      for (int n = 0; n < NV; ++n) p[n * pv.sn] = w[n];
      for (int n = NV; n < pv.nvar; ++n) p[n * pv.sn] = u[n * pv.sn] * di;  // passive scalars
    }
    if constexpr (WITH_DT) {  // EstimateHyperbolicTimestep (hydro.cpp:845-895) on the fresh primitives
      double lx, ly = 0.0, lz = 0.0;
      if constexpr (FLUID == APK_FLUID_EULER) {
        lx = ly = lz = sound_speed(eos.gamma, w[IDN], w[IPR]);
      } else {
        lx = fast_speed(eos.gamma, w[IDN], w[IPR], w[IB1], w[IB2], w[IB3]);
        if (pv.ndim > 1) ly = fast_speed(eos.gamma, w[IDN], w[IPR], w[IB2], w[IB3], w[IB1]);
        if (pv.ndim > 2) lz = fast_speed(eos.gamma, w[IDN], w[IPR], w[IB3], w[IB1], w[IB2]);
      }
      lane_min = fmin(lane_min, blk.dx[0] / (fabs(w[IV1]) + lx));
      if (pv.ndim > 1) lane_min = fmin(lane_min, blk.dx[1] / (fabs(w[IV2]) + ly));
      if (pv.ndim > 2) lane_min = fmin(lane_min, blk.dx[2] / (fabs(w[IV3]) + lz));
    }
  }
  if constexpr (WITH_DT) {
    // one candidate per workgroup, and an atomic only if it beats the word's current value (a plain read: a
    // stale, larger value merely costs an atomic that changes nothing) -- 65536 workgroups on 256^3 would
    // otherwise queue on one address for milliseconds
    __shared__ double wmin[4];
    double m = lane_min;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if (threadIdx.x == 0) wmin[threadIdx.y] = m;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
      m = fmin(fmin(wmin[0], wmin[1]), fmin(wmin[2], wmin[3]));
      const double cur = __longlong_as_double((long long)*reinterpret_cast<volatile unsigned long long *>(dt_bits));
      if (m < cur) atomicMin(dt_bits, (unsigned long long)__double_as_longlong(m));
    }
  }
}

template <int NQ>
int finish_sums(apk_ctx *ctx, int nwg, double *out, hipStream_t s) {
  double *d_out = ctx->d_partial + (size_t)nwg * NQ;
  if (nwg > 8 * kMidGroups) {  // (callers reserve nwg * 4 + kMidGroups * 4 + 8 doubles)
    double *d_mid = d_out + NQ;
    hipLaunchKernelGGL(mid_sum_kernel<NQ>, dim3(kMidGroups), dim3(256), 0, s, ctx->d_partial, nwg, d_mid);
    hipLaunchKernelGGL(final_sum_kernel<NQ>, dim3(1), dim3(256), 0, s, d_mid, kMidGroups, d_out);
  } else {
    hipLaunchKernelGGL(final_sum_kernel<NQ>, dim3(1), dim3(256), 0, s, ctx->d_partial, nwg, d_out);
  }
  auto *h = static_cast<double *>(ctx->h_pinned) + 16;
  if (hipMemcpyAsync(h, d_out, NQ * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return APK_ERR_DEVICE;
  if (hipStreamSynchronize(s) != hipSuccess) return APK_ERR_DEVICE;
  std::memcpy(out, h, NQ * sizeof(double));
  return APK_OK;
}

}  // namespace
}  // namespace apk

using namespace apk;

namespace {
// The acceleration field lives in arrays of its own, [3][Nk][Nj][Ni] per block, addressed with the pack's cell offsets:
// packs with explicit strides (apk_pack_desc.stride) are refused rather than read across rows.
static bool natural_layout(const apk_pack *md) {
  const PackView &v = md->view;
  return v.sj == v.ni && v.sk == (int64_t)v.ni * v.nj && v.sn == (int64_t)v.ni * v.nj * v.nk;
}
#define APK_TURB_NATURAL(ctx, md) \
  do { if (!natural_layout(md)) return set_err((ctx), APK_ERR_UNSUPPORTED, "turbulence driver: packs with explicit strides are not supported"); } while (0)

int turb_apply_fill_impl(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, double norm, double dt, int fluid, const apk_eos *eos,
                         int estimate_dt, bool store_prim, apk_stream_t stream) {
  if (!ctx || !md || !f || !eos || f->nblocks != md->view.nblocks || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9) || (!store_prim && md->view.nvar != md->view.nhydro))
    return set_err(ctx, APK_ERR_INVALID, "apk_turb_apply_fill: bad argument");
  APK_TURB_NATURAL(ctx, md);
  for (const auto &b : md->h_blocks)
    if (store_prim && !b.prim) return set_err(ctx, APK_ERR_INVALID, "apk_turb_apply_fill: block without prim pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  unsigned long long *dt_bits = ctx->d_u64 + 4;  // the stage's word: apk_stage_dt_read / apk_stage_dt_flags_read
  if (estimate_dt && apk::prepare_dt_word(ctx, s) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "apk_turb_apply_fill", hipGetLastError());
  const dim3 g = igrid(md->view), blk(64, 4, 1);
#define APK_LAUNCH_KICK(FL, DT, SP) \
  hipLaunchKernelGGL((turb_apply_fill_kernel<FL, DT, SP>), g, blk, 0, s, md->view, f->d_blocks, norm, dt, *eos, ctx->d_flags, dt_bits)
  if (fluid == APK_FLUID_EULER) {
    if (!store_prim) APK_LAUNCH_KICK(APK_FLUID_EULER, true, false);
    else if (estimate_dt) APK_LAUNCH_KICK(APK_FLUID_EULER, true, true);
    else APK_LAUNCH_KICK(APK_FLUID_EULER, false, true);
  } else {
    if (!store_prim) APK_LAUNCH_KICK(APK_FLUID_GLMMHD, true, false);
    else if (estimate_dt) APK_LAUNCH_KICK(APK_FLUID_GLMMHD, true, true);
    else APK_LAUNCH_KICK(APK_FLUID_GLMMHD, false, true);
  }
#undef APK_LAUNCH_KICK
  return hipGetLastError() == hipSuccess ? APK_OK : set_err(ctx, APK_ERR_DEVICE, "turb_apply_fill launch", hipGetLastError());
}
}  // namespace

extern "C" {

int apk_fmft_create(apk_ctx *ctx, const apk_fmft_block *blocks, int nblocks, int num_modes, apk_fmft **out) {
  if (!ctx || !blocks || !out || nblocks <= 0 || num_modes <= 0) return APK_ERR_INVALID;
  *out = nullptr;
  apk_fmft *f = new (std::nothrow) apk_fmft();
  if (!f) return APK_ERR_INVALID;
  f->nblocks = nblocks;
  f->num_modes = num_modes;
  hipError_t e = hipMalloc(&f->d_blocks, sizeof(apk_fmft_block) * nblocks);
  if (e == hipSuccess) e = hipMemcpy(f->d_blocks, blocks, sizeof(apk_fmft_block) * nblocks, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&f->d_var_hat, sizeof(double) * 3 * num_modes * 2);
  if (e != hipSuccess) {
    apk_fmft_destroy(f);
    return set_err(ctx, APK_ERR_DEVICE, "apk_fmft_create", e);
  }
  *out = f;
  return APK_OK;
}

void apk_fmft_destroy(apk_fmft *f) {
  if (!f) return;
  if (f->d_blocks) (void)hipFree(f->d_blocks);
  if (f->d_var_hat) (void)hipFree(f->d_var_hat);
  delete f;
}

int apk_fmft_inverse(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, const double *var_hat_host,
                     apk_stream_t stream) {
  if (!ctx || !md || !f || !var_hat_host || f->nblocks != md->view.nblocks)
    return set_err(ctx, APK_ERR_INVALID, "apk_fmft_inverse: bad argument");
  APK_TURB_NATURAL(ctx, md);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  APK_HIP_TRY(ctx, hipMemcpyAsync(f->d_var_hat, var_hat_host, sizeof(double) * 3 * f->num_modes * 2,
                                  hipMemcpyHostToDevice, s));
#ifndef APK_FP_STRICT
  const PackView &pv = md->view;
  if (f->num_modes <= kMaxRowModes) {
    const int Mpad = (f->num_modes + kModeChunk - 1) / kModeChunk * kModeChunk;
    const int groups = (pv.nx2 * pv.nx3 + 4 * kRowsPerWave - 1) / (4 * kRowsPerWave), xpasses = (pv.nx1 + 127) / 128;
    const size_t lds = sizeof(double) * 4 * kRowsPerWave * (size_t)Mpad * 6;
    // (30 modes: 46 KB; more than 64 KB -- 42 modes and up -- has to be asked for: gfx950 has 160 KB per CU)
    constexpr int lds_max = (int)sizeof(double) * 4 * kRowsPerWave * ((kMaxRowModes + kModeChunk - 1) / kModeChunk * kModeChunk) * 6;
    static const bool lds_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(fmft_inverse_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   lds_max) == hipSuccess;
    if (lds_ok || lds <= 64 * 1024) {
      hipLaunchKernelGGL(fmft_inverse_rows_kernel, dim3((unsigned)(groups * pv.nblocks * xpasses), 1, 1), dim3(64, 4, 1), lds, s, pv,
                         f->d_blocks, f->d_var_hat, f->num_modes, Mpad, groups, xpasses);
      return hipGetLastError() == hipSuccess ? APK_OK : set_err(ctx, APK_ERR_DEVICE, "fmft_inverse launch", hipGetLastError());
    }
  }
#endif
  hipLaunchKernelGGL(fmft_inverse_kernel, igrid(md->view), dim3(64, 4, 1), 0, s, md->view, f->d_blocks,
                     f->d_var_hat, f->num_modes);
  return hipGetLastError() == hipSuccess ? APK_OK : set_err(ctx, APK_ERR_DEVICE, "fmft_inverse launch", hipGetLastError());
}

int apk_turb_mean_momentum(apk_ctx *ctx, const apk_pack *md, const apk_fmft *f, double *sums4,
                           apk_stream_t stream) {
  if (!ctx || !md || !f || !sums4 || f->nblocks != md->view.nblocks) return set_err(ctx, APK_ERR_INVALID, "apk_turb_mean_momentum: bad argument");
  APK_TURB_NATURAL(ctx, md);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 g = igrid(md->view);
  const int nwg = g.x * g.y * g.z;
  if (ensure_partial_cap(ctx, (size_t)nwg * 4 + kMidGroups * 4 + 8) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "partial buffer");
  hipLaunchKernelGGL(turb_mean_momentum_kernel, g, dim3(64, 4, 1), 0, s, md->view, f->d_blocks, ctx->d_partial);
  return finish_sums<4>(ctx, nwg, sums4, s);
}

int apk_turb_remove_mean(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, const double *sums4, double *ampl_sum,
                         apk_stream_t stream) {
  if (!ctx || !md || !f || !sums4 || !ampl_sum || f->nblocks != md->view.nblocks)
    return set_err(ctx, APK_ERR_INVALID, "apk_turb_remove_mean: bad argument");
  APK_TURB_NATURAL(ctx, md);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 g = igrid(md->view);
  const int nwg = g.x * g.y * g.z;
  if (ensure_partial_cap(ctx, (size_t)nwg * 4 + kMidGroups * 4 + 8) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "partial buffer");
  hipLaunchKernelGGL(turb_remove_mean_kernel, g, dim3(64, 4, 1), 0, s, md->view, f->d_blocks, sums4[1] / sums4[0],
                     sums4[2] / sums4[0], sums4[3] / sums4[0], ctx->d_partial);
  return finish_sums<1>(ctx, nwg, ampl_sum, s);
}

int apk_turb_apply(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, double norm, double dt, apk_stream_t stream) {
  if (!ctx || !md || !f || f->nblocks != md->view.nblocks) return set_err(ctx, APK_ERR_INVALID, "apk_turb_apply: bad argument");
  APK_TURB_NATURAL(ctx, md);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(turb_apply_kernel, igrid(md->view), dim3(64, 4, 1), 0, s, md->view, f->d_blocks, norm, dt);
  return hipGetLastError() == hipSuccess ? APK_OK : set_err(ctx, APK_ERR_DEVICE, "turb_apply launch", hipGetLastError());
}

int apk_turb_apply_fill(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, double norm, double dt, int fluid, const apk_eos *eos,
                        int estimate_dt, apk_stream_t stream) {
  return turb_apply_fill_impl(ctx, md, f, norm, dt, fluid, eos, estimate_dt, true, stream);
}
int apk_turb_apply_dt(apk_ctx *ctx, const apk_pack *md, apk_fmft *f, double norm, double dt, int fluid, const apk_eos *eos,
                      apk_stream_t stream) {
  return turb_apply_fill_impl(ctx, md, f, norm, dt, fluid, eos, 1, false, stream);
}

int apk_turbulence_history(apk_ctx *ctx, const apk_pack *md, int fluid, double gamma, double *out3,
                           apk_stream_t stream) {
  if (!ctx || !md || !out3) return set_err(ctx, APK_ERR_INVALID, "apk_turbulence_history: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 g = igrid(md->view);
  const int nwg = g.x * g.y * g.z;
  if (ensure_partial_cap(ctx, (size_t)nwg * 4 + kMidGroups * 4 + 8) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "partial buffer");
  if (fluid == APK_FLUID_EULER)
    hipLaunchKernelGGL(turb_history_kernel<APK_FLUID_EULER>, g, dim3(64, 4, 1), 0, s, md->view, gamma, ctx->d_partial);
  else
    hipLaunchKernelGGL(turb_history_kernel<APK_FLUID_GLMMHD>, g, dim3(64, 4, 1), 0, s, md->view, gamma, ctx->d_partial);
  return finish_sums<3>(ctx, nwg, out3, s);
}

int apk_history_user_reldivb(apk_ctx *ctx, const apk_pack *md, double B0, double *out, apk_stream_t stream) {
  if (!ctx || !md || !out || md->view.nhydro != 9 || md->view.ndim < 2 || !(B0 > 0.0))
    return set_err(ctx, APK_ERR_INVALID, "apk_history_user_reldivb: bad argument (GLM-MHD, >= 2-D, B0 > 0)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 g = igrid(md->view);
  const int nwg = g.x * g.y * g.z;
  if (ensure_partial_cap(ctx, (size_t)nwg * 4 + kMidGroups * 4 + 8) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "partial buffer");
  hipLaunchKernelGGL(user_reldivb_kernel, g, dim3(64, 4, 1), 0, s, md->view, B0, ctx->d_partial);
  return finish_sums<1>(ctx, nwg, out, s);
}

}  // extern "C"
