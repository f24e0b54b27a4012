// kernels_block.hip -- the streaming kernels either side of the flux sweeps:
// flux-divergence/RK update, Dedner source, cons->prim, dt and history reductions,
// first-order flux correction, strided box copies (ghost zones).
// All are HBM-bound: one lane per cell along x1 (coalesced 512 B per wave per variable),
// variables looped inside the lane so each cell's nvar doubles stream once.
#include "apk_internal.hpp"
#include "hydro_math.hpp"

namespace apk {

namespace {

struct CellIdx {
  int b, k, j, i;
  bool ok;
};

// interior cell of this lane; grid = (ceil(nx1/64), ceil(nx2/4), nx3*nblocks), block (64,4)
APK_DEV CellIdx interior_cell(const PackView &pv) {
  CellIdx c;
  int io, jo;
  c.ok = rect_ij(pv.nx1, pv.nx2, io, jo);
  c.i = pv.is + io;
  c.j = pv.js + jo;
  c.b = blockIdx.z / pv.nx3;
  c.k = pv.ks + blockIdx.z % pv.nx3;
  return c;
}

inline dim3 interior_grid(const PackView &pv) { return rect_grid(pv.nx1, pv.nx2, pv.nx3 * pv.nblocks); }

// Parthenon Update::FluxDivHelper (un-vendored):
// du = A1 F1(i+1) - A1 F1(i) [+ A2 ..][+ A3 ..];  return -du / V
APK_DEV double flux_div(const PackView &pv, const apk_block_desc &blk, int64_t idx,
                        const double (&area)[3], double vol) {
  const double *f1 = blk.flux[0] + idx;
  double du = (area[0] * f1[1] - area[0] * f1[0]);
  if (pv.ndim >= 2) {
    const double *f2 = blk.flux[1] + idx;
    du += (area[1] * f2[pv.sj] - area[1] * f2[0]);
  }
  if (pv.ndim == 3) {
    const double *f3 = blk.flux[2] + idx;
    du += (area[2] * f3[pv.sk] - area[2] * f3[0]);
  }
  return -du / vol;
}

APK_DEV void block_areas(const apk_block_desc &blk, double (&area)[3], double &vol) {
  area[0] = blk.dx[1] * blk.dx[2];
  area[1] = blk.dx[0] * blk.dx[2];
  area[2] = blk.dx[0] * blk.dx[1];
  vol = blk.dx[0] * blk.dx[1] * blk.dx[2];
}

// ---- UpdateWithFluxDivergence (call site src/hydro/hydro_driver.cpp:534-537) ----------
__global__ void __launch_bounds__(256)
update_flux_div_kernel(PackView u0, PackView u1, double gam0, double gam1, double beta_dt) {
  const CellIdx c = interior_cell(u0);
  if (!c.ok) return;
  const apk_block_desc b0 = u0.blocks[c.b];
  const double *c1 = u1.blocks[c.b].cons;
  double area[3], vol;
  block_areas(b0, area, vol);
  const int64_t cell = c.k * u0.sk + c.j * u0.sj + c.i;
  for (int n = 0; n < u0.nvar; ++n) {
    const int64_t idx = n * u0.sn + cell;
    // gam0 == 0 (first stage of every integrator): the old contents of u0 are not an input
    const double old = (gam0 != 0.0) ? b0.cons[idx] : 0.0;
    b0.cons[idx] = gam0 * old + gam1 * c1[idx] + beta_dt * flux_div(u0, b0, idx, area, vol);
  }
}

// ---- DednerSource (src/hydro/glmmhd/dedner_source.cpp:17-75) ----------------------------
template <bool EXTENDED>
__global__ void __launch_bounds__(256)
dedner_kernel(PackView pv, double coeff, double beta_dt) {
  const CellIdx c = interior_cell(pv);
  if (!c.ok) return;
  const apk_block_desc blk = pv.blocks[c.b];
  const int64_t cell = c.k * pv.sk + c.j * pv.sj + c.i;
  double *u = blk.cons + cell;
  if constexpr (EXTENDED) {
    const double *w = blk.prim + cell;
    const int64_t so = (pv.ndim >= 2) ? pv.sj : 0;
    const int64_t ko = (pv.ndim >= 3) ? pv.sk : 0;  // k_offset = 0 in 2-D (:34-40)
    const double *b1 = w + IB1 * pv.sn, *b2 = w + IB2 * pv.sn, *b3 = w + IB3 * pv.sn;
    const double *ps = w + IPS * pv.sn;
    const double divB = 0.5 * ((b1[1] - b1[-1]) / blk.dx[0] + (b2[so] - b2[-so]) / blk.dx[1] +
                               (b3[ko] - b3[-ko]) / blk.dx[2]);
    u[IM1 * pv.sn] -= beta_dt * divB * b1[0];
    u[IM2 * pv.sn] -= beta_dt * divB * b2[0];
    u[IM3 * pv.sn] -= beta_dt * divB * b3[0];
    u[IEN * pv.sn] -= 0.5 * beta_dt *
                      (b1[0] * (ps[1] - ps[-1]) / blk.dx[0] +
                       b2[0] * (ps[so] - ps[-so]) / blk.dx[1] +
                       b3[0] * (ps[ko] - ps[-ko]) / blk.dx[2]);
  }
  u[IPS * pv.sn] *= coeff;
}

// ---- wave/workgroup reductions -------------------------------------------------------------
APK_DEV double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  return v;
}
APK_DEV double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---- ConservedToPrimitive over the ENTIRE block (src/eos/adiabatic_hydro.cpp:33-55) -----
// store_vars: bit n set = primitive n is stored (all ones: FillDerived; apk_cons_to_prim_dt_select stores a subset)
template <int FLUID>
APK_DEV void cons_to_prim_at_w(const PackView &pv, const apk_block_desc &blk, const apk_eos &eos, unsigned *flags,
                               int64_t cell, double (&w)[nvars<FLUID>()], unsigned store_vars = ~0u) {
  constexpr int NV = nvars<FLUID>();
  double u[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) u[n] = blk.cons[n * pv.sn + cell];
  const double d_in = u[IDN], m1 = u[IM1], m2 = u[IM2], m3 = u[IM3], e_in = u[IEN];
  double di;
  const unsigned fl = cons_to_prim_cell<FLUID>(eos, u, w, di);
  if (fl) atomicOr(flags, fl);
  // floors / ceilings write the conserved state back (adiabatic_hydro.hpp:81,92-136);
  // only touched entries are stored so the common no-floor case stays read-only on cons.
  if (u[IDN] != d_in) blk.cons[IDN * pv.sn + cell] = u[IDN];
  if (u[IM1] != m1) blk.cons[IM1 * pv.sn + cell] = u[IM1];
  if (u[IM2] != m2) blk.cons[IM2 * pv.sn + cell] = u[IM2];
  if (u[IM3] != m3) blk.cons[IM3 * pv.sn + cell] = u[IM3];
  if (u[IEN] != e_in) blk.cons[IEN * pv.sn + cell] = u[IEN];
#pragma unroll
  for (int n = 0; n < NV; ++n)
    if ((store_vars >> n) & 1u) blk.prim[n * pv.sn + cell] = w[n];
  for (int n = NV; n < pv.nvar; ++n)  // passive scalars (:139-141)
    if (store_vars == ~0u) blk.prim[n * pv.sn + cell] = blk.cons[n * pv.sn + cell] * di;
}
template <int FLUID>
APK_DEV void cons_to_prim_at(const PackView &pv, const apk_block_desc &blk, const apk_eos &eos, unsigned *flags,
                             int64_t cell) {
  double w[nvars<FLUID>()];
  cons_to_prim_at_w<FLUID>(pv, blk, eos, flags, cell, w);
}

// EstimateHyperbolicTimestep (hydro.cpp:845-895) of one cell from its primitives in registers: min over the active
// directions of dx_d / (|v_d| + c_d) -- the expressions of min_dt_kernel
template <int FLUID>
APK_DEV double cell_dt_hyp(const PackView &pv, const apk_block_desc &blk, double gamma, const double (&w)[nvars<FLUID>()]) {
  double lx, ly = 0.0, lz = 0.0;
  if constexpr (FLUID == APK_FLUID_EULER) {
    lx = ly = lz = sound_speed(gamma, w[IDN], w[IPR]);
  } else {
    lx = fast_speed(gamma, w[IDN], w[IPR], w[IB1], w[IB2], w[IB3]);
    if (pv.ndim > 1) ly = fast_speed(gamma, w[IDN], w[IPR], w[IB2], w[IB3], w[IB1]);
    if (pv.ndim > 2) lz = fast_speed(gamma, w[IDN], w[IPR], w[IB3], w[IB1], w[IB2]);
  }
  double m = blk.dx[0] / (fabs(w[IV1]) + lx);
  if (pv.ndim > 1) m = fmin(m, blk.dx[1] / (fabs(w[IV2]) + ly));
  if (pv.ndim > 2) m = fmin(m, blk.dx[2] / (fabs(w[IV3]) + lz));
  return m;
}
// (64, 4) workgroups: one candidate per workgroup, and an atomic only if it beats the word's current value (a plain
// read: a stale, larger value merely costs an atomic that changes nothing).  Every thread of the workgroup calls it.
APK_DEV void block_min_to_word(double lane_min, unsigned long long *word) {
  __shared__ double wmin[4];
  const double m = wave_min(lane_min);
  if (threadIdx.x == 0) wmin[threadIdx.y] = m;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    const double g = fmin(fmin(wmin[0], wmin[1]), fmin(wmin[2], wmin[3]));
    // (blockDim must be (64, 4, 1): wmin[4] / threadIdx.y above)
    const double cur = __longlong_as_double((long long)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (g < cur) atomicMin(word, (unsigned long long)__double_as_longlong(g));
  }
}

// WITH_DT: EstimateHyperbolicTimestep (hydro.cpp:845-895) of the interior cells on the way, from the primitives in
// registers -- the values min_dt_kernel would read back -- reduced into *dt_bits (apk_cons_to_prim_dt).
template <int FLUID, bool WITH_DT>
__global__ void __launch_bounds__(256)
cons_to_prim_kernel(PackView pv, apk_eos eos, unsigned *flags, unsigned long long *dt_bits, int depth, const int *face_nbr,
                    unsigned store_vars) {
  // depth >= 0: only the cells within that many layers of the interior (the shell a shallow ghost exchange has filled) --
  // the launch covers that box, cell after cell (16^3 blocks with four ghost layers, depth 2: 20^3 of 24^3 cells, 32
  // workgroups a block instead of 72)
  const bool boxed = WITH_DT && depth >= 0;
  int i, j, k, b;
  bool ok;
  if (boxed) {
    const int ei = pv.nx1 + 2 * depth, ej = (pv.ndim >= 2) ? pv.nx2 + 2 * depth : pv.nj, ek = (pv.ndim >= 3) ? pv.nx3 + 2 * depth : pv.nk;
    // (the box's workgroups are dealt over grid x and y: a box of more than 65535 workgroups -- blocks of ~256^3 cells --
    // still has its own launch shape instead of falling back to the whole block)
    const unsigned f = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.y * 64u + threadIdx.x, plane = (unsigned)(ei * ej);
    const unsigned kk = f / plane, r = f - kk * plane, jj = r / (unsigned)ei;
    ok = kk < (unsigned)ek;
    i = pv.is - depth + (int)(r - jj * (unsigned)ei);
    j = ((pv.ndim >= 2) ? pv.js - depth : 0) + (int)jj;
    k = ((pv.ndim >= 3) ? pv.ks - depth : 0) + (int)kk;
    b = blockIdx.z;
  } else {
    ok = rect_ij(pv.ni, pv.nj, i, j);
    if (!WITH_DT && !ok) return;
    b = blockIdx.z / pv.nk;
    k = blockIdx.z % pv.nk;
  }
  if (WITH_DT && face_nbr) {  // (a ghost zone behind a face its readers cross by the face table: left as it is)
    const int *fn = face_nbr + 6 * b;  // (block-uniform: scalar loads, next to the descriptor's)
    const int f0 = fn[0], f1 = fn[1], f2 = fn[2], f3 = fn[3], f4 = fn[4], f5 = fn[5];
    const int ghost = ((i < pv.is) || (i > pv.ie)) + ((j < pv.js) || (j > pv.je)) + ((k < pv.ks) || (k > pv.ke));
    const int nb = (i < pv.is) ? f0 : (i > pv.ie) ? f1 : (j < pv.js) ? f2 : (j > pv.je) ? f3 : (k < pv.ks) ? f4 : f5;
    if (ghost == 1 && nb >= 0) ok = false;
  }
  double lane_min = 1.7976931348623157e308;
  if (ok) {
    const apk_block_desc blk = pv.blocks[b];
    double w[nvars<FLUID>()];
    cons_to_prim_at_w<FLUID>(pv, blk, eos, flags, k * pv.sk + j * pv.sj + i, w, store_vars);
    if constexpr (WITH_DT) {
      if (i >= pv.is && i <= pv.ie && j >= pv.js && j <= pv.je && k >= pv.ks && k <= pv.ke) lane_min = cell_dt_hyp<FLUID>(pv, blk, eos.gamma, w);
    }
  }
  if constexpr (WITH_DT) block_min_to_word(lane_min, dt_bits);
}

// Interior cells and the ghost cells straight behind a FACE of the block (at most one ghost coordinate):
// what the sweeps of the next stage read.  Rows behind edges and corners are skipped whole (no memory
// traffic): 26 % of the cells of a 16^3 block with nghost = 4.
template <int FLUID, bool WITH_DT>
__global__ void __launch_bounds__(256)
cons_to_prim_faces_kernel(PackView pv, apk_eos eos, unsigned *flags, const int *face_nbr, unsigned long long *dt_bits) {
  int i, j;
  bool ok = rect_ij(pv.ni, pv.nj, i, j);
  if (!WITH_DT && !ok) return;
  const int b = blockIdx.z / pv.nk;
  const int k = blockIdx.z % pv.nk;
  const int ghost = ((i < pv.is) || (i > pv.ie)) + ((j < pv.js) || (j > pv.je)) + ((k < pv.ks) || (k > pv.ke));
  if (ghost > 1) ok = false;
  if (ok && ghost == 1 && face_nbr) {  // a zone the stages do not read (they follow the face table to that neighbour's interior)
    const int f = (i < pv.is) ? 0 : (i > pv.ie) ? 1 : (j < pv.js) ? 2 : (j > pv.je) ? 3 : (k < pv.ks) ? 4 : 5;
    if (face_nbr[6 * b + f] >= 0) ok = false;
  }
  if (!WITH_DT && !ok) return;
  double lane_min = 1.7976931348623157e308;
  if (ok) {
    const apk_block_desc blk = pv.blocks[b];
    double w[nvars<FLUID>()];
    cons_to_prim_at_w<FLUID>(pv, blk, eos, flags, k * pv.sk + j * pv.sj + i, w);
    if constexpr (WITH_DT) {
      if (ghost == 0) lane_min = cell_dt_hyp<FLUID>(pv, blk, eos.gamma, w);
    }
  }
  if constexpr (WITH_DT) block_min_to_word(lane_min, dt_bits);
}

// The end of a cycle of a refined mesh whose stage loop stores no primitives (apk_tag_blocks_begin_from_cons): the
// pressure-gradient criterion (refinement/gradient.cpp:18-61) and the time-step estimate (hydro.cpp:845-895) in ONE pass
// over the conserved state.  A workgroup takes the planes [k0, k1] of the criterion's range [ks - 1, ke + 1] of one block:
//   1  the pressure of the planes k0 - 1 .. k1 + 1 over [js - 2, je + 2] x [is - 2, ie + 2] into LDS, each cell by the
//      lean ConsToPrim -- a ghost cell straight behind a face whose face_nbr entry is >= 0 from that neighbour's interior
//      (the zones the exchange in front of the check leaves out, as apk_tag_blocks_begin_skip reads them) -- and, for the
//      interior cells of its own planes, the time-step estimate from the primitives it has in registers;
//   2  the criterion of its cells from LDS, the expressions of tag_kernel.
// Against apk_cons_to_prim_dt_select(pressure) + apk_tag_blocks_begin_skip: no pressure array written and read back, one
// launch; the same maxima and the same minimum, bit for bit.
template <int FLUID>
__global__ void __launch_bounds__(256)
tag_pgrad_from_cons_kernel(PackView pv, apk_eos eos, unsigned *flags, unsigned long long *dt_bits, unsigned long long *block_max,
                           int kchunks, const int *face_nbr) {
  constexpr int NV = nvars<FLUID>();
  extern __shared__ __attribute__((aligned(16))) double ptile[];
  const int b = blockIdx.x / kchunks, chunk = blockIdx.x - b * kchunks;
  const apk_block_desc blk = pv.blocks[b];
  const int kl = pv.ks - 1, ku = pv.ke + 1;
  const int klen = (ku - kl + kchunks) / kchunks;
  const int k0 = kl + chunk * klen, k1 = (k0 + klen - 1 < ku) ? k0 + klen - 1 : ku;
  const int ti = pv.nx1 + 4, tj = pv.nx2 + 4, tk = k1 - k0 + 3;  // tile extents (k0 - 1 .. k1 + 1)
  const int tid = threadIdx.y * 64 + threadIdx.x;
  double lane_min = 1.7976931348623157e308, m = 0.0;
  if (k0 <= ku) {
    const int *fn = face_nbr ? face_nbr + 6 * b : nullptr;
    const int f0 = fn ? fn[0] : -1, f1 = fn ? fn[1] : -1, f2 = fn ? fn[2] : -1, f3 = fn ? fn[3] : -1, f4 = fn ? fn[4] : -1,
              f5 = fn ? fn[5] : -1;
    const double gm1 = eos.gamma - 1.0, vceil_sq = eos.vceil * eos.vceil, pf = eos.pfloor / gm1;
    for (int t = tid; t < ti * tj * tk; t += 256) {
      const int kk = t / (ti * tj), r = t - kk * (ti * tj), jj = r / ti, ii = r - jj * ti;
      int i = pv.is - 2 + ii, j = pv.js - 2 + jj, k = k0 - 1 + kk;
      const int gi = (i < pv.is) ? 1 : ((i > pv.ie) ? 2 : 0), gj = (j < pv.js) ? 1 : ((j > pv.je) ? 2 : 0),
                gk = (k < pv.ks) ? 1 : ((k > pv.ke) ? 2 : 0);
      const double *base = blk.cons;
      if ((gi != 0) + (gj != 0) + (gk != 0) == 1) {
        const int nb = gi ? (gi == 1 ? f0 : f1) : (gj ? (gj == 1 ? f2 : f3) : (gk == 1 ? f4 : f5));
        if (nb >= 0) {
          base = pv.blocks[nb].cons;
          if (gi) i += (gi == 1) ? pv.nx1 : -pv.nx1;
          if (gj) j += (gj == 1) ? pv.nx2 : -pv.nx2;
          if (gk) k += (gk == 1) ? pv.nx3 : -pv.nx3;
        }
      }
      const int64_t cell = (int64_t)k * pv.sk + (int64_t)j * pv.sj + i;
      double u[NV], w[NV], di;
#pragma unroll
      for (int n = 0; n < NV; ++n) u[n] = base[n * pv.sn + cell];
      const unsigned fl = cons_to_prim_core<FLUID, 1>(eos, gm1, vceil_sq, pf, u, w, di);
      if (fl) atomicOr(flags, fl);
      ptile[t] = w[IPR];
      // (the estimate: interior cells of the chunk's own planes -- every interior cell belongs to exactly one chunk)
      if (gi == 0 && gj == 0 && gk == 0 && kk >= 1 && kk <= tk - 2) lane_min = fmin(lane_min, cell_dt_hyp<FLUID>(pv, blk, eos.gamma, w));
    }
  }
  __syncthreads();
  if (k0 <= ku) {
    const int ci = pv.nx1 + 2, cj = pv.nx2 + 2, ck = k1 - k0 + 1;  // the criterion's cells: [s - 1, e + 1]
    for (int t = tid; t < ci * cj * ck; t += 256) {
      const int kk = t / (ci * cj), r = t - kk * (ci * cj), jj = r / ci, ii = r - jj * ci;
      const double *p = ptile + ((kk + 1) * tj + (jj + 1)) * ti + (ii + 1);
      const double a = 0.5 * (p[1] - p[-1]), bb = 0.5 * (p[ti] - p[-ti]), cc = 0.5 * (p[ti * tj] - p[-ti * tj]);
      const double eps = sqrt(sqr(a) + sqr(bb) + sqr(cc)) / p[0];
      m = fmax(m, eps);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
  __shared__ double part[4];
  if ((tid & 63) == 0) part[tid >> 6] = m;
  __syncthreads();
  if (tid == 0 && k0 <= ku) {
    const double mm = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
    atomicMax(block_max + b, (unsigned long long)__double_as_longlong(mm));
  }
  block_min_to_word(lane_min, dt_bits);
}

// Ghost zones only (the interior was converted by the finishing sweep of the fused stage).  The
// ghost shell of a block is enumerated as three groups of slabs so that no thread is launched
// for an interior cell: x3 slabs (whole planes), x2 slabs of the interior planes (whole rows),
// x1 slabs of the interior rows (2 ng cells per row).
template <int FLUID>
__global__ void __launch_bounds__(256)
cons_to_prim_ghosts_kernel(PackView pv, apk_eos eos, unsigned *flags, int64_t na, int64_t nb_, int64_t nc,
                           const unsigned *late_regions, int part) {
  const int b = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int gk = pv.nk - pv.nx3, gj = pv.nj - pv.nx2, gi = pv.ni - pv.nx1;  // ghost layers (both sides)
  int i, j, k;
  if (t < na) {  // x3 slabs: (kk, j, i), i fastest
    const int64_t plane = (int64_t)pv.nj * pv.ni;
    const int kk = (int)(t / plane);
    const int64_t r = t - kk * plane;
    j = (int)(r / pv.ni);
    i = (int)(r - (int64_t)j * pv.ni);
    k = (kk < gk / 2) ? kk : pv.ke + 1 + (kk - gk / 2);
  } else if (t < na + nb_) {  // x2 slabs of interior planes: (k, jj, i)
    const int64_t q = t - na;
    const int64_t per = (int64_t)gj * pv.ni;
    k = pv.ks + (int)(q / per);
    const int64_t r = q - (int64_t)(k - pv.ks) * per;
    const int jj = (int)(r / pv.ni);
    i = (int)(r - (int64_t)jj * pv.ni);
    j = (jj < gj / 2) ? jj : pv.je + 1 + (jj - gj / 2);
  } else if (t < na + nb_ + nc) {  // x1 slabs of interior rows: (k, j, ii)
    const int64_t q = t - na - nb_;
    const int64_t per = (int64_t)pv.nx2 * gi;
    k = pv.ks + (int)(q / per);
    const int64_t r = q - (int64_t)(k - pv.ks) * per;
    j = pv.js + (int)(r / gi);
    const int ii = (int)(r - (int64_t)(j - pv.js) * gi);
    i = (ii < gi / 2) ? ii : pv.ie + 1 + (ii - gi / 2);
  } else {
    return;
  }
  if (part != 0) {
    // split around a halo exchange in flight: bit (sx+1) + 3 (sy+1) + 9 (sz+1) of late_regions[b]
    // marks the neighbour region at offset (sx, sy, sz) as filled only when the exchange
    // completes; part 1 converts the ghost cells of the other regions, part 2 those
    const int sx = (i < pv.is) ? 0 : ((i > pv.ie) ? 2 : 1);
    const int sy = (j < pv.js) ? 0 : ((j > pv.je) ? 2 : 1);
    const int sz = (k < pv.ks) ? 0 : ((k > pv.ke) ? 2 : 1);
    const bool late = (late_regions[b] >> (sx + 3 * sy + 9 * sz)) & 1u;
    if (late != (part == 2)) return;
  }
  cons_to_prim_at<FLUID>(pv, pv.blocks[b], eos, flags, k * pv.sk + j * pv.sj + i);
}

// ---- EstimateHyperbolicTimestep (src/hydro/hydro.cpp:828-896) ----------------------------
// positive doubles order like their bit patterns, so the global min is one 64-bit atomicMin
template <int FLUID>
__global__ void __launch_bounds__(256)
min_dt_kernel(PackView pv, double gamma, unsigned long long *min_bits, int kchunks) {
  // grid = rect_grid(nx1, nx2, nblocks*kchunks): every workgroup walks a slab of k planes of its
  // (i,j) tile, so the pack costs ~2k atomics instead of one per plane tile
  CellIdx c;
  int io, jo;
  const bool inside = rect_ij(pv.nx1, pv.nx2, io, jo);
  c.i = pv.is + io;
  c.j = pv.js + jo;
  c.b = blockIdx.z / kchunks;
  const int chunk = blockIdx.z % kchunks;
  const int klen = (pv.nx3 + kchunks - 1) / kchunks;
  const int k0 = pv.ks + chunk * klen;
  const int k1 = (k0 + klen - 1 < pv.ke) ? k0 + klen - 1 : pv.ke;
  c.ok = inside;
  double min_dt = 1.7976931348623157e308;
  const apk_block_desc blk = pv.blocks[c.b];
  for (c.k = k0; c.ok && c.k <= k1; ++c.k) {
    const double *w = blk.prim + c.k * pv.sk + c.j * pv.sj + c.i;
    const double d = w[IDN * pv.sn], v1 = w[IV1 * pv.sn], v2 = w[IV2 * pv.sn],
                 v3 = w[IV3 * pv.sn], p = w[IPR * pv.sn];
    double lx, ly = 0.0, lz = 0.0;
    if constexpr (FLUID == APK_FLUID_EULER) {
      lx = sound_speed(gamma, d, p);
      ly = lx;
      lz = lx;
    } else {
      const double b1 = w[IB1 * pv.sn], b2 = w[IB2 * pv.sn], b3 = w[IB3 * pv.sn];
      lx = fast_speed(gamma, d, p, b1, b2, b3);
      if (pv.ndim > 1) ly = fast_speed(gamma, d, p, b2, b3, b1);
      if (pv.ndim > 2) lz = fast_speed(gamma, d, p, b3, b1, b2);
    }
    min_dt = fmin(min_dt, blk.dx[0] / (fabs(v1) + lx));
    if (pv.ndim > 1) min_dt = fmin(min_dt, blk.dx[1] / (fabs(v2) + ly));
    if (pv.ndim > 2) min_dt = fmin(min_dt, blk.dx[2] / (fabs(v3) + lz));
  }
  min_dt = wave_min(min_dt);
  __shared__ double part[4];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  if ((tid & 63) == 0) part[tid >> 6] = min_dt;
  __syncthreads();
  if (tid == 0) {
    const double m = fmin(fmin(part[0], part[1]), fmin(part[2], part[3]));
    atomicMin(min_bits, (unsigned long long)__double_as_longlong(m));
  }
}

// ---- HydroHst (src/hydro/hydro.cpp:145-208): per-workgroup partials, fixed-order final ----
template <int FLUID>
__global__ void __launch_bounds__(256) history_kernel(PackView pv, double *partial) {
  const CellIdx c = interior_cell(pv);
  double h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c.ok) {
    const apk_block_desc blk = pv.blocks[c.b];
    const double vol = blk.dx[0] * blk.dx[1] * blk.dx[2];
    const double *u = blk.cons + c.k * pv.sk + c.j * pv.sj + c.i;
    const double d = u[IDN * pv.sn], m1 = u[IM1 * pv.sn], m2 = u[IM2 * pv.sn],
                 m3 = u[IM3 * pv.sn];
    h[0] = d * vol;
    h[1] = m1 * vol;
    h[2] = m2 * vol;
    h[3] = m3 * vol;
    h[4] = 0.5 / d * (sqr(m1) + sqr(m2) + sqr(m3)) * vol;
    h[5] = u[IEN * pv.sn] * vol;
    if constexpr (FLUID == APK_FLUID_GLMMHD) {
      const double *b1 = u + IB1 * pv.sn, *b2 = u + IB2 * pv.sn, *b3 = u + IB3 * pv.sn;
      h[6] = 0.5 * (sqr(b1[0]) + sqr(b2[0]) + sqr(b3[0])) * vol;
      const int64_t so = (pv.ndim >= 2) ? pv.sj : 0;
      double divb = (b1[1] - b1[-1]) / blk.dx[0] + (b2[so] - b2[-so]) / blk.dx[1];
      if (pv.ndim == 3) divb += (b3[pv.sk] - b3[-pv.sk]) / blk.dx[2];
      const double abs_b = sqrt(sqr(b1[0]) + sqr(b2[0]) + sqr(b3[0]));
      h[7] = (abs_b != 0)
                 ? 0.5 * (sqrt(sqr(blk.dx[0]) + sqr(blk.dx[1]) + sqr(blk.dx[2]))) * fabs(divb) /
                       abs_b * vol
                 : 0;
    }
  }
  __shared__ double part[4][8];
  const int tid = threadIdx.y * 64 + threadIdx.x;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const double s = wave_sum(h[q]);
    if ((tid & 63) == 0) part[tid >> 6][q] = s;
  }
  __syncthreads();
  if (tid < 8) {
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[(int64_t)wg * 8 + tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
  }
}

__global__ void __launch_bounds__(256) history_final_kernel(const double *partial, int nwg,
                                                            double *out8) {
  // 8 quantities x 32 lanes each; fixed summation order => run-to-run deterministic
  const int q = threadIdx.x / 32, lane = threadIdx.x % 32;
  double s = 0.0;
  for (int w = lane; w < nwg; w += 32) s += partial[(int64_t)w * 8 + q];
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 32);
  if (lane == 0) out8[q] = s;
}

// ---- FirstOrderFluxCorrect (src/hydro/hydro.cpp:1223-1342), two-phase schedule -------------
template <int FLUID>
__global__ void __launch_bounds__(256)
fofc_mark_kernel(PackView u0, PackView u1, double gam0, double gam1, double beta_dt,
                 int attempt, unsigned char *mark, unsigned long long *count) {
  constexpr int NV = nvars<FLUID>();
  const CellIdx c = interior_cell(u0);
  bool bad = false;
  if (c.ok) {
    const apk_block_desc b0 = u0.blocks[c.b];
    const double *c1 = u1.blocks[c.b].cons;
    double area[3], vol;
    block_areas(b0, area, vol);
    const int64_t cell = c.k * u0.sk + c.j * u0.sj + c.i;
    double nc[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const int64_t idx = n * u0.sn + cell;
      const double old = (gam0 != 0.0) ? b0.cons[idx] : 0.0;
      nc[n] = gam0 * old + gam1 * c1[idx] + beta_dt * flux_div(u0, b0, idx, area, vol);
    }
    double new_p = nc[IEN] - 0.5 * (sqr(nc[IM1]) + sqr(nc[IM2]) + sqr(nc[IM3])) / nc[IDN];
    if constexpr (FLUID == APK_FLUID_GLMMHD)
      new_p -= 0.5 * (sqr(nc[IB1]) + sqr(nc[IB2]) + sqr(nc[IB3]));
    bad = !(nc[IDN] > 0.0 && new_p > 0.0);
    if (bad && attempt > 2 && nc[IDN] > 0.0 && new_p < 0.0) bad = false;  // rely on the floor
    mark[(int64_t)c.b * u0.sn + cell] = bad ? 1 : 0;
  }
  const unsigned long long m = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}

// The same test on a state that has been updated already (the fused stage): rho <= 0 or
// E - KE [- ME] <= 0 (hydro.cpp:1297-1306; no gamma - 1, only the sign matters)
template <int FLUID>
__global__ void __launch_bounds__(256) count_unphysical_kernel(PackView u0, unsigned long long *count) {
  const CellIdx c = interior_cell(u0);
  bool bad = false;
  if (c.ok) {
    const double *u = u0.blocks[c.b].cons + c.k * u0.sk + c.j * u0.sj + c.i;
    const double d = u[IDN * u0.sn];
    double new_p = u[IEN * u0.sn] - 0.5 * (sqr(u[IM1 * u0.sn]) + sqr(u[IM2 * u0.sn]) + sqr(u[IM3 * u0.sn])) / d;
    if constexpr (FLUID == APK_FLUID_GLMMHD)
      new_p -= 0.5 * (sqr(u[IB1 * u0.sn]) + sqr(u[IB2 * u0.sn]) + sqr(u[IB3 * u0.sn]));
    bad = !(d > 0.0 && new_p > 0.0);
  }
  const unsigned long long m = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}

// DC + LLF flux of one face straight from prim (hydro_dc_llf.hpp:43-142,
// glmmhd_dc_llf.hpp:46-179), including passive scalars
template <int FLUID, int DIR>
APK_DEV void llf_face(const PackView &pv, const apk_block_desc &blk, int64_t cell, double gamma,
                      double c_h) {
  constexpr int NV = nvars<FLUID>();
  const int64_t st = (DIR == 1) ? 1 : ((DIR == 2) ? pv.sj : pv.sk);
  const double *p = blk.prim + cell;
  double wl[NV], wr[NV], f[NV];
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    wl[s] = p[perm<DIR>(s) * pv.sn - st];
    wr[s] = p[perm<DIR>(s) * pv.sn];
  }
  riemann<FLUID, APK_RS_LLF>(wl, wr, gamma, c_h, f);
  double *fo = blk.flux[DIR - 1] + cell;
#pragma unroll
  for (int s = 0; s < NV; ++s) fo[perm<DIR>(s) * pv.sn] = f[s];
  for (int n = NV; n < pv.nvar; ++n)
    fo[n * pv.sn] = (f[IDN] >= 0.0) ? f[IDN] * p[n * pv.sn - st] : f[IDN] * p[n * pv.sn];
}

template <int FLUID>
__global__ void __launch_bounds__(256)
fofc_fix_kernel(PackView pv, double gamma, double c_h, const unsigned char *mark) {
  const CellIdx c = interior_cell(pv);
  if (!c.ok) return;
  const int64_t cell = c.k * pv.sk + c.j * pv.sj + c.i;
  if (!mark[(int64_t)c.b * pv.sn + cell]) return;
  const apk_block_desc blk = pv.blocks[c.b];
  llf_face<FLUID, 1>(pv, blk, cell, gamma, c_h);
  llf_face<FLUID, 1>(pv, blk, cell + 1, gamma, c_h);
  if (pv.ndim >= 2) {
    llf_face<FLUID, 2>(pv, blk, cell, gamma, c_h);
    llf_face<FLUID, 2>(pv, blk, cell + pv.sj, gamma, c_h);
  }
  if (pv.ndim >= 3) {
    llf_face<FLUID, 3>(pv, blk, cell, gamma, c_h);
    llf_face<FLUID, 3>(pv, blk, cell + pv.sk, gamma, c_h);
  }
}

// ---- strided box copies (ghost exchange / message packing / physical boundaries) -----------
// One workgroup per chunk of the plan's work list (apk_copy_chunk): at most kCopyChunkItems
// (cell, variable) items of one box, cells fastest.
__global__ void __launch_bounds__(256)
copy_regions_kernel(const apk_copy_region *regions, const apk_copy_chunk *chunks) {
  const apk_copy_chunk ch = chunks[blockIdx.x];
  const apk_copy_region r = regions[ch.region];
  const int plane = r.ext[0] * r.ext[1];
  const int cells = plane * r.ext[2], items = cells * r.nvar;
  const int end = (ch.first + kCopyChunkItems < items) ? ch.first + kCopyChunkItems : items;
  for (int t = ch.first + (int)threadIdx.x; t < end; t += 256) {
    const int v = t / cells;
    const int c = t - v * cells;
    const int k = c / plane;
    const int rem = c - k * plane;
    const int j = rem / r.ext[0];
    const int i = rem - j * r.ext[0];
    const double x = r.src[i * r.src_stride[0] + j * r.src_stride[1] + k * r.src_stride[2] + v * r.src_stride[3]];
    r.dst[i * r.dst_stride[0] + j * r.dst_stride[1] + k * r.dst_stride[2] + v * r.dst_stride[3]] = (v == r.flip_var) ? -x : x;
  }
}

// The same copy, one thread per CELL of a chunk of at most kCopyChunkCells cells, moving all of its variables: the
// index arithmetic (three integer divisions by run-time extents) is paid once per cell instead of once per value --
// 134 VALU instructions per 8 bytes made the per-item form compute-bound on the small boxes of a refined mesh
// (refined mesh of 232 16^3 blocks: 0.893 -> 0.885 ms per cycle).  APK_COPY_PER_ITEM=1: the per-item form (A/B).
__global__ void __launch_bounds__(256)
copy_regions_cells_kernel(const apk_copy_region *regions, const apk_copy_chunk *chunks) {
  const apk_copy_chunk ch = chunks[blockIdx.x];
  const apk_copy_region r = regions[ch.region];
  const int plane = r.ext[0] * r.ext[1];
  const int cells = plane * r.ext[2];
  const int t = ch.first + (int)threadIdx.x;
  if (t >= cells) return;
  const int k = t / plane;
  const int rem = t - k * plane;
  const int j = rem / r.ext[0];
  const int i = rem - j * r.ext[0];
  const double *src = r.src + (i * r.src_stride[0] + j * r.src_stride[1] + k * r.src_stride[2]);
  double *dst = r.dst + (i * r.dst_stride[0] + j * r.dst_stride[1] + k * r.dst_stride[2]);
  // (all loads of a batch are issued before its first store: with the variable count a run-time number the plain
  // loop waited for every value in turn -- nine memory round trips per cell; message packing of a 128^3 brick's
  // outer faces: 182 -> see DESIGN.md section 6)
  constexpr int kBatch = 12;
  for (int v0 = 0; v0 < r.nvar; v0 += kBatch) {
    double x[kBatch];
#pragma unroll
    for (int q = 0; q < kBatch; ++q)
      if (v0 + q < r.nvar) x[q] = src[(v0 + q) * r.src_stride[3]];
#pragma unroll
    for (int q = 0; q < kBatch; ++q)
      if (v0 + q < r.nvar) dst[(v0 + q) * r.dst_stride[3]] = (v0 + q == r.flip_var) ? -x[q] : x[q];
  }
}

// The same copy with ConservedToPrimitive of every destination cell fused in (ghost-zone fills:
// same-rank copies, message unpacking, physical boundaries): the thread that moves the nvar values
// of a cell already holds its conserved state, so the primitives go out with it -- prim lives at
// dst + prim_delta -- and the separate ghost ConsToPrim pass (one more read of cons) disappears.
// Only used when no floor / ceiling is active (they would write cons back, and a later boundary
// phase would read the floored instead of the copied value; the unfused order is kept for that).
// One workgroup per chunk of at most kCopyChunkCells cells.
// STORE_CONS = false: the primitives only (ghost zones whose conserved values nobody reads: the half-step state of
// VL2, apk_copy_plan_run_c2p_prim_only) -- half the strided stores.
template <int FLUID, bool STORE_CONS>
__global__ void __launch_bounds__(256)
copy_regions_c2p_kernel(const apk_copy_region *regions, const apk_copy_chunk *chunks, apk_eos eos, unsigned *flags,
                        int64_t prim_delta) {
  constexpr int NV = nvars<FLUID>();
  const apk_copy_chunk ch = chunks[blockIdx.x];
  const apk_copy_region r = regions[ch.region];
  const int plane = r.ext[0] * r.ext[1];
  const int cells = plane * r.ext[2];
  const int t = ch.first + (int)threadIdx.x;
  if (t < cells) {
    const int k = t / plane;
    const int rem = t - k * plane;
    const int j = rem / r.ext[0];
    const int i = rem - j * r.ext[0];
    const int64_t so = i * r.src_stride[0] + j * r.src_stride[1] + k * r.src_stride[2];
    const int64_t dof = i * r.dst_stride[0] + j * r.dst_stride[1] + k * r.dst_stride[2];
    double u[NV], w[NV], di;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const double x = r.src[so + v * r.src_stride[3]];
      u[v] = (v == r.flip_var) ? -x : x;
      if constexpr (STORE_CONS) r.dst[dof + v * r.dst_stride[3]] = u[v];
    }
    const unsigned fl = cons_to_prim_cell<FLUID>(eos, u, w, di);
    if (fl && flags) atomicOr(flags, fl);
    double *p = r.dst + prim_delta + dof;
#pragma unroll
    for (int v = 0; v < NV; ++v) p[v * r.dst_stride[3]] = w[v];
    for (int v = NV; v < r.nvar; ++v) {  // passive scalars
      const double x = r.src[so + v * r.src_stride[3]];
      if constexpr (STORE_CONS) r.dst[dof + v * r.dst_stride[3]] = x;
      p[v * r.dst_stride[3]] = x * di;
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
int launch_update_flux_div(const PackView &u0, const PackView &u1, double gam0, double gam1,
                           double beta_dt, hipStream_t s) {
  hipLaunchKernelGGL(update_flux_div_kernel, interior_grid(u0), dim3(64, 4, 1), 0, s, u0, u1,
                     gam0, gam1, beta_dt);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

int launch_dedner(const PackView &pv, int extended, double coeff, double beta_dt,
                  hipStream_t s) {
  if (extended)
    hipLaunchKernelGGL(dedner_kernel<true>, interior_grid(pv), dim3(64, 4, 1), 0, s, pv, coeff,
                       beta_dt);
  else
    hipLaunchKernelGGL(dedner_kernel<false>, interior_grid(pv), dim3(64, 4, 1), 0, s, pv, coeff,
                       beta_dt);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

int launch_cons_to_prim(const PackView &pv, int fluid, const apk_eos &eos, unsigned *d_flags,
                        hipStream_t s, bool ghosts_only, const unsigned *late_regions, int part, bool faces_only,
                        const int *face_nbr, unsigned long long *dt_bits, int depth, unsigned store_vars) {
  if (ghosts_only) {
    const int64_t na = (int64_t)(pv.nk - pv.nx3) * pv.nj * pv.ni;
    const int64_t nb = (int64_t)pv.nx3 * (pv.nj - pv.nx2) * pv.ni;
    const int64_t nc = (int64_t)pv.nx3 * pv.nx2 * (pv.ni - pv.nx1);
    const dim3 grid((unsigned)((na + nb + nc + 255) / 256), pv.nblocks, 1);
    if (fluid == APK_FLUID_EULER)
      hipLaunchKernelGGL(cons_to_prim_ghosts_kernel<APK_FLUID_EULER>, grid, dim3(256), 0, s, pv, eos, d_flags, na, nb, nc,
                         late_regions, part);
    else
      hipLaunchKernelGGL(cons_to_prim_ghosts_kernel<APK_FLUID_GLMMHD>, grid, dim3(256), 0, s, pv, eos, d_flags, na, nb, nc,
                         late_regions, part);
    return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
  }
  dim3 grid = rect_grid(pv.ni, pv.nj, pv.nk * pv.nblocks);
  if (depth >= pv.ng && store_vars == ~0u) depth = -1;  // (the whole block)
  if (depth > pv.ng) depth = pv.ng;               // (a subset of the primitives: the box that is the whole block)
  if (store_vars != ~0u && (faces_only || !dt_bits || depth < 0)) return APK_ERR_UNSUPPORTED;  // (the boxed pass with the estimate)
  if (!faces_only && dt_bits && depth >= 0) {  // (cons_to_prim_kernel's box)
    const int64_t cells = (int64_t)(pv.nx1 + 2 * depth) * (pv.ndim >= 2 ? pv.nx2 + 2 * depth : pv.nj) * (pv.ndim >= 3 ? pv.nx3 + 2 * depth : pv.nk);
    const int64_t wgs = (cells + 255) / 256, gy = (wgs + 32767) / 32768, gx = (wgs + gy - 1) / gy;
    if (cells < ((int64_t)1 << 31) && gy <= 65535) grid = dim3((unsigned)gx, (unsigned)gy, pv.nblocks);
    else return APK_ERR_UNSUPPORTED;  // (never: a block of 2^31 cells; widening the region would convert stale ghost cells)
  }
  if (faces_only) {
    const bool euler = fluid == APK_FLUID_EULER;
    if (euler && dt_bits)
      hipLaunchKernelGGL((cons_to_prim_faces_kernel<APK_FLUID_EULER, true>), grid, dim3(64, 4, 1), 0, s, pv, eos, d_flags, face_nbr, dt_bits);
    else if (euler)
      hipLaunchKernelGGL((cons_to_prim_faces_kernel<APK_FLUID_EULER, false>), grid, dim3(64, 4, 1), 0, s, pv, eos, d_flags, face_nbr, dt_bits);
    else if (dt_bits)
      hipLaunchKernelGGL((cons_to_prim_faces_kernel<APK_FLUID_GLMMHD, true>), grid, dim3(64, 4, 1), 0, s, pv, eos, d_flags, face_nbr, dt_bits);
    else
      hipLaunchKernelGGL((cons_to_prim_faces_kernel<APK_FLUID_GLMMHD, false>), grid, dim3(64, 4, 1), 0, s, pv, eos, d_flags, face_nbr, dt_bits);
  } else if (dt_bits) {
    if (fluid == APK_FLUID_EULER)
      hipLaunchKernelGGL((cons_to_prim_kernel<APK_FLUID_EULER, true>), grid, dim3(64, 4, 1), 0, s, pv, eos, d_flags, dt_bits, depth, face_nbr, store_vars);
    else
      hipLaunchKernelGGL((cons_to_prim_kernel<APK_FLUID_GLMMHD, true>), grid, dim3(64, 4, 1), 0, s, pv, eos, d_flags, dt_bits, depth, face_nbr, store_vars);
  } else if (fluid == APK_FLUID_EULER)
    hipLaunchKernelGGL((cons_to_prim_kernel<APK_FLUID_EULER, false>), grid, dim3(64, 4, 1), 0, s, pv, eos, d_flags, dt_bits, -1, nullptr, ~0u);
  else
    hipLaunchKernelGGL((cons_to_prim_kernel<APK_FLUID_GLMMHD, false>), grid, dim3(64, 4, 1), 0, s, pv, eos, d_flags, dt_bits, -1, nullptr, ~0u);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

// planes per workgroup so that the pressure tile fits the LDS a workgroup may take; 0: the blocks are too wide
int tag_from_cons_kchunks(const PackView &pv) {
  const size_t plane = sizeof(double) * (size_t)(pv.nx1 + 4) * (size_t)(pv.nx2 + 4);
  const int budget = 48 * 1024;
  const int planes = (int)(budget / plane);  // tile planes a workgroup can hold: its own and one either side
  if (pv.ndim != 3 || planes < 3 || pv.ng < 2) return 0;
  const int own = planes - 2, range = pv.nx3 + 2;
  int kchunks = (range + own - 1) / own;
  if (kchunks < 3 && pv.nx3 >= 12) kchunks = 3;  // (as the tag kernel: a pack of a few hundred narrow blocks is too few workgroups)
  return kchunks;
}

int launch_tag_pgrad_from_cons(const PackView &pv, int fluid, const apk_eos &eos, unsigned *d_flags, unsigned long long *dt_bits,
                               unsigned long long *block_max, const int *face_nbr, hipStream_t s) {
  const int kchunks = tag_from_cons_kchunks(pv);
  if (kchunks <= 0) return APK_ERR_UNSUPPORTED;
  const int range = pv.nx3 + 2, klen = (range - 1 + kchunks) / kchunks;
  const size_t lds = sizeof(double) * (size_t)(pv.nx1 + 4) * (size_t)(pv.nx2 + 4) * (size_t)(klen + 2);
  const dim3 grid((unsigned)(pv.nblocks * kchunks), 1, 1), block(64, 4, 1);
  if (fluid == APK_FLUID_EULER)
    hipLaunchKernelGGL(tag_pgrad_from_cons_kernel<APK_FLUID_EULER>, grid, block, lds, s, pv, eos, d_flags, dt_bits, block_max, kchunks, face_nbr);
  else
    hipLaunchKernelGGL(tag_pgrad_from_cons_kernel<APK_FLUID_GLMMHD>, grid, block, lds, s, pv, eos, d_flags, dt_bits, block_max, kchunks, face_nbr);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

int launch_min_dt(const PackView &pv, int fluid, double gamma, unsigned long long *d_min_bits,
                  hipStream_t s) {
  // (narrow meshblocks come in large numbers: one slab per block keeps the atomics on the one
  // result word few)
  const int kchunks = pv.nx1 < 48 ? 1 : (pv.nx3 >= 8 ? 8 : pv.nx3);
  const dim3 grid = rect_grid(pv.nx1, pv.nx2, pv.nblocks * kchunks);
  if (fluid == APK_FLUID_EULER)
    hipLaunchKernelGGL(min_dt_kernel<APK_FLUID_EULER>, grid, dim3(64, 4, 1), 0, s, pv, gamma,
                       d_min_bits, kchunks);
  else
    hipLaunchKernelGGL(min_dt_kernel<APK_FLUID_GLMMHD>, grid, dim3(64, 4, 1), 0, s, pv, gamma,
                       d_min_bits, kchunks);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

int launch_history(const PackView &pv, int fluid, double *d_partial, int *nwg_out, double *d_out8,
                   hipStream_t s) {
  const dim3 grid = interior_grid(pv);
  const int nwg = grid.x * grid.y * grid.z;
  if (nwg_out) *nwg_out = nwg;
  if (!d_partial) return APK_OK;  // size query
  if (fluid == APK_FLUID_EULER)
    hipLaunchKernelGGL(history_kernel<APK_FLUID_EULER>, grid, dim3(64, 4, 1), 0, s, pv, d_partial);
  else
    hipLaunchKernelGGL(history_kernel<APK_FLUID_GLMMHD>, grid, dim3(64, 4, 1), 0, s, pv,
                       d_partial);
  hipLaunchKernelGGL(history_final_kernel, dim3(1), dim3(256), 0, s, d_partial, nwg, d_out8);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

int launch_fofc_mark(const PackView &u0, const PackView &u1, int fluid, double gam0, double gam1,
                     double beta_dt, int attempt, unsigned char *d_mark,
                     unsigned long long *d_count, hipStream_t s) {
  if (fluid == APK_FLUID_EULER)
    hipLaunchKernelGGL(fofc_mark_kernel<APK_FLUID_EULER>, interior_grid(u0), dim3(64, 4, 1), 0, s,
                       u0, u1, gam0, gam1, beta_dt, attempt, d_mark, d_count);
  else
    hipLaunchKernelGGL(fofc_mark_kernel<APK_FLUID_GLMMHD>, interior_grid(u0), dim3(64, 4, 1), 0, s,
                       u0, u1, gam0, gam1, beta_dt, attempt, d_mark, d_count);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

int launch_count_unphysical(const PackView &u0, int fluid, unsigned long long *d_count, hipStream_t s) {
  if (fluid == APK_FLUID_EULER)
    hipLaunchKernelGGL(count_unphysical_kernel<APK_FLUID_EULER>, interior_grid(u0), dim3(64, 4, 1), 0, s, u0, d_count);
  else
    hipLaunchKernelGGL(count_unphysical_kernel<APK_FLUID_GLMMHD>, interior_grid(u0), dim3(64, 4, 1), 0, s, u0, d_count);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

int launch_fofc_fix(const PackView &u0, int fluid, double gamma, double c_h,
                    const unsigned char *d_mark, hipStream_t s) {
  if (fluid == APK_FLUID_EULER)
    hipLaunchKernelGGL(fofc_fix_kernel<APK_FLUID_EULER>, interior_grid(u0), dim3(64, 4, 1), 0, s,
                       u0, gamma, c_h, d_mark);
  else
    hipLaunchKernelGGL(fofc_fix_kernel<APK_FLUID_GLMMHD>, interior_grid(u0), dim3(64, 4, 1), 0, s,
                       u0, gamma, c_h, d_mark);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

int launch_copy_regions(const apk_copy_plan &plan, hipStream_t s, int c2p_fluid, const apk_eos *eos, unsigned *d_flags,
                        int64_t prim_delta, bool prim_only) {
  if (plan.n <= 0) return APK_OK;
  // the plain copy works per (cell, variable), the ConsToPrim variants per cell
  if (c2p_fluid == APK_FLUID_EULER || c2p_fluid == APK_FLUID_GLMMHD) {
    if (plan.nchunks_cells > 0) {
      const dim3 grid(plan.nchunks_cells), block(256);
#define APK_LAUNCH_COPY_C2P(FL, SC) \
  hipLaunchKernelGGL((copy_regions_c2p_kernel<FL, SC>), grid, block, 0, s, plan.d_regions, plan.d_chunks_cells, *eos, d_flags, prim_delta)
      if (c2p_fluid == APK_FLUID_EULER) {
        if (prim_only) APK_LAUNCH_COPY_C2P(APK_FLUID_EULER, false);
        else APK_LAUNCH_COPY_C2P(APK_FLUID_EULER, true);
      } else {
        if (prim_only) APK_LAUNCH_COPY_C2P(APK_FLUID_GLMMHD, false);
        else APK_LAUNCH_COPY_C2P(APK_FLUID_GLMMHD, true);
      }
#undef APK_LAUNCH_COPY_C2P
    }
  } else {
    if (plan.nchunks_cells > 0)
      hipLaunchKernelGGL(copy_regions_cells_kernel, dim3(plan.nchunks_cells), dim3(256), 0, s, plan.d_regions, plan.d_chunks_cells);
  }
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

}  // namespace apk
