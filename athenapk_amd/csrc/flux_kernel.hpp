// flux_kernel.hpp -- "flux arrays" path: one kernel per sweep direction that reconstructs
// the L/R face states, solves the Riemann problem and stores flux(dir, v, k, j, i) over the
// SAME index ranges the reference writes (src/hydro/hydro.cpp:1025-1208).  This is the
// path behind apk_calculate_fluxes(): it materialises the face fluxes in HBM, which the
// first-order flux correction and AMR flux correction consume.  The bandwidth-lean fused
// path is in kernels_fused.hip.
//
// Mapping: one lane per face, 64 consecutive lanes along x1 so every stencil load is a
// coalesced 512-B row segment (rows of narrow meshblocks are flattened, rect_ij); the 4..6
// stencil rows a lane touches along the sweep direction are shared with its neighbours through
// L1/L2.
#pragma once

#include "apk_internal.hpp"
#include "hydro_math.hpp"

namespace apk {

struct FluxExtent {
  int i0, i1, j0, j1, k0, k1;
};

// the reference's loop limits (hydro.cpp:1031-1039, 1106-1110, 1158)
inline FluxExtent flux_extent(const PackView &pv, int dir) {
  FluxExtent e;
  const bool d2 = pv.nx2 > 1, d3 = pv.nx3 > 1;
  if (dir == 1) {
    e.i0 = pv.is;
    e.i1 = pv.ie + 1;
    e.j0 = d2 ? pv.js - 1 : pv.js;
    e.j1 = d2 ? pv.je + 1 : pv.je;
    e.k0 = (d2 && d3) ? pv.ks - 1 : pv.ks;
    e.k1 = (d2 && d3) ? pv.ke + 1 : pv.ke;
  } else if (dir == 2) {
    e.i0 = pv.is - 1;
    e.i1 = pv.ie + 1;
    e.j0 = pv.js;
    e.j1 = pv.je + 1;
    e.k0 = d3 ? pv.ks - 1 : pv.ks;
    e.k1 = d3 ? pv.ke + 1 : pv.ke;
  } else {
    e.i0 = pv.is - 1;
    e.i1 = pv.ie + 1;
    e.j0 = pv.js - 1;
    e.j1 = pv.je + 1;
    e.k0 = pv.ks;
    e.k1 = pv.ke + 1;
  }
  return e;
}

// CalculateFluxesTight limits (hydro.cpp:1006-1009)
inline FluxExtent tight_extent(const PackView &pv, int dir) {
  FluxExtent e;
  e.i0 = pv.is;
  e.i1 = pv.ie + 1;
  e.j0 = pv.js;
  e.j1 = (pv.ndim >= 2) ? pv.je + 1 : pv.je;
  e.k0 = pv.ks;
  e.k1 = (pv.ndim >= 3) ? pv.ke + 1 : pv.ke;
  (void)dir;
  return e;
}

// L state of the face = ql of the lower cell, R state = qr of the upper cell.
template <int RECON>
APK_DEV void face_states(const double *c, int64_t st, double dx, int var, double &wl,
                         double &wr) {
  double dummy;
  if constexpr (RECON == APK_RC_DC) {
    wl = c[-st];
    wr = c[0];
  } else if constexpr (RECON == APK_RC_PPM || RECON == APK_RC_WENOZ) {
    const double qm3 = c[-3 * st], qm2 = c[-2 * st], qm1 = c[-st], q0 = c[0], qp1 = c[st],
                 qp2 = c[2 * st];
    reconstruct<RECON>(qm3, qm2, qm1, q0, qp1, dx, var, wl, dummy);
    reconstruct<RECON>(qm2, qm1, q0, qp1, qp2, dx, var, dummy, wr);
  } else {
    const double qm2 = c[-2 * st], qm1 = c[-st], q0 = c[0], qp1 = c[st];
    reconstruct<RECON>(0.0, qm2, qm1, q0, 0.0, dx, var, wl, dummy);
    reconstruct<RECON>(0.0, qm1, q0, qp1, 0.0, dx, var, dummy, wr);
  }
}

// Boundary planes from the CONSERVED state (apk_calculate_fluxes_boundary_list_from_cons: a refined mesh whose stages
// derive their input from the conserved state and store no primitives): the stencil cells are converted in registers by
// the lean ConsToPrim -- the function the pass over the block would have applied: same bits.  `delta`: from a block's
// cons array of the pack to the array that holds the input state (doubles; 0: the pack's own).
// (struct FluxConsInput: apk_internal.hpp)
template <int FLUID, int RECON>
APK_DEV void face_states_from_cons(const double *c, int64_t sn, int64_t st, double dx, const FluxConsInput &ci,
                                   double (&wl)[nvars<FLUID>()], double (&wr)[nvars<FLUID>()]) {
  constexpr int NV = nvars<FLUID>();
  constexpr int LO = (RECON == APK_RC_DC) ? -1 : ((RECON == APK_RC_PPM || RECON == APK_RC_WENOZ) ? -3 : -2);
  constexpr int HI = (RECON == APK_RC_DC) ? 0 : ((RECON == APK_RC_PPM || RECON == APK_RC_WENOZ) ? 2 : 1);
  double w[HI - LO + 1][NV];
#pragma unroll
  for (int m = LO; m <= HI; ++m) {
    double u[NV], di;
#pragma unroll
    for (int n = 0; n < NV; ++n) u[n] = c[n * sn + m * st];
    (void)cons_to_prim_core<FLUID, 1>(ci.eos, ci.eos_gm1, ci.vceil_sq, ci.pfloor_over_gm1, u, w[m - LO], di);
  }
  double dummy;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    if constexpr (RECON == APK_RC_DC) {
      wl[n] = w[0][n];
      wr[n] = w[1][n];
    } else if constexpr (RECON == APK_RC_PPM || RECON == APK_RC_WENOZ) {
      reconstruct<RECON>(w[0][n], w[1][n], w[2][n], w[3][n], w[4][n], dx, n, wl[n], dummy);
      reconstruct<RECON>(w[1][n], w[2][n], w[3][n], w[4][n], w[5][n], dx, n, dummy, wr[n]);
    } else {
      reconstruct<RECON>(0.0, w[0][n], w[1][n], w[2][n], 0.0, dx, n, wl[n], dummy);
      reconstruct<RECON>(0.0, w[1][n], w[2][n], w[3][n], 0.0, dx, n, dummy, wr[n]);
    }
  }
}

// the flux through the lower DIR-face of cell (k, j, i) of one block
template <int FLUID, int RECON, int RS, int DIR, bool FROM_CONS = false>
APK_DEV void flux_face(const PackView &pv, const apk_block_desc &blk, int i, int j, int k, double gamma, double c_h,
                       const FluxConsInput *ci = nullptr) {
  constexpr int NV = nvars<FLUID>();
  const int64_t st = (DIR == 1) ? 1 : ((DIR == 2) ? pv.sj : pv.sk);
  const int64_t cell = k * pv.sk + j * pv.sj + i;
  const double *p = blk.prim + cell;
  const double dx = blk.dx[DIR - 1];

  double wln[NV], wrn[NV];  // natural order
  if constexpr (FROM_CONS) {
    face_states_from_cons<FLUID, RECON>(blk.cons + ci->delta + cell, pv.sn, st, dx, *ci, wln, wrn);
  } else {
#pragma unroll
    for (int n = 0; n < NV; ++n) face_states<RECON>(p + n * pv.sn, st, dx, n, wln[n], wrn[n]);
  }

  double wl[NV], wr[NV], f[NV];  // direction-permuted order
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    wl[s] = wln[perm<DIR>(s)];
    wr[s] = wrn[perm<DIR>(s)];
  }
  riemann<FLUID, RS>(wl, wr, gamma, c_h, f);

  double *fo = blk.flux[DIR - 1] + cell;
#pragma unroll
  for (int s = 0; s < NV; ++s) fo[perm<DIR>(s) * pv.sn] = f[s];

  // passive scalars: upwind on the mass flux (hydro.cpp:1088-1097)
  for (int n = NV; n < pv.nvar; ++n) {
    double sl, sr;
    face_states<RECON>(p + n * pv.sn, st, dx, n, sl, sr);
    fo[n * pv.sn] = (f[IDN] >= 0.0) ? f[IDN] * sl : f[IDN] * sr;
  }
}

template <int FLUID, int RECON, int RS, int DIR>
__global__ void __launch_bounds__(256)
flux_kernel(PackView pv, FluxExtent e, double gamma, double c_h) {
  // the two longest axes of the index box share the workgroup (rect_ij), the third one the grid's z
  // with the block number: a boundary plane of direction 1 / 2 (one face along i / j) is spread
  // over (j, k) / (i, k) instead of leaving all but a few lanes idle
  const int nie = e.i1 - e.i0 + 1, nje = e.j1 - e.j0 + 1, nke = e.k1 - e.k0 + 1;
  int i, j, k, b, a0, a1;
  if (nie == 1 && nke > 1) {
    if (!rect_ij(nje, nke, a0, a1)) return;
    i = e.i0, j = e.j0 + a0, k = e.k0 + a1, b = blockIdx.z;
  } else if (nje == 1 && nke > 1) {
    if (!rect_ij(nie, nke, a0, a1)) return;
    i = e.i0 + a0, j = e.j0, k = e.k0 + a1, b = blockIdx.z;
  } else {
    if (!rect_ij(nie, nje, a0, a1)) return;
    i = e.i0 + a0, j = e.j0 + a1;
    b = blockIdx.z / nke;
    k = e.k0 + blockIdx.z % nke;
  }
  flux_face<FLUID, RECON, RS, DIR>(pv, pv.blocks[b], i, j, k, gamma, c_h);
}

template <int FLUID, int RECON, int RS, int DIR>
inline void launch_flux_dir(const PackView &pv, const FluxExtent &e, double gamma, double c_h,
                            hipStream_t s) {
  const int nie = e.i1 - e.i0 + 1, nje = e.j1 - e.j0 + 1, nke = e.k1 - e.k0 + 1;
  dim3 block(64, 4, 1);
  const dim3 grid = (nie == 1 && nke > 1) ? rect_grid(nje, nke, pv.nblocks)
                                           : ((nje == 1 && nke > 1) ? rect_grid(nie, nke, pv.nblocks) : rect_grid(nie, nje, nke * pv.nblocks));
  hipLaunchKernelGGL((flux_kernel<FLUID, RECON, RS, DIR>), grid, block, 0, s, pv, e, gamma, c_h);
}

// Boundary planes of a LIST of (block, face) pairs in one launch: faces[n] = 6 * block + face with
// face = {x1 lower, x1 upper, x2 lower, ...} -- on a refined mesh the faces with a coarser or finer
// block behind them, a fraction of all faces, and six launches of a few dozen small planes each
// would run at the launch-latency floor.
template <int FLUID, int RECON, int RS, bool FROM_CONS = false>
__global__ void __launch_bounds__(256)
flux_planes_kernel(PackView pv, const int *faces, double gamma, double c_h, FluxConsInput ci) {
  const int code = faces[blockIdx.y];
  const int b = code / 6, f = code - 6 * b, dir = f / 2 + 1, side = f & 1;
  const int na = (dir == 1) ? pv.nx2 : pv.nx1;
  const int nb = (dir == 3) ? pv.nx2 : pv.nx3;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int a1 = t / na, a0 = t - a1 * na;
  if (a1 >= nb) return;
  const apk_block_desc blk = pv.blocks[b];
  if (dir == 1) flux_face<FLUID, RECON, RS, 1, FROM_CONS>(pv, blk, side ? pv.ie + 1 : pv.is, pv.js + a0, pv.ks + a1, gamma, c_h, &ci);
  else if (dir == 2) flux_face<FLUID, RECON, RS, 2, FROM_CONS>(pv, blk, pv.is + a0, side ? pv.je + 1 : pv.js, pv.ks + a1, gamma, c_h, &ci);
  else flux_face<FLUID, RECON, RS, 3, FROM_CONS>(pv, blk, pv.is + a0, pv.js + a1, side ? pv.ke + 1 : pv.ks, gamma, c_h, &ci);
}

// which faces a flux call covers
enum FluxFaces {
  FLUX_FACES_REFERENCE = 0,  // the reference's loop limits (one transverse ghost row / plane)
  FLUX_FACES_TIGHT = 1,      // CalculateFluxesTight's limits: [s, e + 1] in every active direction
  FLUX_FACES_BOUNDARY = 2    // only the two block-boundary planes of each direction, interior transverse
};

// the lower (side 0) / upper (side 1) boundary plane of direction dir
inline FluxExtent boundary_extent(const PackView &pv, int dir, int side) {
  FluxExtent e;
  e.i0 = pv.is, e.i1 = pv.ie, e.j0 = pv.js, e.j1 = pv.je, e.k0 = pv.ks, e.k1 = pv.ke;
  if (dir == 1) e.i0 = e.i1 = side ? pv.ie + 1 : pv.is;
  else if (dir == 2) e.j0 = e.j1 = side ? pv.je + 1 : pv.js;
  else e.k0 = e.k1 = side ? pv.ke + 1 : pv.ks;
  return e;
}

template <int FLUID, int RECON, int RS, int DIR>
inline void launch_flux_faces(const PackView &pv, double gamma, double c_h, hipStream_t s, int faces) {
  if (faces == FLUX_FACES_BOUNDARY) {
    launch_flux_dir<FLUID, RECON, RS, DIR>(pv, boundary_extent(pv, DIR, 0), gamma, c_h, s);
    launch_flux_dir<FLUID, RECON, RS, DIR>(pv, boundary_extent(pv, DIR, 1), gamma, c_h, s);
  } else {
    launch_flux_dir<FLUID, RECON, RS, DIR>(pv, faces == FLUX_FACES_TIGHT ? tight_extent(pv, DIR) : flux_extent(pv, DIR), gamma,
                                           c_h, s);
  }
}

// face_list != NULL (with faces == FLUX_FACES_BOUNDARY): only the listed (block, face) planes, one launch
template <int FLUID, int RECON, int RS>
inline int launch_flux_all_dirs(const PackView &pv, double gamma, double c_h, hipStream_t s,
                                int faces = FLUX_FACES_REFERENCE, const int *face_list = nullptr, int nlist = 0,
                                const FluxConsInput *from_cons = nullptr) {
  if (from_cons && !face_list) return APK_ERR_UNSUPPORTED;
  if (face_list) {
    if (faces != FLUX_FACES_BOUNDARY) return APK_ERR_UNSUPPORTED;
    if (nlist <= 0) return APK_OK;
    int64_t plane = (int64_t)pv.nx1 * pv.nx2;
    if ((int64_t)pv.nx1 * pv.nx3 > plane) plane = (int64_t)pv.nx1 * pv.nx3;
    if ((int64_t)pv.nx2 * pv.nx3 > plane) plane = (int64_t)pv.nx2 * pv.nx3;
    for (int off = 0; off < nlist; off += 65535) {  // gridDim.y is limited to 65535
      const int m = (nlist - off > 65535) ? 65535 : nlist - off;
      const dim3 grid((unsigned)((plane + 255) / 256), (unsigned)m, 1);
      if (from_cons) {
        // (passive scalars ride the stored primitives' arrays: not offered)
        if (pv.nvar != nvars<FLUID>()) return APK_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((flux_planes_kernel<FLUID, RECON, RS, true>), grid, dim3(256), 0, s, pv, face_list + off, gamma, c_h, *from_cons);
      } else {
        hipLaunchKernelGGL((flux_planes_kernel<FLUID, RECON, RS, false>), grid, dim3(256), 0, s, pv, face_list + off, gamma, c_h, FluxConsInput{});
      }
    }
    return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
  }
  launch_flux_faces<FLUID, RECON, RS, 1>(pv, gamma, c_h, s, faces);
  if (pv.ndim >= 2) launch_flux_faces<FLUID, RECON, RS, 2>(pv, gamma, c_h, s, faces);
  if (pv.ndim >= 3) launch_flux_faces<FLUID, RECON, RS, 3>(pv, gamma, c_h, s, faces);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

// recon dispatch for one (fluid, riemann) family: the registry of hydro.cpp:386-416
template <int FLUID, int RS>
inline int launch_flux_family(const PackView &pv, int recon, double gamma, double c_h,
                              hipStream_t s, int faces, const int *face_list = nullptr, int nlist = 0,
                              const FluxConsInput *from_cons = nullptr) {
  switch (recon) {
  case APK_RC_DC: return launch_flux_all_dirs<FLUID, APK_RC_DC, RS>(pv, gamma, c_h, s, faces, face_list, nlist, from_cons);
  case APK_RC_PLM: return launch_flux_all_dirs<FLUID, APK_RC_PLM, RS>(pv, gamma, c_h, s, faces, face_list, nlist, from_cons);
  case APK_RC_PPM: return launch_flux_all_dirs<FLUID, APK_RC_PPM, RS>(pv, gamma, c_h, s, faces, face_list, nlist, from_cons);
  case APK_RC_WENOZ: return launch_flux_all_dirs<FLUID, APK_RC_WENOZ, RS>(pv, gamma, c_h, s, faces, face_list, nlist, from_cons);
  case APK_RC_WENO3: return launch_flux_all_dirs<FLUID, APK_RC_WENO3, RS>(pv, gamma, c_h, s, faces, face_list, nlist, from_cons);
  case APK_RC_LIMO3: return launch_flux_all_dirs<FLUID, APK_RC_LIMO3, RS>(pv, gamma, c_h, s, faces, face_list, nlist, from_cons);
  default: return APK_ERR_UNSUPPORTED;
  }
}

}  // namespace apk
