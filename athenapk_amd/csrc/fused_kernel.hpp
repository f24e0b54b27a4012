// fused_kernel.hpp -- fused reconstruct -> Riemann -> flux-difference sweeps, one kernel per
// direction per pack, no face fluxes in HBM (behind apk_stage_fused()).
//
// Register-resident pencils, wave64-first:
//  * x1 sweep: the block's interior rows of one k-plane are one contiguous run of cells; a
//    wave takes 64 consecutive cells of that run (overlapping its neighbours by 2), every lane
//    reconstructs ITS cell once, the L state travels one lane to the right and the face flux
//    one lane to the left with DPP wave shifts (no LDS, no barrier).  62 of 64 lanes retire a
//    cell; lanes that land on ghost columns idle.
//  * x2 / x3 sweeps: one lane per x1 column (coalesced 512 B rows), marching along the sweep
//    direction with the 3/5-point stencil, the previous face's L state and the previous face
//    flux held in VGPRs -- the reference's "march j / march k with swapped scratch pencils"
//    (src/hydro/hydro.cpp:1112-1199) without scratch memory and without team barriers.
//  The flux difference is accumulated in the reference's order, du = x1 term (+ x2 term)
//  (+ x3 term), through one scratch array, and the sweep of the last active direction applies
//  u0 <- gam0 u0 + gam1 u1 + beta_dt (-du/V) and the Dedner source, so results are
//  bit-identical to CalculateFluxes + UpdateWithFluxDivergence + DednerSource.
#pragma once

#include "apk_internal.hpp"
#include "hydro_math.hpp"

namespace apk {

struct StageParams {
  double gamma, c_h;
  double gam0, gam1, beta_dt;
  double dedner_coeff;
  int dedner;  // 0 off, 1 plain, 2 extended
  double *du;  // scratch: [nblocks][nvar][Nk][Nj][Ni]
  apk_ctx *ctx;  // host side only (kernel timing); never dereferenced on the device
};

// ---- DPP wave shifts (gfx9: wave_shr:1 = 0x138, wave_shl:1 = 0x130) ------------------------
APK_DEV double wave_shr1(double x) {  // lane l receives lane l-1 (lane 0 keeps its own)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
APK_DEV double wave_shl1(double x) {  // lane l receives lane l+1 (lane 63 keeps its own)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// ---- end-of-stage update of one cell (FINAL sweep) -------------------------------------------
// UpdateWithFluxDivergence (hydro_driver.cpp:534-537) then DednerSource
// (dedner_source.cpp:42-74), in that order, exactly as the task list runs them.
template <int FLUID>
APK_DEV void finish_cell(const PackView &pv, const apk_block_desc &b0, const double *c1,
                         int64_t cell, const double (&du)[nvars<FLUID>()], double vol,
                         const StageParams &sp) {
  constexpr int NV = nvars<FLUID>();
  double un[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const int64_t idx = n * pv.sn + cell;
    const double old = (sp.gam0 != 0.0) ? b0.cons[idx] : 0.0;
    un[n] = sp.gam0 * old + sp.gam1 * c1[idx] + sp.beta_dt * (-du[n] / vol);
  }
  if constexpr (FLUID == APK_FLUID_GLMMHD) {
    if (sp.dedner == 2) {
      const double *w = b0.prim + cell;
      const int64_t so = (pv.ndim >= 2) ? pv.sj : 0;
      const int64_t ko = (pv.ndim >= 3) ? pv.sk : 0;
      const double *b1 = w + IB1 * pv.sn, *b2 = w + IB2 * pv.sn, *b3 = w + IB3 * pv.sn;
      const double *ps = w + IPS * pv.sn;
      const double divB = 0.5 * ((b1[1] - b1[-1]) / b0.dx[0] + (b2[so] - b2[-so]) / b0.dx[1] +
                                 (b3[ko] - b3[-ko]) / b0.dx[2]);
      un[IM1] -= sp.beta_dt * divB * b1[0];
      un[IM2] -= sp.beta_dt * divB * b2[0];
      un[IM3] -= sp.beta_dt * divB * b3[0];
      un[IEN] -= 0.5 * sp.beta_dt *
                 (b1[0] * (ps[1] - ps[-1]) / b0.dx[0] + b2[0] * (ps[so] - ps[-so]) / b0.dx[1] +
                  b3[0] * (ps[ko] - ps[-ko]) / b0.dx[2]);
    }
    if (sp.dedner != 0) un[IPS] *= sp.dedner_coeff;
  }
#pragma unroll
  for (int n = 0; n < NV; ++n) b0.cons[n * pv.sn + cell] = un[n];
}

// ==============================================================================================
// x1 sweep
// ==============================================================================================
template <int FLUID, int RECON, int RS, bool FINAL>
__global__ void __launch_bounds__(256)
fused_x1_kernel(PackView u0, PackView u1, StageParams sp, int waves_per_plane) {
  constexpr int NV = nvars<FLUID>();
  constexpr int H = recon_halfwidth(RECON);
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= waves_per_plane) return;  // whole wave exits together
  const int b = blockIdx.y / u0.nx3;
  const int k = u0.ks + blockIdx.y % u0.nx3;
  const apk_block_desc b0 = u0.blocks[b];

  const int64_t run = (int64_t)u0.nx2 * u0.ni;  // contiguous cells of this plane's interior rows
  const int64_t t = (int64_t)wave * 62 + lane - 1;
  const int row = (int)((t >= 0 ? t : 0) / u0.ni);
  const int i = (int)(t - (int64_t)row * u0.ni);
  const bool in_run = (t >= 0) && (t < run);
  const bool do_recon = in_run && (i >= u0.is - 1) && (i <= u0.ie + 1);
  const int64_t cell = k * u0.sk + (int64_t)(u0.js + row) * u0.sj + i;
  const double dx = b0.dx[0];

  double qln[NV], qrn[NV];  // natural order: L state at face i+1, R state at face i
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    double qm2 = 1.0, qm1 = 1.0, q0 = 1.0, qp1 = 1.0, qp2 = 1.0;
    if (do_recon) {
      const double *c = b0.prim + n * u0.sn + cell;
      q0 = c[0];
      if constexpr (H >= 1) {
        qm1 = c[-1];
        qp1 = c[1];
      }
      if constexpr (H >= 2) {
        qm2 = c[-2];
        qp2 = c[2];
      }
    }
    reconstruct<RECON>(qm2, qm1, q0, qp1, qp2, dx, n, qln[n], qrn[n]);
  }
  // face i: L state from the lane on the left
  double wl[NV], wr[NV], f[NV];
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    wl[s] = wave_shr1(qln[perm<1>(s)]);
    wr[s] = qrn[perm<1>(s)];
  }
  riemann<FLUID, RS>(wl, wr, sp.gamma, sp.c_h, f);

  // cell i needs F(i) (own) and F(i+1) (lane on the right)
  const double a1 = b0.dx[1] * b0.dx[2];
  double du[NV];
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const double fup = wave_shl1(f[s]);
    du[perm<1>(s)] = (a1 * fup - a1 * f[s]);
  }
  const bool do_cell = in_run && (lane >= 1) && (lane <= 62) && (i >= u0.is) && (i <= u0.ie);
  if (!do_cell) return;
  if constexpr (FINAL) {
    const double vol = b0.dx[0] * b0.dx[1] * b0.dx[2];
    finish_cell<FLUID>(u0, b0, u1.blocks[b].cons, cell, du, vol, sp);
  } else {
    double *d = sp.du + (int64_t)b * u0.sn * u0.nvar + cell;
#pragma unroll
    for (int n = 0; n < NV; ++n) d[n * u0.sn] = du[n];
  }
}

// ==============================================================================================
// x2 / x3 sweeps: register-resident march
// ==============================================================================================
template <int FLUID, int RECON, int RS, int DIR, bool FINAL>
__global__ void __launch_bounds__(256)
fused_march_kernel(PackView u0, PackView u1, StageParams sp) {
  static_assert(DIR == 2 || DIR == 3, "march is for x2/x3");
  constexpr int NV = nvars<FLUID>();
  constexpr int H = recon_halfwidth(RECON);
  constexpr int W = 2 * H + 1;
  // lanes along x1 (interior only), the transverse index on blockIdx.y
  const int i = u0.is + blockIdx.x * 64 + (threadIdx.x & 63);
  const int trans = blockIdx.y * 4 + (threadIdx.x >> 6);  // k for DIR==2, j for DIR==3
  const int ntrans = (DIR == 2) ? u0.nx3 : u0.nx2;
  if (trans >= ntrans) return;  // wave-uniform
  const bool active = (i <= u0.ie);
  const int ii = active ? i : u0.ie;  // idle lanes shadow a valid column, never store
  const int b = blockIdx.z;
  const apk_block_desc b0 = u0.blocks[b];
  const double *c1 = u1.blocks[b].cons;

  const int64_t st = (DIR == 2) ? u0.sj : u0.sk;
  const int s = (DIR == 2) ? u0.js : u0.ks;  // first / last interior index along the march
  const int e = (DIR == 2) ? u0.je : u0.ke;
  const int64_t base = (DIR == 2) ? ((int64_t)(u0.ks + trans) * u0.sk + ii)
                                  : ((int64_t)(u0.js + trans) * u0.sj + ii);
  const double dx = b0.dx[DIR - 1];
  const double area = (DIR == 2) ? b0.dx[0] * b0.dx[2] : b0.dx[0] * b0.dx[1];
  const double vol = b0.dx[0] * b0.dx[1] * b0.dx[2];
  double *dscratch = sp.du + (int64_t)b * u0.sn * u0.nvar;

  // stencil registers P[n][0..W-1] = prim(n, c-H .. c+H) for the current cell c
  double P[NV][W];
  int c = s - 1;
#pragma unroll
  for (int n = 0; n < NV; ++n)
#pragma unroll
    for (int m = 0; m < W; ++m) P[n][m] = b0.prim[n * u0.sn + base + (int64_t)(c - H + m) * st];

  double wl_prev[NV];  // permuted L state at face c (from cell c-1)
  double f_prev[NV];   // permuted flux at face c-1
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    wl_prev[q] = 0.0;
    f_prev[q] = 0.0;
  }

  for (; c <= e + 1; ++c) {
    // prefetch the row entering the stencil next iteration
    double Pn[NV];
    const bool more = (c < e + 1);
    if (more) {
#pragma unroll
      for (int n = 0; n < NV; ++n) Pn[n] = b0.prim[n * u0.sn + base + (int64_t)(c + 1 + H) * st];
    }
    // reconstruct cell c
    double qln[NV], qrn[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      if constexpr (H == 0)
        reconstruct<RECON>(0.0, 0.0, P[n][0], 0.0, 0.0, dx, n, qln[n], qrn[n]);
      else if constexpr (H == 1)
        reconstruct<RECON>(0.0, P[n][0], P[n][1], P[n][2], 0.0, dx, n, qln[n], qrn[n]);
      else
        reconstruct<RECON>(P[n][0], P[n][1], P[n][2], P[n][3], P[n][4], dx, n, qln[n], qrn[n]);
    }
    if (c >= s) {
      double wr[NV], f[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) wr[q] = qrn[perm<DIR>(q)];
      riemann<FLUID, RS>(wl_prev, wr, sp.gamma, sp.c_h, f);
      if (c >= s + 1) {
        // cell c-1 is complete: (A F(c) - A F(c-1)) joins du
        const int64_t cell = base + (int64_t)(c - 1) * st;
        double du[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<DIR>(q);
          du[n] = dscratch[n * u0.sn + cell] + (area * f[q] - area * f_prev[q]);
        }
        if (active) {
          if constexpr (FINAL) {
            finish_cell<FLUID>(u0, b0, c1, cell, du, vol, sp);
          } else {
#pragma unroll
            for (int n = 0; n < NV; ++n) dscratch[n * u0.sn + cell] = du[n];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) f_prev[q] = f[q];
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) wl_prev[q] = qln[perm<DIR>(q)];
    if (more) {
#pragma unroll
      for (int n = 0; n < NV; ++n) {
#pragma unroll
        for (int m = 0; m + 1 < W; ++m) P[n][m] = P[n][m + 1];
        P[n][W - 1] = Pn[n];
      }
    }
  }
}

// ---- launch helpers ---------------------------------------------------------------------------
template <int FLUID, int RECON, int RS>
inline int launch_fused_stage(const PackView &u0, const PackView &u1, const StageParams &sp,
                              hipStream_t s) {
  const int64_t run = (int64_t)u0.nx2 * u0.ni;
  const int wpp = (int)((run + 61) / 62);
  const dim3 g1((wpp + 3) / 4, u0.nx3 * u0.nblocks, 1);
  // timing slots: donor-cell stages (VL2 predictor) are accounted separately
  constexpr int TS = (RECON == APK_RC_DC) ? (int)APK_T_FUSED_DC_X1 : (int)APK_T_FUSED_X1;
  if (u0.ndim == 1) {
    ScopedTiming t(sp.ctx, TS + 0, s);
    hipLaunchKernelGGL((fused_x1_kernel<FLUID, RECON, RS, true>), g1, dim3(256), 0, s, u0, u1, sp, wpp);
  } else {
    {
      ScopedTiming t(sp.ctx, TS + 0, s);
      hipLaunchKernelGGL((fused_x1_kernel<FLUID, RECON, RS, false>), g1, dim3(256), 0, s, u0, u1, sp, wpp);
    }
    const dim3 g2((u0.nx1 + 63) / 64, (u0.nx3 + 3) / 4, u0.nblocks);
    if (u0.ndim == 2) {
      ScopedTiming t(sp.ctx, TS + 1, s);
      hipLaunchKernelGGL((fused_march_kernel<FLUID, RECON, RS, 2, true>), g2, dim3(256), 0, s, u0, u1, sp);
    } else {
      {
        ScopedTiming t(sp.ctx, TS + 1, s);
        hipLaunchKernelGGL((fused_march_kernel<FLUID, RECON, RS, 2, false>), g2, dim3(256), 0, s, u0, u1, sp);
      }
      const dim3 g3((u0.nx1 + 63) / 64, (u0.nx2 + 3) / 4, u0.nblocks);
      ScopedTiming t(sp.ctx, TS + 2, s);
      hipLaunchKernelGGL((fused_march_kernel<FLUID, RECON, RS, 3, true>), g3, dim3(256), 0, s, u0, u1, sp);
    }
  }
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

template <int FLUID, int RS>
inline int launch_fused_family(const PackView &u0, const PackView &u1, int recon,
                               const StageParams &sp, hipStream_t s) {
  switch (recon) {
  case APK_RC_DC: return launch_fused_stage<FLUID, APK_RC_DC, RS>(u0, u1, sp, s);
  case APK_RC_PLM: return launch_fused_stage<FLUID, APK_RC_PLM, RS>(u0, u1, sp, s);
  case APK_RC_PPM: return launch_fused_stage<FLUID, APK_RC_PPM, RS>(u0, u1, sp, s);
  case APK_RC_WENOZ: return launch_fused_stage<FLUID, APK_RC_WENOZ, RS>(u0, u1, sp, s);
  case APK_RC_WENO3: return launch_fused_stage<FLUID, APK_RC_WENO3, RS>(u0, u1, sp, s);
  case APK_RC_LIMO3: return launch_fused_stage<FLUID, APK_RC_LIMO3, RS>(u0, u1, sp, s);
  default: return APK_ERR_UNSUPPORTED;
  }
}

int launch_fused_euler_hlle(const PackView &u0, const PackView &u1, int recon, const StageParams &sp, hipStream_t s);
int launch_fused_euler_hllc(const PackView &u0, const PackView &u1, int recon, const StageParams &sp, hipStream_t s);
int launch_fused_mhd_hlle(const PackView &u0, const PackView &u1, int recon, const StageParams &sp, hipStream_t s);
int launch_fused_mhd_hlld(const PackView &u0, const PackView &u1, int recon, const StageParams &sp, hipStream_t s);

}  // namespace apk
