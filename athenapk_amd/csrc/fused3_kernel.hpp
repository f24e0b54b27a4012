// fused3_kernel.hpp -- a whole 3-D stage of a THREE-POINT reconstruction (PLM, WENO3, LimO3) in ONE march.
//
// The two-kernel stage (fused2_kernel.hpp) moves the x3 sweep's flux differences through memory once (write + read)
// and reads the input state twice: 200 - 240 B per cell of a hydro stage whose algorithm needs 120 - 160,
// on kernels that run at the memory system's rate (hydro PLM+HLLC: x3 sweep 5.0 TB/s, finishing march 4.3).  With a
// three-point stencil the donor-cell predictor's form (fused_dc3r2_kernel) carries over: a lane owns the SAME column
// (j, i) of two neighbouring x2 rows and marches along k,
//   x1   stencil neighbours i -/+ 1 from the neighbouring lanes by DPP wave shifts, L states one lane right, fluxes one
//        lane left (60 of 64 lanes retire: two stencil lanes either side);
//   x2   rows j-2 .. j+3 of the plane: the lane's own two + two halo rows either side (loaded from the L2 / MALL, where
//        the waves two chunks on -- same XCD, same plane -- have just put them), 4 reconstructions and 3 Riemann
//        problems for 2 cells (the face between the two cells is solved once);
//   x3   carried along the march: planes c-1, c in registers while plane c+1 arrives, the previous face's L state and
//        flux and the (x1 + x2) flux difference of the plane before in registers / a private LDS stash;
// and finishes the cell of plane c-1: du = (d1 + d2) + d3 in the reference's accumulation order (hydro.cpp:1070-1199),
// RK update, [Dedner,] ConsToPrim for the time-step estimate -- no flux-difference array at all.  The input is the
// CONSERVED state (apk_stage_args.prim_from_cons: the form a prim-free RK cycle is made of), converted in registers
// where a row is loaded; the raw values of the two planes behind stay in registers, because the cell that retires is
// one of them and its conserved value is what the update reads (as u1 where the input is u1's state, as the old u0 where
// it is u0's).  Per cell-stage: 40 B read + 40 B written (+ 40 B of u1 in stages with gam0 != 0) for hydro -- the
// algorithm's 80 / 120 B, against 200 / 240.
//
// Same pointwise functions, same operation order as the three reference tasks: bit-identical in the parity build
// (tests/test_gpu_parity.py: registry + prim-free forms).
// (included by fused_kernel.hpp after fused2_kernel.hpp, before its launch helpers)
#pragma once

namespace apk {

constexpr int kS3Halo = 2;               // lanes either side of a wave that retire nothing (stencil + face)
constexpr int kS3Cells = 64 - 2 * kS3Halo;  // cells a wave retires per row and plane

// SRC = apk_stage_args.prim_from_cons: 1 the input state is u1.cons, 2 it is u0.cons (the result goes to cons + out_delta)
template <int FLUID, int RECON, int RS, int EXTRA, int SRC>
__global__ void __launch_bounds__(64, 2)
fused_s3_kernel(PackView u0, PackView u1, StageParams sp, int kseg, int wpb, int nseg, int per_xcd) {
  static_assert(recon_halfwidth(RECON) == 1, "three-point reconstructions only");
  static_assert(SRC == 1 || SRC == 2, "input from a conserved state");
  constexpr int NV = nvars<FLUID>();
  double lane_min_dt = 1.7976931348623157e308;
  const int lane = threadIdx.x;
  const int vid = (int)(blockIdx.x % 8u) * per_xcd + (int)(blockIdx.x / 8u);  // XCD-aware order (see fused_dc3_kernel)
  if (vid >= wpb * nseg * u0.nblocks) return;
  const int chunk = vid % wpb;
  const int segid = (vid / wpb) % nseg;
  const int b = vid / (wpb * nseg);
  apk_block_desc b0 = u0.blocks[b];
  b0.cons = uniform_ptr(b0.cons);
  const double *c1 = uniform_ptr(u1.blocks[b].cons);
  double *prim_dst = (EXTRA != EXTRA_NONE && !sp.no_prim_store) ? uniform_ptr(u1.blocks[b].prim) : nullptr;

  const int i0 = u0.is - kS3Halo, rl = u0.nx1 + 2 * kS3Halo;
  const int64_t run = (int64_t)(u0.nx2 / 2) * rl;
  const int64_t t = (int64_t)chunk * kS3Cells + lane - kS3Halo;
  if ((int64_t)chunk * kS3Cells - kS3Halo >= run) return;
  const bool in_run = (t >= 0) && (t < run);
  const int64_t tc = in_run ? t : (t < 0 ? 0 : run - 1);  // out-of-run lanes shadow a valid column
  const int rowpair = (int)(tc / rl);
  const int i = i0 + (int)(tc - (int64_t)rowpair * rl);
  const bool active = in_run && (lane >= kS3Halo) && (lane <= 63 - kS3Halo) && (i >= u0.is) && (i <= u0.ie);
  const int ja = u0.js + 2 * rowpair;  // rows ja (cell A) and ja + 1 (cell B)
  const int64_t col = (int64_t)ja * u0.sj + i;
  const unsigned col_boff = (unsigned)(col * (int64_t)sizeof(double));  // (RowCellAt: the lane's share of a cell's address)

  const apk_block_desc *srcb = (SRC == 1) ? u1.blocks : u0.blocks;
  const double *in = srcb[b].cons + col;                           // rows ja, ja + 1
  const double *in_lo = in - 2 * u0.sj, *in_hi = in + 2 * u0.sj;   // rows ja - 2, ja - 1 / ja + 2, ja + 3
  const double *in_klo = in, *in_khi = in;                         // planes below ks / above ke
  if (sp.face_nbr) {  // direct neighbour addressing, as in fused_dc3r2_kernel
    const int *fn = sp.face_nbr + 6 * b;
    if (i < u0.is || i > u0.ie) {
      const int nb = fn[i < u0.is ? 0 : 1];
      if (nb >= 0) in = srcb[nb].cons + col + (i < u0.is ? u0.nx1 : -u0.nx1);
      in_lo = in - 2 * u0.sj;
      in_hi = in + 2 * u0.sj;
      in_klo = in_khi = in;
    } else {
      if (ja - 1 < u0.js && fn[2] >= 0) in_lo = srcb[fn[2]].cons + col - 2 * u0.sj + (int64_t)u0.nx2 * u0.sj;
      if (ja + 2 > u0.je && fn[3] >= 0) in_hi = srcb[fn[3]].cons + col + 2 * u0.sj - (int64_t)u0.nx2 * u0.sj;
      if (fn[4] >= 0) in_klo = srcb[fn[4]].cons + col + (int64_t)u0.nx3 * u0.sk;
      if (fn[5] >= 0) in_khi = srcb[fn[5]].cons + col - (int64_t)u0.nx3 * u0.sk;
    }
  }
  // (the blocks behind the x3 faces as element offsets from `in`: a select between integers, not between pointers)
  const int64_t off_klo = in_klo - in, off_khi = in_khi - in;
  const auto g_in = as_global(in), g_lo = as_global(in_lo), g_hi = as_global(in_hi);
  const double dx1 = b0.dx[0], dx2 = b0.dx[1], dx3 = b0.dx[2];
  const double area1 = to_sgpr(b0.dx[1] * b0.dx[2]), area2 = to_sgpr(b0.dx[0] * b0.dx[2]), area3 = to_sgpr(b0.dx[0] * b0.dx[1]);
  const double vol = to_sgpr(b0.dx[0] * b0.dx[1] * b0.dx[2]);
  const double upd = update_coefficient(sp, vol);
  const int s = u0.ks + segid * kseg;
  if (s > u0.ke) return;
  const int e = (s + kseg - 1 < u0.ke) ? s + kseg - 1 : u0.ke;

  extern __shared__ __attribute__((aligned(16))) double stash[];
  // [cell][f3 | du][var][lane]: the previous x3 face's flux and the (x1 + x2) flux difference of the plane before;
  // then [plane parity][cell][var][lane]: the conserved values of the two planes behind as loaded (the cell that retires is
  // one of them: its value is what the update reads as u1 / as the old u0)
  double *st_f3[2] = {stash + lane, stash + 2 * NV * 64 + lane};
  double *st_du[2] = {stash + NV * 64 + lane, stash + 3 * NV * 64 + lane};
  double *st_raw = stash + 4 * NV * 64 + lane;  // + ((p & 1) * 2 + r) * NV * 64 + n * 64

  // (returns the APK_FLAG_* bits of the converted cell, see cons_row_to_prim)
  auto to_prim = [&](const double (&u)[NV], double (&w)[NV]) -> unsigned {
    double tmp[NV], di;
#pragma unroll
    for (int n = 0; n < NV; ++n) tmp[n] = u[n];
    return cons_to_prim_core<FLUID, true>(sp.eos, sp.k.eos_gm1, sp.k.vceil_sq, sp.k.pfloor_over_gm1, tmp, w, di);
  };
  // the own rows of plane p (wave-uniform choice of the block behind an x3 face)
  auto load_plane = [&](int p, double (&raw)[2][NV]) {
    const int64_t po = (int64_t)p * u0.sk + ((p < u0.ks) ? off_klo : ((p > u0.ke) ? off_khi : (int64_t)0));
    const auto pc = g_in + po;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int n = 0; n < NV; ++n) raw[r][n] = pc[n * u0.sn + r * u0.sj];
  };

  double wm[2][NV], w0[2][NV];  // primitives of planes c-1, c
  double ql3_prev[2][NV];       // L state at x3 face c (from cell c-1), natural variable order
  {
    double ra[2][NV], rb[2][NV];
    load_plane(s - 2, ra);
    load_plane(s - 1, rb);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      (void)to_prim(ra[r], wm[r]);
      (void)to_prim(rb[r], w0[r]);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        st_f3[r][n * 64] = 0.0;
        st_du[r][n * 64] = 0.0;
        ql3_prev[r][n] = 0.0;
      }
    }
  }

  for (int c = s - 1; c <= e + 1; ++c) {
    const int64_t off = (int64_t)c * u0.sk;
    const bool mid = (c >= s) && (c <= e);  // wave-uniform: plane c has x1 / x2 faces to solve
    // ---- (0) the own rows of plane c+1: in flight during the x1 phase
    double rawp[2][NV];
    load_plane(c + 1, rawp);
    // ---- (1) x1 faces of both rows of plane c (kept in registers: the stash still holds plane c-1's differences)
    double du1[2][NV];
    if (mid) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        double ql1[NV], qr1[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          const double q0 = w0[r][n];
          reconstruct<RECON>(0.0, wave_shr1(q0), q0, wave_shl1(q0), 0.0, dx1, n, ql1[n], qr1[n]);
        }
        double wl[NV], wr[NV], f[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wl[q] = wave_shr1(ql1[perm<1>(q)]);
          wr[q] = qr1[perm<1>(q)];
        }
        riemann<FLUID, RS>(wl, wr, sp.k, f);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const double fup = wave_shl1(f[q]);
          du1[r][perm<1>(q)] = (area1 * fup - area1 * f[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- (2) x3: convert plane c+1, reconstruct cell c, solve face c - 1/2, retire cell c-1
    double wp[2][NV];
    {
      // every interior cell of the segment's planes is converted here exactly once: a negative density / pressure the
      // previous stage of a prim-free cycle left behind is latched now (see cons_row_to_prim)
      const unsigned fl = to_prim(rawp[0], wp[0]) | to_prim(rawp[1], wp[1]);
      if (active && fl) atomicOr(sp.flags, fl);
    }
    // (the two x2 halo rows below the pair: requested here, consumed after the x3 solves)
    double hlo[2][NV];
    if (mid) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int n = 0; n < NV; ++n) hlo[r][n] = g_lo[n * u0.sn + r * u0.sj + off];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      double ql3[NV], qr3[NV];
#pragma unroll
      for (int n = 0; n < NV; ++n) reconstruct<RECON>(0.0, wm[r][n], w0[r][n], wp[r][n], 0.0, dx3, n, ql3[n], qr3[n]);
      double *const rawslot = st_raw + (((c + 1) & 1) * 2 + r) * NV * 64;  // plane c-1 now, plane c+1 from here on
      if (c >= s) {
        double wl[NV], wr[NV], f3[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wl[q] = ql3_prev[r][perm<3>(q)];
          wr[q] = qr3[perm<3>(q)];
        }
        riemann<FLUID, RS>(wl, wr, sp.k, f3);
        if (c >= s + 1) {
          const RowCellAt done{(int64_t)(c - 1) * u0.sk + r * u0.sj, col_boff};
          double du[NV], u1v[NV], held[NV];
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            const int n = perm<3>(q);
            du[n] = st_du[r][n * 64] + (area3 * f3[q] - area3 * st_f3[r][q * 64]);
          }
#pragma unroll
          for (int n = 0; n < NV; ++n) held[n] = rawslot[n * 64];
          if constexpr (SRC == 1) {
            // (the input state IS u1: the plane just completed is at hand)
            if (active) finish_cell_at<FLUID, EXTRA, true>(u0, b0, held, done, du, vol, sp, lane_min_dt, prim_dst, upd);
          } else {
            if (active) {
              load_vars<NV>(c1, u0.sn, done, u1v);
              // (the input state is the old u0 the update reads)
              finish_cell_old_held<FLUID, EXTRA, true>(u0, b0, u1v, done, du, vol, sp, lane_min_dt, prim_dst, upd, held);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) st_f3[r][q * 64] = f3[q];
      }
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        ql3_prev[r][n] = ql3[n];
        rawslot[n * 64] = rawp[r][n];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (mid) {
      // (the two halo rows above the pair: in flight during the lower x2 face)
      double hhi[2][NV];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int n = 0; n < NV; ++n) hhi[r][n] = g_hi[n * u0.sn + r * u0.sj + off];
      // ---- (3) x2 faces: below A, between A and B (ONE solve, both cells use it), above B
      double qlA[NV], qrB[NV], flo[NV];
      {
        double wlo2[NV], wlo1[NV], ql_lo[NV], qrA[NV], dummy;
        (void)to_prim(hlo[0], wlo2);
        (void)to_prim(hlo[1], wlo1);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          reconstruct<RECON>(0.0, wlo2[n], wlo1[n], w0[0][n], 0.0, dx2, n, ql_lo[n], dummy);
          reconstruct<RECON>(0.0, wlo1[n], w0[0][n], w0[1][n], 0.0, dx2, n, qlA[n], qrA[n]);
        }
        double wl[NV], wr[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wl[q] = ql_lo[perm<2>(q)];
          wr[q] = qrA[perm<2>(q)];
        }
        riemann<FLUID, RS>(wl, wr, sp.k, flo);
      }
      __builtin_amdgcn_sched_barrier(0);
      double qlB[NV], qr_hi[NV];
      {
        double whi1[NV], whi2[NV], dummy;
        (void)to_prim(hhi[0], whi1);
        (void)to_prim(hhi[1], whi2);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          reconstruct<RECON>(0.0, w0[0][n], w0[1][n], whi1[n], 0.0, dx2, n, qlB[n], qrB[n]);
          reconstruct<RECON>(0.0, w0[1][n], whi1[n], whi2[n], 0.0, dx2, n, dummy, qr_hi[n]);
        }
      }
      double fmid[NV];
      {
        double wl[NV], wr[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wl[q] = qlA[perm<2>(q)];
          wr[q] = qrB[perm<2>(q)];
        }
        riemann<FLUID, RS>(wl, wr, sp.k, fmid);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<2>(q);
          st_du[0][n * 64] = du1[0][n] + (area2 * fmid[q] - area2 * flo[q]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        double wl[NV], wr[NV], fhi[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wl[q] = qlB[perm<2>(q)];
          wr[q] = qr_hi[perm<2>(q)];
        }
        riemann<FLUID, RS>(wl, wr, sp.k, fhi);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<2>(q);
          st_du[1][n * 64] = du1[1][n] + (area2 * fhi[q] - area2 * fmid[q]);
        }
      }
    }
    // ---- plane c+1 becomes plane c
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        wm[r][n] = w0[r][n];
        w0[r][n] = wp[r][n];
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (EXTRA == EXTRA_C2P_DT) {
    double m = lane_min_dt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if (lane == 0) atomicMin(sp.dt_bits, (unsigned long long)__double_as_longlong(m));
  }
}

// does a stage take the single-march form?  3-D, a three-point reconstruction, the lean form with its input derived from
// a conserved state (what a prim-free RK cycle asks of every stage), whole blocks (a split stage keeps the two kernels:
// its x3 sweep runs on plane windows while the halo messages fly), an even number of x2 rows, no primitives stored.
// Hydro with PLM for now: 256 VGPRs, no scratch (WENO3 / LimO3 spill 44 - 98 registers in this form, GLM-MHD would hold
// 2 x 9 x 7 doubles across a Riemann solve): the others keep the two-kernel stage.  APK_S3=0 switches it off (A/B).
template <int FLUID, int RECON>
constexpr bool single_march_compiled() { return FLUID == APK_FLUID_EULER && RECON == APK_RC_PLM; }
template <int FLUID, int RECON>
inline bool single_march_stage_applies(const PackView &u0, int extra, const StageParams &sp) {
  static const int mode = std::getenv("APK_S3") ? std::atoi(std::getenv("APK_S3")) : 1;
  if constexpr (!single_march_compiled<FLUID, RECON>()) return false;
  // (rows of 32 cells and more: on the 16^3 blocks of a refined mesh the march's x1 halo lanes outnumber its cells and the
  // two-kernel form is faster -- refined hydro blast of BASELINE config 5, zone-cycles/s, same box: 16^3 blocks 2.02e9 with
  // this march against 2.32e9 with the two-kernel stage; 32^3: 4.55e9 against 4.17e9; 48^3: 5.85e9 against 5.23e9)
  return mode != 0 && u0.ndim == 3 && u0.nx1 >= 32 && (uint64_t)u0.sn * sizeof(double) < (1ull << 32) && sp.prim_from_cons != 0 && sp.phase == 0 && sp.window == nullptr && stage_is_lean(sp) &&
         u0.nx2 % 2 == 0 && u0.nx2 >= 4 && u0.ng >= 2 && (extra == EXTRA_NONE || (extra == EXTRA_C2P_DT && sp.no_prim_store)) &&
         (sp.prim_from_cons == 1 || sp.out_delta != 0);
}

template <int FLUID, int RECON, int RS>
inline void launch_s3(const PackView &u0, const PackView &u1, const StageParams &sp, int extra, hipStream_t s) {
  if constexpr (single_march_compiled<FLUID, RECON>()) {
    const int64_t run = (int64_t)(u0.nx2 / 2) * (u0.nx1 + 2 * kS3Halo);
    const int wpb = (int)((run + kS3Cells - 1) / kS3Cells);
    // a segment costs its planes plus two for the prologue (12.5 % at 16).  Measured on 8 x 128^3 (round 6, same box, ms per
    // stage): 8 planes 0.742, 12: 0.749, 13: 0.747, **14: 0.705, 15: 0.705**, 16: 0.722, 19: 0.723, 22: 0.737, 24: 0.730,
    // 32: 0.795, 64: 0.97 -- 15 (nine segments, 10152 waves = 4.96 rounds of the 2048 the device holds) where 16 is
    // 4.41 rounds with the fifth one mostly empty
    static const int forced_kseg = std::getenv("APK_S3_KSEG") ? std::atoi(std::getenv("APK_S3_KSEG")) : 0;  // A/B switch
    int kseg = forced_kseg > 0 ? forced_kseg : 15;
    if (kseg > u0.nx3) kseg = u0.nx3;
    const int nseg = (u0.nx3 + kseg - 1) / kseg;
    const int64_t total = (int64_t)wpb * nseg * u0.nblocks;
    const int per_xcd = (int)((total + 7) / 8);
    const dim3 g((unsigned)(per_xcd * 8), 1, 1);
    constexpr int lds = 8 * nvars<FLUID>() * 64 * (int)sizeof(double);  // f3 + du + two planes as loaded, two cells each
#define APK_LAUNCH_S3(EXTRA_, SRC_) \
  hipLaunchKernelGGL((fused_s3_kernel<FLUID, RECON, RS, EXTRA_, SRC_>), g, dim3(64), lds, s, u0, u1, sp, kseg, wpb, nseg, per_xcd)
    if (extra == EXTRA_C2P_DT) {
      if (sp.prim_from_cons == 2) APK_LAUNCH_S3(EXTRA_C2P_DT, 2);
      else APK_LAUNCH_S3(EXTRA_C2P_DT, 1);
    } else {
      if (sp.prim_from_cons == 2) APK_LAUNCH_S3(EXTRA_NONE, 2);
      else APK_LAUNCH_S3(EXTRA_NONE, 1);
    }
#undef APK_LAUNCH_S3
  }
}

}  // namespace apk
