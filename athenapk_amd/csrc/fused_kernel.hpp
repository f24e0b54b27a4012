// fused_kernel.hpp -- fused reconstruct -> Riemann -> flux-difference sweeps, one kernel per
// direction per pack, no face fluxes in HBM (behind apk_stage_fused()).
//
// Register-resident pencils, wave64-first:
//  * x1 sweep: the block's interior rows of one k-plane are one contiguous run of cells; a
//    wave takes 64 consecutive cells of that run (overlapping its neighbours by 2), every lane
//    reconstructs ITS cell once, the L state travels one lane to the right and the face flux
//    one lane to the left with DPP wave shifts (no LDS, no barrier).  62 of 64 lanes retire a
//    cell; lanes that land on ghost columns idle.
//  * x2 / x3 sweeps: one lane per x1 column (coalesced 512 B rows), one wave per workgroup
//    marching along the sweep direction; the 3/5-point stencil sits in a private LDS ring, the
//    newest row, the previous face's L state and the previous face flux in VGPRs -- the
//    reference's "march j / march k with swapped scratch pencils"
//    (src/hydro/hydro.cpp:1112-1199) without global scratch and without team barriers.
//  The flux difference is accumulated in the reference's order, du = x1 term (+ x2 term)
//  (+ x3 term), through one scratch array, and the sweep of the last active direction applies
//  u0 <- gam0 u0 + gam1 u1 + beta_dt (-du/V) and the Dedner source, so results are
//  bit-identical to CalculateFluxes + UpdateWithFluxDivergence + DednerSource.
#pragma once

#include <cstdlib>
#include <type_traits>

#include "apk_internal.hpp"
#include "hydro_math.hpp"

namespace apk {

struct StageParams {
  StageConsts k;  // gamma, c_h and what the pointwise functions derive from them (host-evaluated: hydro_math.hpp)
  double gamma, c_h;
  double gam0, gam1, beta_dt;
  double dedner_coeff;
  int dedner;  // 0 off, 1 plain, 2 extended
  double *du;  // scratch: [nblocks][nvar][Nk][Nj][Ni]
  int du_first;  // non-finishing march: WRITE its flux difference to du instead of adding to it
  // Two-kernel stage: the x3 sweep's flux differences in a layout of their own -- interior cells only, rows of du_pitch
  // doubles (a multiple of 16) from a 128-byte boundary: [block][var][k][j][du_pitch].  The sweep's 512-byte stores are
  // then whole cache lines (in the cell arrays' layout they straddle five, two of them partly).  0: the cells' layout.
  int du_pitch;
  // optional work of the finishing sweep (apk_stage_args.fill_derived / estimate_dt)
  apk_eos eos;                   // floors / ceilings for the in-place ConsToPrim
  unsigned *flags;               // latched APK_FLAG_* word
  unsigned long long *dt_bits;   // min over cells of dx_d/(|v_d|+c_d), as ordered bits
  unsigned long long *bad_count;  // trial stage: number of cells failing FirstOrderFluxCorrect's test (or NULL)
  int64_t out_delta;  // the updated conserved state goes to cons + out_delta (elements; 0 = in place)
  int prim_to_u1;                // fill_derived = 2: the new primitives go to u1's prim arrays
  // optional per-block index window (apk_stage_args.window): {i0, rl, ilo, ihi, jlo, jhi, klo, khi}:
  // rows are flattened with length rl starting at column i0 and only cells ilo..ihi retire
  // passive scalars: the sweeps store the mass flux through both faces of every cell they retire,
  // mflux[(d * nblocks + b) * sn + cell] = flux through the lower d-face of `cell`; NULL without scalars
  double *mflux;
  const int *window;
  int window_rl, window_rows;  // largest rl / row count in `window` (size the grid)
  int phase;                   // apk_stage_args.phase
  // apk_stage_args.face_neighbor: [nblocks][6] pack index of the block behind each face whose interior
  // stands in for this block's ghost zone there (-1: the ghost zone itself), or NULL
  const int *face_nbr;
  int cons_store;  // apk_stage_args.cons_store (0 all cells, 1 the nghost-deep shell of every block, 2 none)
  int no_prim_store;   // apk_stage_args.fill_derived = 3: ConsToPrim for the dt estimate only
  int prim_from_cons;  // apk_stage_args.prim_from_cons
  // apk_stage_args.x1_halo: per block the exchange-buffer segments its x1 ghost columns are read from / its x1 boundary
  // columns are also stored into (device array, or NULL), their depths in columns, and what is stored (0 cons, 1 prim)
  const apk_x1_halo_block *x1_blocks;
  int x1_recv_depth, x1_send_depth, x1_send_field;
  apk_ctx *ctx;  // host side only (kernel timing); never dereferenced on the device
};

// A lane's share of apk_stage_args.x1_halo on the sending side: where the cells it retires along its march go in the
// send segment -- seg[n * sn + step * stride], `step` counting along the march from the first interior index -- or
// seg == nullptr for the lanes (nearly all) whose column is not within `depth` of an x1 face with a segment.  Segments
// are [var][k][j][depth]; `fixed` = the offset the lane's coordinates across the march contribute (j march: k * depth *
// nx2, stride depth; k march: j * depth, stride depth * nx2).
// Everything here lives in VECTOR registers on purpose, wave-uniform values included, and the lane test is redone at
// every use (lane_fresh): the marches have vector registers to spare and no scalar ones -- a loop-invariant lane mask or
// stride in scalar registers is spilled to vector lanes and costs a v_readlane per use (first form of this: +46 of them
// per iteration of the finishing march).
struct X1Store {
  double *seg = nullptr;
  int64_t sn = 0, stride = 0;
};
APK_DEV int64_t lane_fresh(int64_t x) {  // (the value as the compiler last saw it -- in a vector register pair, not to be hoisted)
  asm volatile("" : "+v"(x));
  return x;
}
template <class T>
APK_DEV T *lane_fresh(T *p) {
  asm volatile("" : "+v"(p));
  return p;
}
// (IN_VGPRS: the finishing march of the two-kernel stage; the donor-cell march has scalar registers to spare and no
// vector ones -- there the strides stay wave-uniform)
template <bool IN_VGPRS>
APK_DEV X1Store x1_store_of(const StageParams &sp, const PackView &pv, int b, int i, bool active, int64_t fixed, int64_t stride) {
  X1Store x;
  if (sp.x1_blocks && sp.x1_send_depth > 0 && active) {
    const int dpt = sp.x1_send_depth;
    const int side = (i < pv.is + dpt) ? 0 : ((i > pv.ie - dpt) ? 1 : -1);
    if (side >= 0) {
      double *seg = sp.x1_blocks[b].send[side];
      if (seg) x.seg = seg + fixed + (side ? i - (pv.ie - dpt + 1) : i - pv.is);
    }
  }
  x.sn = (int64_t)sp.x1_send_depth * pv.nx2 * pv.nx3;
  x.stride = stride;
  if constexpr (IN_VGPRS) {
    x.sn = lane_fresh(x.sn);
    x.stride = lane_fresh(x.stride);
  }
  return x;
}
template <int NV, bool IN_VGPRS = true>
APK_DEV void x1_store_row(X1Store &x, int step, const double (&v)[NV]) {
  if constexpr (IN_VGPRS) {
    x.seg = lane_fresh(x.seg);
    if (x.seg) {
      x.sn = lane_fresh(x.sn);
      double *p = x.seg + lane_fresh(x.stride) * step;
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        store_result(as_global(p), v[n]);
        if (n + 1 < NV) p = lane_fresh(p + x.sn);
      }
    }
  } else if (x.seg) {
    double *p = x.seg + x.stride * step;
#pragma unroll
    for (int n = 0; n < NV; ++n) store_result(&as_global(p)[n * x.sn], v[n]);
  }
}

// L / R state at the lower `st`-face of the cell c points at (L = ql of the cell below, R = qr of
// this cell), any reconstruction
template <int RECON>
APK_DEV void face_states_any(const double *c, int64_t st, double dx, int var, double &wl, double &wr) {
  double dummy;
  if constexpr (RECON == APK_RC_DC) {
    wl = c[-st];
    wr = c[0];
  } else if constexpr (RECON == APK_RC_PPM || RECON == APK_RC_WENOZ) {
    const double qm3 = c[-3 * st], qm2 = c[-2 * st], qm1 = c[-st], q0 = c[0], qp1 = c[st], qp2 = c[2 * st];
    reconstruct<RECON>(qm3, qm2, qm1, q0, qp1, dx, var, wl, dummy);
    reconstruct<RECON>(qm2, qm1, q0, qp1, qp2, dx, var, dummy, wr);
  } else {
    const double qm2 = c[-2 * st], qm1 = c[-st], q0 = c[0], qp1 = c[st];
    reconstruct<RECON>(0.0, qm2, qm1, q0, 0.0, dx, var, wl, dummy);
    reconstruct<RECON>(0.0, qm1, q0, qp1, 0.0, dx, var, dummy, wr);
  }
}

// what the finishing sweep does besides the RK update + Dedner source
enum { EXTRA_NONE = 0, EXTRA_C2P = 1, EXTRA_C2P_DT = 2 };

// x1 sweep (lanes along the flattened rows): first lane of a wave that retires a cell and the
// number of cells a wave retires.  Lane l needs the L state of lane l-1 and the flux of lane l+1;
// with PPM lane l-1's state in turn needs the interface value of lane l-2 (face sharing).
constexpr int x1_first_lane(int recon) { return recon == APK_RC_PPM ? 2 : 1; }
constexpr int x1_cells_per_wave(int recon) { return 63 - x1_first_lane(recon); }

// ---- DPP wave shifts (gfx9: wave_shr:1 = 0x138, wave_shl:1 = 0x130) ------------------------
// bound_ctrl:1 with a zero `old` operand: the lane without a neighbour (lane 0 / lane 63) receives 0
// -- no lane that retires a cell ever uses it -- and, unlike "keep the own value", the destination
// register needs no copy of the source before the v_mov_b32_dpp (one VALU instruction per half
// instead of two: -126 instructions per iteration of the finishing x1 + x2 march).
APK_DEV double wave_shr1(double x) {  // lane l receives lane l-1 (lane 0: 0.0)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
APK_DEV double wave_shl1(double x) {  // lane l receives lane l+1 (lane 63: 0.0)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// ---- where a kernel finds a cell in a block's arrays ------------------------------------------
// CellAt: a per-lane element index -- every access is a 64-bit vector add of n * sn + cell on the array's pointer.
// RowCellAt (the marches of the two-kernel stage): the rows a wave works on are wave-uniform, so the address splits into
// a scalar part (array + row + n * sn, advanced by the scalar unit: s_add_u32 / s_addc_u32, no vector-ALU issue slot and
// ONE scalar register pair per array instead of one per (array, variable)) and the lane's 32-bit byte offset inside the
// row, which never changes along a march: `global_load_dwordx2 v, v_off, s[base:base+1]`.  Two things keep the compiler
// on that form: the scalar pointer is re-materialised after every step (next_uniform: an empty asm with an SGPR
// constraint -- otherwise array + n * sn is hoisted out of the march as nine loop-invariant register pairs per array, which
// is what filled the scalar register file and spilled ~100 of them to vector lanes), and the lane offset is "fresh" in the
// basic block that uses it (the zero extension has to be visible to instruction selection there).  Needs a block's
// variable stride to fit 32 bits in bytes (two_kernel_stage_applies checks).
struct CellAt {
  int64_t cell;
};
struct RowCellAt {
  int64_t row;    // element offset of the row's first cell (wave-uniform)
  unsigned boff;  // byte offset of the lane's cell in the row
};
APK_DEV int64_t next_uniform(int64_t x) {
  // (readfirstlane: folded away where the compiler knows the value to be wave-uniform -- everywhere but in a few forms
  // where its divergence analysis gives up on a value that is uniform by construction)
  const int lo = __builtin_amdgcn_readfirstlane((int)(x & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
  x = ((int64_t)hi << 32) | (int64_t)(unsigned)lo;
  asm volatile("" : "+s"(x));
  return x;
}
template <class T>
APK_DEV T *next_uniform(T *p) {
  return reinterpret_cast<T *>(next_uniform(reinterpret_cast<int64_t>(p)));
}
// a pointer out of a block descriptor: the descriptors are read with vector loads where the kernel has stored before
// (no scalar load of memory that may have been clobbered), which leaves wave-uniform pointers in vector registers
template <class T>
APK_DEV T *uniform_ptr(T *p) {
  const int64_t x = reinterpret_cast<int64_t>(p);
  const int lo = __builtin_amdgcn_readfirstlane((int)(x & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
  return reinterpret_cast<T *>(((int64_t)hi << 32) | (int64_t)(unsigned)lo);
}
APK_DEV unsigned fresh_offset(unsigned x) {
  asm volatile("" : "+v"(x));
  return x;
}
template <int NV>
APK_DEV void load_vars(const double *arr, int64_t sn, const CellAt &a, double (&v)[NV]) {
#pragma unroll
  for (int n = 0; n < NV; ++n) v[n] = as_global(arr)[n * sn + a.cell];
}
template <int NV>
APK_DEV void load_vars(const double *arr, int64_t sn, const RowCellAt &a, double (&v)[NV]) {
  const double *p = next_uniform(arr + a.row);
  const unsigned boff = fresh_offset(a.boff);
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    v[n] = *(const __attribute__((address_space(1))) double *)((const __attribute__((address_space(1))) char *)p + boff);
    if (n + 1 < NV) p = next_uniform(p + sn);
  }
}
template <int NV>
APK_DEV void store_vars(double *arr, int64_t sn, const CellAt &a, const double (&v)[NV]) {
#pragma unroll
  for (int n = 0; n < NV; ++n) store_result(&as_global(arr)[n * sn + a.cell], v[n]);
}
template <int NV>
APK_DEV void store_vars(double *arr, int64_t sn, const RowCellAt &a, const double (&v)[NV]) {
  double *p = next_uniform(arr + a.row);
  const unsigned boff = fresh_offset(a.boff);
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    store_result((__attribute__((address_space(1))) double *)((__attribute__((address_space(1))) char *)p + boff), v[n]);
    if (n + 1 < NV) p = next_uniform(p + sn);
  }
}

// ---- end-of-stage update of one cell (FINAL sweep) -------------------------------------------
// UpdateWithFluxDivergence (hydro_driver.cpp:534-537) then DednerSource
// (dedner_source.cpp:42-74), in that order, exactly as the task list runs them.
// A stage is LEAN when none of the finishing sweep's optional work is asked for: no mass-flux workspace (passive
// scalars), no trial count (first-order flux correction), no extended Dedner source, and an equation of state whose
// velocity ceiling, pressure floor and energy ceiling are off (eos_is_lean).  The uniform-mesh cycles of the decks and
// of the benchmark are all of this kind.  finish_cell<.., LEAN = true> compiles none of that work: no scalar
// registers for its parameters (the finishing march spills ~100 of them to vector lanes, one v_readlane per use), one
// wave-uniform branch around the nine loads of the old state instead of nine, and -- product build -- the update as
// one fused multiply-add per variable on the wave-uniform coefficient `upd` = -beta_dt / V.
inline bool stage_is_lean(const StageParams &sp) {
  return sp.mflux == nullptr && sp.bad_count == nullptr && sp.dedner != 2 && eos_is_lean(sp.eos);
}
// 1: lean; LEAN_PFLOOR (2): lean but for a pressure floor and / or the trial count of first-order flux correction -- the
// Orszag-Tang deck has both -- whose few instructions the forms <.., LEAN = 2> compile in; 0: the general form
inline int stage_lean_level(const StageParams &sp) {
  if (stage_is_lean(sp)) return 1;
  return (sp.mflux == nullptr && sp.dedner != 2 && eos_is_lean_but_pfloor(sp.eos)) ? LEAN_PFLOOR : 0;
}
// -beta_dt / V of a block (product build; the parity build divides by V per cell as the reference does)
APK_DEV double update_coefficient(const StageParams &sp, double vol) { return to_sgpr(-sp.beta_dt / vol); }

// (HELD: old_held is the old u0 of this cell, already in registers -- the from-cons finishing march keeps the rows it loaded)
template <int FLUID, int EXTRA, int LEAN, bool HELD, class AT, bool XV = true>
APK_DEV void finish_cell_impl(const PackView &pv, const apk_block_desc &b0,
                              const double (&u1v)[nvars<FLUID>()], const AT &at,
                              const double (&du)[nvars<FLUID>()], double vol, const StageParams &sp,
                              double &lane_min_dt, double *prim_dst, double upd, bool store_cons,
                              const double (&old_held)[nvars<FLUID>()], X1Store *xs = nullptr, int xrow = 0) {
  constexpr int NV = nvars<FLUID>();
  // (the general form's per-variable accesses, the extended Dedner source: a per-lane cell index; lean forms only reach
  // here with a RowCellAt)
  [[maybe_unused]] int64_t cell = 0;
  if constexpr (std::is_same<AT, CellAt>::value) cell = at.cell;
  else static_assert(LEAN, "row addressing: lean forms only");
  double un[NV];
  if constexpr (LEAN) {
#ifdef APK_FP_STRICT
#define APK_UPD_TERM(n) (sp.beta_dt * (-du[n] / vol))
#else
#define APK_UPD_TERM(n) (upd * du[n])
#endif
    if (sp.gam0 != 0.0) {  // wave-uniform
      double old[NV];
      if constexpr (HELD) {
#pragma unroll
        for (int n = 0; n < NV; ++n) old[n] = old_held[n];
      } else {
        load_vars<NV>(b0.cons, pv.sn, at, old);
      }
#pragma unroll
      for (int n = 0; n < NV; ++n) un[n] = sp.gam0 * old[n] + sp.gam1 * u1v[n] + APK_UPD_TERM(n);
    } else {
      // (gam0 = 0: the reference's 0 * u0 + gam1 * u1 + ... is gam1 * u1 + ... to the bit -- x + 0 = x)
#pragma unroll
      for (int n = 0; n < NV; ++n) un[n] = sp.gam1 * u1v[n] + APK_UPD_TERM(n);
    }
#undef APK_UPD_TERM
  } else {
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const int64_t idx = n * pv.sn + cell;
    const double old = (sp.gam0 != 0.0) ? as_global(b0.cons)[idx] : 0.0;
    un[n] = sp.gam0 * old + sp.gam1 * u1v[n] + sp.beta_dt * (-du[n] / vol);
  }
  }
  if constexpr (FLUID == APK_FLUID_GLMMHD) {
    if constexpr (!LEAN) {
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
    }
    if (sp.dedner != 0) un[IPS] *= sp.dedner_coeff;
  }
  // Trial stage of first-order flux correction: FirstOrderFluxCorrect's admissibility test
  // (hydro.cpp:1297-1306: rho <= 0 or E - KE [- ME] <= 0, on the update BEFORE any floor) applied to
  // the very `un` this kernel computed.
  bool bad = false;
  if constexpr (LEAN != 1) {  // (the general form and the lean form with the trial count / the pressure floor)
  if (sp.bad_count) {
    double new_p = un[IEN] - 0.5 * (sqr(un[IM1]) + sqr(un[IM2]) + sqr(un[IM3])) / un[IDN];
    if constexpr (FLUID == APK_FLUID_GLMMHD) new_p -= 0.5 * (sqr(un[IB1]) + sqr(un[IB2]) + sqr(un[IB3]));
    bad = !(un[IDN] > 0.0 && new_p > 0.0);
  }
  }
  if constexpr (EXTRA != EXTRA_NONE) {
    // FillDerived for this cell (adiabatic_hydro.hpp:52-142): in a march along x2/x3 no other lane
    // reads this column of prim during the sweep and this lane's stencil copy sits in its LDS
    // ring, so prim can be replaced in place (prim_dst = this block's prim); kernels whose lanes
    // read neighbouring columns from memory get prim_dst = u1's prim arrays instead.
    // Floors/ceilings act on `un` before it is stored.
    double w[NV], di;
    const unsigned fl = cons_to_prim_cell<FLUID, LEAN>(sp.eos, sp.k, un, w, di);
    if (fl) atomicOr(sp.flags, fl);
    // (ConsToPrim forms the pressure with 1/rho where the test above divides: a trial stage is
    // only accepted if neither sees a negative state, so an accepted stage never raises flags)
    if constexpr (LEAN != 1) {
      if (sp.bad_count && fl) bad = true;
    }
    if (prim_dst) {  // (wave-uniform; NULL: fill_derived = 3, the primitives only feed the time-step estimate)
      store_vars<NV>(prim_dst, pv.sn, at, w);
    }
    if (xs && sp.x1_send_field == 1) x1_store_row<NV, XV>(*xs, xrow, w);  // (apk_stage_args.x1_halo: the new primitives)
    if constexpr (EXTRA == EXTRA_C2P_DT) {
      // EstimateHyperbolicTimestep (hydro.cpp:845-895) on the fresh primitives
      lane_min_dt = fmin(lane_min_dt, cell_dt_hyp<FLUID>(sp.eos.gamma, w, di, pv.ndim, b0.dx[0], b0.dx[1], b0.dx[2]));
    }
  }
  if (store_cons) store_vars<NV>(b0.cons + sp.out_delta, pv.sn, at, un);
  if (xs && sp.x1_send_field == 0) x1_store_row<NV, XV>(*xs, xrow, un);  // (apk_stage_args.x1_halo: the updated conserved state)
  if constexpr (LEAN != 1) {
    if (bad) atomicAdd(sp.bad_count, 1ull);
  }
}

template <int FLUID, int EXTRA = EXTRA_NONE, int LEAN = 0>
APK_DEV void finish_cell(const PackView &pv, const apk_block_desc &b0, const double (&u1v)[nvars<FLUID>()], int64_t cell,
                         const double (&du)[nvars<FLUID>()], double vol, const StageParams &sp, double &lane_min_dt,
                         double *prim_dst = nullptr, double upd = 0.0, bool store_cons = true, X1Store *xs = nullptr, int xrow = 0) {
  const double none[nvars<FLUID>()] = {};
  finish_cell_impl<FLUID, EXTRA, LEAN, false, CellAt, false>(pv, b0, u1v, CellAt{cell}, du, vol, sp, lane_min_dt, prim_dst, upd, store_cons, none, xs, xrow);
}
template <int FLUID, int EXTRA, int LEAN, class AT>
APK_DEV void finish_cell_at(const PackView &pv, const apk_block_desc &b0, const double (&u1v)[nvars<FLUID>()], const AT &at,
                            const double (&du)[nvars<FLUID>()], double vol, const StageParams &sp, double &lane_min_dt,
                            double *prim_dst, double upd, X1Store *xs = nullptr, int xrow = 0) {
  const double none[nvars<FLUID>()] = {};
  finish_cell_impl<FLUID, EXTRA, LEAN, false>(pv, b0, u1v, at, du, vol, sp, lane_min_dt, prim_dst, upd, true, none, xs, xrow);
}
template <int FLUID, int EXTRA, int LEAN, class AT>
APK_DEV void finish_cell_old_held(const PackView &pv, const apk_block_desc &b0, const double (&u1v)[nvars<FLUID>()], const AT &at,
                                  const double (&du)[nvars<FLUID>()], double vol, const StageParams &sp, double &lane_min_dt,
                                  double *prim_dst, double upd, const double (&old_held)[nvars<FLUID>()], X1Store *xs = nullptr, int xrow = 0) {
  static_assert(LEAN, "lean form only");
  finish_cell_impl<FLUID, EXTRA, LEAN, true>(pv, b0, u1v, at, du, vol, sp, lane_min_dt, prim_dst, upd, true, old_held, xs, xrow);
}

// ==============================================================================================
// x1 sweep
// ==============================================================================================
template <int FLUID, int RECON, int RS, bool FINAL>
__global__ void __launch_bounds__(256, 4)
fused_x1_kernel(PackView u0, PackView u1, StageParams sp, int waves_per_plane) {
  constexpr int NV = nvars<FLUID>();
  constexpr int H = recon_halfwidth(RECON);
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= waves_per_plane) return;  // whole wave exits together
  const int b = blockIdx.y / u0.nx3;
  const int k = u0.ks + blockIdx.y % u0.nx3;
  const apk_block_desc b0 = u0.blocks[b];

  // The interior rows of this plane are flattened into one run of cells, `rl` per row starting at
  // column i0 (the whole row by default; a thin window next to a face when the sweep is split
  // around a halo exchange that is still in flight).  Ghost / window-edge columns separate the rows.
  // lanes that retire a cell: FIRST .. 62 (x1_cells_per_wave of them); waves overlap accordingly
  constexpr int FIRST = x1_first_lane(RECON), CPW = x1_cells_per_wave(RECON);
  // (only the columns whose lanes reconstruct take part: is - FIRST .. ie + 1, not all nghost ghost
  // columns -- 16 of 19 lanes on interior cells of a 16-cell block with nghost = 4 instead of 16 of 24)
  int i0 = u0.is - FIRST, rl = u0.nx1 + FIRST + 1, lo = u0.is, hi = u0.ie;
  if (sp.window) {
    const int *w = sp.window + 8 * b;
    i0 = w[0], rl = w[1], lo = w[2], hi = w[3];
    if (rl <= 0) return;  // nothing to do in this block
  }
  const int64_t run = (int64_t)u0.nx2 * rl;
  const int64_t t = (int64_t)wave * CPW + lane - FIRST;
  if ((int64_t)wave * CPW - FIRST >= run) return;  // whole wave beyond this block's run
  const int row = (int)((t >= 0 ? t : 0) / rl);
  const int i = i0 + (int)(t - (int64_t)row * rl);
  const bool in_run = (t >= 0) && (t < run);
  const bool do_recon = in_run && (i >= u0.is - FIRST) && (i <= u0.ie + 1);
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
        // (PPM's first reconstructing column is is - 2: with nghost = 3 its i - 2 lies one element in front of the row,
        // which in 1-D -- one row per block, no rows below it -- is in front of the ARRAY for the first variable of the
        // first block: an 8-byte read outside the allocation, a fault whenever that allocation starts a mapping.  The
        // value only feeds that lane's own states, which nothing uses (PPM waves retire lanes 2 .. 62).)
        qm2 = (i >= 2) ? c[-2] : q0;
        qp2 = c[2];
      }
    }
    if constexpr (RECON == APK_RC_PPM) {
      // every limited interface value once: the lane computes the one above its cell and receives
      // the one below from the lane on its left (see ppm_interface); lane 0 has no left neighbour
      // and produces no valid state, which is why PPM waves retire lanes 2..62
      // (lanes that do not reconstruct hold a flat dummy stencil -- "an extremum" by the <= of the test -- and lane 0 a
      // zero for the interface below it: neither may take the wave into the limiter branches)
      const double face_p = ppm_interface(qm1, q0, qp1, qp2, do_recon ? 0.0 : kPpmNever);
      const double face_m = wave_shr1(face_p);
      ppm_cell(qm2, qm1, q0, qp1, qp2, face_m, face_p, qln[n], qrn[n], (do_recon && lane >= 1) ? 0.0 : kPpmNever);
    } else {
      reconstruct<RECON>(qm2, qm1, q0, qp1, qp2, dx, n, qln[n], qrn[n]);
    }
  }
  // face i: L state from the lane on the left
  double wl[NV], wr[NV], f[NV];
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    wl[s] = wave_shr1(qln[perm<1>(s)]);
    wr[s] = qrn[perm<1>(s)];
  }
  riemann<FLUID, RS>(wl, wr, sp.k, f);

  // cell i needs F(i) (own) and F(i+1) (lane on the right)
  const double a1 = b0.dx[1] * b0.dx[2];
  double du[NV], fup0 = 0.0;
#pragma unroll
  for (int s = 0; s < NV; ++s) {
    const double fup = wave_shl1(f[s]);
    if (s == 0) fup0 = fup;
    du[perm<1>(s)] = (a1 * fup - a1 * f[s]);
  }
  const bool do_cell = in_run && (lane >= FIRST) && (lane <= 62) && (i >= lo) && (i <= hi);
  if (!do_cell) return;
  if (sp.mflux) {  // mass flux through both x1 faces of this cell (neighbours store the same values)
    double *m = sp.mflux + ((int64_t)0 * u0.nblocks + b) * u0.sn + cell;
    m[0] = f[0];
    m[1] = fup0;
  }
  if constexpr (FINAL) {
    const double vol = b0.dx[0] * b0.dx[1] * b0.dx[2];
    double unused_dt = 0.0;
    double u1v[NV];
    const double *c1 = u1.blocks[b].cons;
#pragma unroll
    for (int n = 0; n < NV; ++n) u1v[n] = c1[n * u0.sn + cell];
    finish_cell<FLUID>(u0, b0, u1v, cell, du, vol, sp, unused_dt);
  } else {
    double *d = sp.du + (int64_t)b * u0.sn * u0.nvar + cell;
#pragma unroll
    for (int n = 0; n < NV; ++n) d[n * u0.sn] = du[n];
  }
}

// ==============================================================================================
// x2 / x3 sweeps: one wave per workgroup marching along the sweep direction
// ==============================================================================================
// Stencil rows c-H .. c+H-1 of the wave's 64 columns live in a private LDS ring
// (ring[slot][var][lane], conflict-free ds_read_b64, no barrier: a lane only ever touches its own
// column); the newest row c+H is loaded from HBM one iteration ahead and sits in registers while
// the Riemann problem of the previous face is solved.  Keeping the ring out of the VGPR file
// (it would be 2H*NV doubles = 72 VGPRs for PPM/GLM-MHD) is what lets two waves share a SIMD.
constexpr int kMarchMinWaves = 2;

// apk_stage_args.prim_from_cons in the two-kernel stage: the sweeps load rows of the CONSERVED input state and convert
// them (ConsToPrim in its lean form, the function the stage that produced the state would have applied: same bits) --
// a row is requested one iteration ahead and converted in registers where it is first used.  The input state is u1's
// (prim_from_cons = 1: stages with gam0 = 0, whose output is another buffer) or u0's (2: the stage then writes its
// result out of place, cons_out_delta != 0, because neighbouring waves still read the old values).
// Returns the APK_FLAG_* bits of the converted cell.  A prim-free RK cycle stores the results of its stages without
// ConsToPrim, so the x3 sweep of the NEXT stage -- whose lanes are interior columns and whose rows are all valid cells
// -- is where a negative density / pressure of that state is first seen: it latches the bits (the reference aborts in
// the FillDerived right after the stage, adiabatic_hydro.hpp:77-79,111-113; the driver reads the word once per cycle).
template <int FLUID>
APK_DEV unsigned cons_row_to_prim(const StageParams &sp, double (&q)[nvars<FLUID>()]) {
  constexpr int NV = nvars<FLUID>();
  double u[NV], w[NV], di;
#pragma unroll
  for (int n = 0; n < NV; ++n) u[n] = q[n];
  const unsigned fl = cons_to_prim_core<FLUID, true>(sp.eos, sp.k.eos_gm1, sp.k.vceil_sq, sp.k.pfloor_over_gm1, u, w, di);
#pragma unroll
  for (int n = 0; n < NV; ++n) q[n] = w[n];
  return fl;
}

template <int FLUID, int RECON>
constexpr int march_lds_bytes() {
  return 2 * recon_halfwidth(RECON) * nvars<FLUID>() * 64 * (int)sizeof(double);
}

template <int FLUID, int RECON, int RS, int DIR, bool FINAL, int EXTRA = EXTRA_NONE, bool FC = false>
__global__ void __launch_bounds__(64, kMarchMinWaves)
fused_march_kernel(PackView u0, PackView u1, StageParams sp, int nseg, int rpw) {
  static_assert(DIR == 2 || DIR == 3, "march is for x2/x3");
  static_assert(FINAL || EXTRA == EXTRA_NONE, "extras belong to the finishing sweep");
  static_assert(!FC || (DIR == 3 && !FINAL), "prim_from_cons: the x3 sweep of the two-kernel stage");
  double lane_min_dt = 1.7976931348623157e308;
  constexpr int NV = nvars<FLUID>();
  constexpr int H = recon_halfwidth(RECON);
  constexpr int NS = 2 * H;  // ring slots
  extern __shared__ __attribute__((aligned(16))) double ring[];
  const int lane = threadIdx.x;
  // Narrow meshblocks (AMR-sized: nx1 = 16 or 32) would leave 3/4 or 1/2 of the wave idle with one
  // row of columns per wave: the wave then takes rpw = 4 or 2 transverse rows, 64 / rpw columns each.
  const int cpr = 64 / rpw;
  const int sub = lane / cpr;
  const int i = u0.is + blockIdx.x * cpr + (lane - sub * cpr);
  const int ntrans = (DIR == 2) ? u0.nx3 : u0.nx2;
  const int trans_raw = blockIdx.y * rpw + sub;  // k for DIR==2, j for DIR==3
  const bool active = (i <= u0.ie) && (trans_raw < ntrans);
  const int ii = (i <= u0.ie) ? i : u0.ie;  // idle lanes shadow a valid column, never store
  const int trans = (trans_raw < ntrans) ? trans_raw : ntrans - 1;
  // the march may be cut into nseg segments (blockIdx.z = block * nseg + segment): few, long
  // waves cannot fill the machine when the pack is small (one 256^3 block is 1024 waves for 2048
  // slots); each segment re-reads 2H+1 rows and redoes one face
  const int b = blockIdx.z / nseg;
  const int seg = blockIdx.z - b * nseg;
  const apk_block_desc b0 = u0.blocks[b];
  const double *c1 = u1.blocks[b].cons;
  // the x3 sweep of the two-kernel stage addresses its rows as scalar pointer + the lane's byte offset (RowCellAt)
  constexpr bool ROWADDR = (DIR == 3 && !FINAL);

  const int64_t st = (DIR == 2) ? u0.sj : u0.sk;
  int s0 = (DIR == 2) ? u0.js : u0.ks;
  int e_all = (DIR == 2) ? u0.je : u0.ke;
  if (sp.window) {  // x3 sweep of a split two-kernel stage: the planes klo..khi of this block only
    const int *w = sp.window + 8 * b;
    if (w[1] <= 0 || w[7] < w[6]) return;
    s0 = w[6];
    e_all = w[7];
  }
  const int n_along = e_all - s0 + 1;
  const int seg_len = (n_along + nseg - 1) / nseg;
  const int s = s0 + seg * seg_len;  // first / last interior index along the march
  const int e = (s + seg_len - 1 < e_all) ? s + seg_len - 1 : e_all;
  if (s > e_all) return;
  const int64_t base = (DIR == 2) ? ((int64_t)(u0.ks + trans) * u0.sk + ii)
                                  : ((int64_t)(u0.js + trans) * u0.sj + ii);
  const double dx = b0.dx[DIR - 1];
  const double area = (DIR == 2) ? b0.dx[0] * b0.dx[2] : b0.dx[0] * b0.dx[1];
  const double vol = b0.dx[0] * b0.dx[1] * b0.dx[2];
  const bool du_compact = !FINAL && sp.du_pitch > 0;  // (wave-uniform)
  const int64_t du_sn = du_compact ? (int64_t)sp.du_pitch * u0.nx2 * u0.nx3 : u0.sn;
  double *dscratch = sp.du + (int64_t)b * du_sn * u0.nvar;
  // the stage's input: u0's primitives, or the conserved state of u1 / u0 (FC, see cons_row_to_prim)
  const apk_block_desc *srcb = (FC && sp.prim_from_cons != 2) ? u1.blocks : u0.blocks;  // (wave-uniform)
  auto input_of = [&](int blk) -> const double * { return FC ? srcb[blk].cons : u0.blocks[blk].prim; };
  const double *in0 = input_of(b);
  const double *prim = in0 + base;
  double *prim_dst = (EXTRA != EXTRA_NONE && sp.prim_to_u1) ? u1.blocks[b].prim : b0.prim;
  // Direct neighbour addressing (sp.face_nbr): the stencil rows below / above the interior come
  // from the interior of the block behind that face (the lanes are interior columns, so the
  // choice is wave-uniform: an offset added to the row's address).
  const int lo_int = (DIR == 2) ? u0.js : u0.ks, hi_int = (DIR == 2) ? u0.je : u0.ke;
  int64_t nbr_lo = 0, nbr_hi = 0;
  if (sp.face_nbr) {
    const int n_int = (DIR == 2) ? u0.nx2 : u0.nx3;
    const int nlo = sp.face_nbr[6 * b + 2 * (DIR - 1)], nhi = sp.face_nbr[6 * b + 2 * (DIR - 1) + 1];
    if (nlo >= 0) nbr_lo = (input_of(nlo) - in0) + (int64_t)n_int * st;
    if (nhi >= 0) nbr_hi = (input_of(nhi) - in0) - (int64_t)n_int * st;
  }
  auto row_off = [&](int r) -> int64_t {
    return (int64_t)r * st + (r < lo_int ? nbr_lo : (r > hi_int ? nbr_hi : (int64_t)0));
  };
  const unsigned base_boff = (unsigned)(base * (int64_t)sizeof(double));
  auto load_row = [&](int r, double (&row)[NV]) {
    if constexpr (ROWADDR) {
      load_vars<NV>(uniform_ptr(in0), u0.sn, RowCellAt{row_off(r), base_boff}, row);
    } else {
#pragma unroll
      for (int n = 0; n < NV; ++n) row[n] = prim[n * u0.sn + row_off(r)];
    }
  };

  // row r of the stencil lives in slot (r - (s-1-H)) mod NS
  int c = s - 1;
  const int r0 = c - H;
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    double row[NV];
    load_row(r0 + m, row);
    if constexpr (FC) {
      const unsigned fl = cons_row_to_prim<FLUID>(sp, row);
      if (active && fl) atomicOr(sp.flags, fl);
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) ring[(m * NV + n) * 64 + lane] = row[n];
  }
  double Pn[NV];  // row c+H (FC: as loaded until the top of the iteration that uses it)
  load_row(c + H, Pn);

  double wl_prev[NV];  // permuted L state at face c (from cell c-1)
  double f_prev[NV];   // permuted flux at face c-1
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    wl_prev[q] = 0.0;
    f_prev[q] = 0.0;
  }
  // PPM: the limited interface value above cell c is the one below cell c+1 (see ppm_interface):
  // every interface is evaluated once and carried to the next iteration
  double face_carry[NV];
  if constexpr (RECON == APK_RC_PPM) {
#pragma unroll
    for (int n = 0; n < NV; ++n)
      face_carry[n] = ppm_interface(ring[(0 * NV + n) * 64 + lane], ring[(1 * NV + n) * 64 + lane],
                                    ring[(2 * NV + n) * 64 + lane], ring[(3 * NV + n) * 64 + lane]);
  }

  int slot0 = 0;  // slot holding row c-H
  for (; c <= e + 1; ++c) {
    // the streaming operands of the cell that completes in this iteration (c-1) are requested
    // first, so they are in flight during the reconstruction and the Riemann solve
    const int64_t done = base + (int64_t)(c - 1) * st;
    if constexpr (FC) {
      const unsigned fl = cons_row_to_prim<FLUID>(sp, Pn);
      if (active && fl) atomicOr(sp.flags, fl);
    }
    double duv[NV], u1v[NV];
    if (c >= s + 1) {
      if (FINAL || !sp.du_first) {
#pragma unroll
        for (int n = 0; n < NV; ++n) duv[n] = dscratch[n * u0.sn + done];
      }
      if constexpr (FINAL) {
#pragma unroll
        for (int n = 0; n < NV; ++n) u1v[n] = c1[n * u0.sn + done];
      }
    }
    // reconstruct cell c from ring rows c-H..c+H-1 and the register row c+H.  The ring rows of
    // variable n+1 are requested before variable n is processed (software-pipelined LDS reads: the
    // branchy limiter code keeps the compiler from hoisting them by itself).
    double qln[NV], qrn[NV];
    double an[NS > 0 ? NS : 1];
    if constexpr (NS > 0) {
#pragma unroll
      for (int m = 0; m < NS; ++m) an[m] = ring[(((slot0 + m) & (NS - 1)) * NV + 0) * 64 + lane];
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      double a[NS > 0 ? NS : 1];
      if constexpr (NS > 0) {
#pragma unroll
        for (int m = 0; m < NS; ++m) a[m] = an[m];
        if (n + 1 < NV) {
#pragma unroll
          for (int m = 0; m < NS; ++m) an[m] = ring[(((slot0 + m) & (NS - 1)) * NV + n + 1) * 64 + lane];
        }
      }
      if constexpr (H == 0) {
        reconstruct<RECON>(0.0, 0.0, Pn[n], 0.0, 0.0, dx, n, qln[n], qrn[n]);
      } else if constexpr (H == 1) {
        const double a0 = a[0], a1 = a[1];
        reconstruct<RECON>(0.0, a0, a1, Pn[n], 0.0, dx, n, qln[n], qrn[n]);
      } else {
        const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
        if constexpr (RECON == APK_RC_PPM) {
          const double face_p = ppm_interface(a1, a2, a3, Pn[n]);
          ppm_cell(a0, a1, a2, a3, Pn[n], face_carry[n], face_p, qln[n], qrn[n]);
          face_carry[n] = face_p;
        } else {
          reconstruct<RECON>(a0, a1, a2, a3, Pn[n], dx, n, qln[n], qrn[n]);
        }
      }
      // The 9 reconstructions are independent; left alone the scheduler interleaves them all and
      // the weighted schemes (WENO-Z: 5 divisions and ~40 live values per call) spill.  An empty
      // volatile asm pins each result before the next variable starts: less ILP, no scratch.
      if constexpr (RECON == APK_RC_WENOZ || RECON == APK_RC_WENO3 || RECON == APK_RC_LIMO3)
        asm volatile("" : "+v"(qln[n]), "+v"(qrn[n]));
    }
    // row c+H replaces row c-H in the ring; fetch row c+1+H for the next iteration now so the
    // loads fly while the Riemann problem below is solved
    const bool more = (c < e + 1);
    if (more) {
      if constexpr (NS > 0) {
#pragma unroll
        for (int n = 0; n < NV; ++n) ring[(slot0 * NV + n) * 64 + lane] = Pn[n];
        slot0 = (slot0 + 1) & (NS - 1);
      }
      load_row(c + 1 + H, Pn);
    }
    if (c >= s) {
      double wr[NV], f[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) wr[q] = qrn[perm<DIR>(q)];
      riemann<FLUID, RS>(wl_prev, wr, sp.k, f);
      if (c >= s + 1) {
        // cell c-1 is complete: (A F(c) - A F(c-1)) joins du
        const int64_t cell = done;
        double du[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<DIR>(q);
          const double diff = (area * f[q] - area * f_prev[q]);
          du[n] = (!FINAL && sp.du_first) ? diff : duv[n] + diff;
        }
        if (active && sp.mflux) {
          double *m = sp.mflux + ((int64_t)(DIR - 1) * u0.nblocks + b) * u0.sn + cell;
          m[0] = f_prev[0];
          m[st] = f[0];
        }
        if (active) {
          if constexpr (FINAL) {
            finish_cell<FLUID, EXTRA>(u0, b0, u1v, cell, du, vol, sp, lane_min_dt, prim_dst);
          } else {
            // (du_compact: DIR = 3, the row (k, j) = (c - 1, trans) of the interior, column ii - is)
            if constexpr (ROWADDR) {
              // (the lane's share: its row `trans` of the plane and its column; the plane is wave-uniform)
              const RowCellAt at{du_compact ? (int64_t)(c - 1 - u0.ks) * u0.nx2 * sp.du_pitch : (int64_t)(c - 1) * st,
                                 du_compact ? (unsigned)(((int64_t)trans * sp.du_pitch + (ii - u0.is)) * (int64_t)sizeof(double)) : base_boff};
              store_vars<NV>(uniform_ptr(dscratch), du_sn, at, du);
            } else {
              const int64_t dcell = du_compact ? ((int64_t)(c - 1 - u0.ks) * u0.nx2 + trans) * sp.du_pitch + (ii - u0.is) : cell;
#pragma unroll
              for (int n = 0; n < NV; ++n) store_result(&dscratch[n * du_sn + dcell], du[n]);
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) f_prev[q] = f[q];
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) wl_prev[q] = qln[perm<DIR>(q)];
  }
  if constexpr (EXTRA == EXTRA_C2P_DT) {
    // one atomic per wave for the whole march (positive doubles order like their bit patterns)
    double m = lane_min_dt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if (lane == 0) atomicMin(sp.dt_bits, (unsigned long long)__double_as_longlong(m));
  }
}

// ==============================================================================================
// 3-D: x1 and x2 sweeps in ONE march (saves a full round trip of the du array and a re-read of
// prim).  Lanes are the flattened (k, i) cells of one j-row of a block, Ni per k (ghost columns
// included so a row edge never needs a neighbour wave), 62 useful lanes per wave as in the x1
// sweep; the wave marches along j.  Row c of the march is the x1 pencil: its fluxes are exchanged
// with DPP wave shifts, its flux difference waits one iteration in registers until the x2 flux
// at face c+1 is known, then du = (x1 term + x2 term) is stored in the reference's order.
// ==============================================================================================
template <int FLUID, int RECON, int RS>
__global__ void __launch_bounds__(64, kMarchMinWaves)
fused_march12_kernel(PackView u0, PackView u1, StageParams sp, int waves_per_block) {
  constexpr int NV = nvars<FLUID>();
  constexpr int H = recon_halfwidth(RECON);
  constexpr int NS = 2 * H;
  extern __shared__ __attribute__((aligned(16))) double ring[];
  const int lane = threadIdx.x;
  const int b = blockIdx.z;
  const apk_block_desc b0 = u0.blocks[b];

  const int64_t run = (int64_t)u0.nx3 * u0.ni;
  const int64_t t = (int64_t)blockIdx.x * 62 + lane - 1;
  const bool in_run = (t >= 0) && (t < run);
  const int64_t tc = in_run ? t : (t < 0 ? 0 : run - 1);  // out-of-run lanes shadow a valid column
  const int krow = (int)(tc / u0.ni);
  const int i = (int)(tc - (int64_t)krow * u0.ni);
  const bool x1_recon = in_run && (i >= u0.is - 1) && (i <= u0.ie + 1);
  // the cell column this lane retires (needs the lane on its left and on its right)
  const bool active = in_run && (lane >= 1) && (lane <= 62) && (i >= u0.is) && (i <= u0.ie);

  const int64_t st = u0.sj;
  const int s = u0.js, e = u0.je;
  const int64_t base = (int64_t)(u0.ks + krow) * u0.sk + i;
  const double dx1 = b0.dx[0], dx2 = b0.dx[1];
  const double area1 = b0.dx[1] * b0.dx[2];
  const double area2 = b0.dx[0] * b0.dx[2];
  double *dscratch = sp.du + (int64_t)b * u0.sn * u0.nvar;
  const double *prim = b0.prim + base;

  int c = s - 1;
  const int r0 = c - H;
#pragma unroll
  for (int m = 0; m < NS; ++m)
#pragma unroll
    for (int n = 0; n < NV; ++n) ring[(m * NV + n) * 64 + lane] = prim[n * u0.sn + (int64_t)(r0 + m) * st];
  double Pn[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) Pn[n] = prim[n * u0.sn + (int64_t)(c + H) * st];

  double wl_prev[NV], f_prev[NV], du1_prev[NV];  // x2 L state / x2 flux / x1 flux difference of row c-1
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    wl_prev[q] = 0.0;
    f_prev[q] = 0.0;
    du1_prev[q] = 0.0;
  }

  int slot0 = 0;
  for (; c <= e + 1; ++c) {
    // ---- x2: reconstruct cell c along j ---------------------------------------------------------
    double qln[NV], qrn[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      if constexpr (H == 0) {
        reconstruct<RECON>(0.0, 0.0, Pn[n], 0.0, 0.0, dx2, n, qln[n], qrn[n]);
      } else if constexpr (H == 1) {
        const double a0 = ring[(slot0 * NV + n) * 64 + lane];
        const double a1 = ring[(((slot0 + 1) & 1) * NV + n) * 64 + lane];
        reconstruct<RECON>(0.0, a0, a1, Pn[n], 0.0, dx2, n, qln[n], qrn[n]);
      } else {
        const double a0 = ring[(slot0 * NV + n) * 64 + lane];
        const double a1 = ring[(((slot0 + 1) & 3) * NV + n) * 64 + lane];
        const double a2 = ring[(((slot0 + 2) & 3) * NV + n) * 64 + lane];
        const double a3 = ring[(((slot0 + 3) & 3) * NV + n) * 64 + lane];
        reconstruct<RECON>(a0, a1, a2, a3, Pn[n], dx2, n, qln[n], qrn[n]);
      }
    }
    const bool more = (c < e + 1);
    if (more) {
      if constexpr (NS > 0) {
#pragma unroll
        for (int n = 0; n < NV; ++n) ring[(slot0 * NV + n) * 64 + lane] = Pn[n];
        slot0 = (slot0 + 1) & (NS - 1);
      }
#pragma unroll
      for (int n = 0; n < NV; ++n) Pn[n] = prim[n * u0.sn + (int64_t)(c + 1 + H) * st];
    }
    if (c >= s) {
      double wr[NV], f[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) wr[q] = qrn[perm<2>(q)];
      riemann<FLUID, RS>(wl_prev, wr, sp.k, f);
      if (c >= s + 1 && active) {
        // cell c-1: du = (x1 term) + (x2 term), the reference's accumulation order
        const int64_t cell = base + (int64_t)(c - 1) * st;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<2>(q);
          dscratch[n * u0.sn + cell] = du1_prev[n] + (area2 * f[q] - area2 * f_prev[q]);
        }
        if (sp.mflux) {
          double *m = sp.mflux + ((int64_t)1 * u0.nblocks + b) * u0.sn + cell;
          m[0] = f_prev[0];
          m[st] = f[0];
        }
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) f_prev[q] = f[q];
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) wl_prev[q] = qln[perm<2>(q)];

    // ---- x1: row c is an interior row -> its x1 flux difference (all lanes take part in the DPP) --
    if (c >= s && c <= e) {  // wave-uniform
      double ql1[NV], qr1[NV];
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        double qm2 = 1.0, qm1 = 1.0, q0 = 1.0, qp1 = 1.0, qp2 = 1.0;
        if (x1_recon) {
          const double *cc = prim + n * u0.sn + (int64_t)c * st;
          q0 = cc[0];
          if constexpr (H >= 1) {
            qm1 = cc[-1];
            qp1 = cc[1];
          }
          if constexpr (H >= 2) {
            qm2 = cc[-2];
            qp2 = cc[2];
          }
        }
        reconstruct<RECON>(qm2, qm1, q0, qp1, qp2, dx1, n, ql1[n], qr1[n]);
      }
      double wl[NV], wr[NV], f[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        wl[q] = wave_shr1(ql1[perm<1>(q)]);
        wr[q] = qr1[perm<1>(q)];
      }
      riemann<FLUID, RS>(wl, wr, sp.k, f);
      double fup0 = 0.0;
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const double fup = wave_shl1(f[q]);
        if (q == 0) fup0 = fup;
        du1_prev[perm<1>(q)] = (area1 * fup - area1 * f[q]);
      }
      if (sp.mflux && active) {
        double *m = sp.mflux + ((int64_t)0 * u0.nblocks + b) * u0.sn + base + (int64_t)c * st;
        m[0] = f[0];
        m[1] = fup0;
      }
    }
  }
}

// ==============================================================================================
// 3-D donor cell (the VL2 predictor): the WHOLE stage in one march, no du array at all.
// A donor-cell face flux depends on the two adjacent cells only, so one lane can own a cell
// column and produce all six of its face fluxes: lanes are the flattened (j, i) cells of the
// interior rows of a block (62 useful lanes per wave, DPP wave shifts for the x1 faces as in the
// x1 sweep), the wave marches along k carrying the lower x3 flux and the (x1 + x2) flux
// difference of the previous plane in registers, and both x2 faces of a cell are solved by its
// own lane from the rows j-1, j, j+1 (4 Riemann problems per cell instead of 3; the second x2
// solve is bit-identical to the neighbour lane's, so the scheme stays conservative to the last
// bit).  Traffic: prim in (the j+-1 rows hit L2), u1 in, u0 out = the algorithmic 216 B/cell
// against 504 B/cell of the two-march schedule.  prim is only read, so no lane can observe a
// half-updated state; FillDerived of the stage is left to ConservedToPrimitive.
// ==============================================================================================
// Input state of a donor-cell march lane: nine loads at `p` and, when the stage derives its primitives from the
// conserved state (apk_stage_args.prim_from_cons), ConsToPrim of what was loaded -- the function the finishing sweep of the
// previous stage applied to the same values (lean form: no floor but the density / energy ones, no flags: the state was
// checked when it was produced).
// the NV variables of one cell when the stride between them is the LANE's (x1_halo: ghost-column lanes read a receive
// segment): a chain of per-lane additions, pinned -- otherwise the multiples n * sn are kept in registers across the march
template <int NV>
APK_DEV void load_chain(const double *p, int64_t lane_sn, double (&v)[NV]) {
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    v[n] = *p;
    if (n + 1 < NV) {
      p += lane_sn;
      asm volatile("" : "+v"(p));
    }
  }
}
template <int FLUID, bool FROM_CONS, bool LANE_SN = false>
APK_DEV void load_input_state(const double *p, int64_t sn, const StageParams &sp, double (&w)[nvars<FLUID>()]) {
  constexpr int NV = nvars<FLUID>();
  if constexpr (FROM_CONS) {
    double u[NV], di;
    if constexpr (LANE_SN) {
      load_chain<NV>(p, sn, u);
    } else {
#pragma unroll
    for (int n = 0; n < NV; ++n) u[n] = p[n * sn];
    }
    (void)cons_to_prim_core<FLUID, true>(sp.eos, sp.k.eos_gm1, sp.k.vceil_sq, sp.k.pfloor_over_gm1, u, w, di);
  } else if constexpr (LANE_SN) {
    load_chain<NV>(p, sn, w);
  } else {
#pragma unroll
    for (int n = 0; n < NV; ++n) w[n] = p[n * sn];
  }
}

template <int FLUID, int RS, int EXTRA = EXTRA_NONE, int LEAN = 0, bool FROM_CONS = false>
__global__ void __launch_bounds__(64, 2)
fused_dc3_kernel(PackView u0, PackView u1, StageParams sp, int kseg, int wpb, int nseg, int per_xcd) {
  constexpr int NV = nvars<FLUID>();
  double lane_min_dt = 1.7976931348623157e308;
  const int lane = threadIdx.x;
  // XCD-aware order: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each
  // with its own L2.  A lane re-reads the rows j-1 and j+1 that the workgroups two chunks away
  // load as their own row, so neighbouring chunks should share an L2: XCD x gets the contiguous
  // range [x * per_xcd, (x + 1) * per_xcd) of the (chunk, k segment, block) order.
  const int vid = (int)(blockIdx.x % 8u) * per_xcd + (int)(blockIdx.x / 8u);
  if (vid >= wpb * nseg * u0.nblocks) return;
  const int chunk = vid % wpb;
  const int segid = (vid / wpb) % nseg;
  const int b = vid / (wpb * nseg);
  const apk_block_desc b0 = u0.blocks[b];
  const double *c1 = u1.blocks[b].cons;
  double *prim_dst = u1.blocks[b].prim;  // EXTRA != NONE only (never in place here)

  // (a donor-cell flux reads one cell either side: the flattened rows need one ghost column each, not nghost --
  // 16 of 18 lanes on interior cells of a 16-cell AMR block with nghost = 4 instead of 16 of 24)
  int i0 = u0.is - 1, rl = u0.nx1 + 2, ilo = u0.is, ihi = u0.ie, jlo = u0.js, jhi = u0.je, klo = u0.ks, khi = u0.ke;
  if (sp.window) {  // split around a halo exchange in flight (apk_stage_args.window)
    const int *w = sp.window + 8 * b;
    i0 = w[0], rl = w[1], ilo = w[2], ihi = w[3], jlo = w[4], jhi = w[5], klo = w[6], khi = w[7];
    if (rl <= 0 || jhi < jlo || khi < klo) return;
  }
  const int64_t run = (int64_t)(jhi - jlo + 1) * rl;
  const int64_t t = (int64_t)chunk * 62 + lane - 1;
  if ((int64_t)chunk * 62 - 1 >= run) return;  // whole wave beyond this block's run
  const bool in_run = (t >= 0) && (t < run);
  const int64_t tc = in_run ? t : (t < 0 ? 0 : run - 1);  // out-of-run lanes shadow a valid column
  const int row = (int)(tc / rl);
  const int i = i0 + (int)(tc - (int64_t)row * rl);
  const bool active = in_run && (lane >= 1) && (lane <= 62) && (i >= ilo) && (i <= ihi);

  const int64_t col = (int64_t)(jlo + row) * u0.sj + i;
  // apk_stage_args.cons_store (lean form): does this lane's column lie in the nghost-deep shell of its block?
  const int jrow = jlo + row;
  const bool shell_ij = (i < u0.is + u0.ng) || (i > u0.ie - u0.ng) || (jrow < u0.js + u0.ng) || (jrow > u0.je - u0.ng);
  static_assert(!FROM_CONS || LEAN, "prim_from_cons: lean form only");
  // the stage's input: u0's primitives, or u1's conserved state (apk_stage_args.prim_from_cons, see load_input_state)
  auto input_of = [&](int blk) -> const double * { return FROM_CONS ? u1.blocks[blk].cons : u0.blocks[blk].prim; };
  const double *prim = input_of(b) + col;
  // Direct neighbour addressing (sp.face_nbr): a lane on a ghost column reads the x1 neighbour's
  // interior column instead, the x2 / x3 neighbours of an interior column come from the block
  // behind that face.  Only face neighbours are ever needed: a donor-cell flux reads the two cells
  // of its face, and lanes on ghost columns retire nothing (their own x2 / x3 neighbours may be
  // stale ghost cells: valid memory, unused results).
  const double *prim_jm = prim - u0.sj, *prim_jp = prim + u0.sj;
  const double *prim_klo = prim, *prim_khi = prim;  // base of the planes below ks / above ke
  if (sp.face_nbr) {
    const int *fn = sp.face_nbr + 6 * b;
    const int j = jlo + row;
    if (i < u0.is || i > u0.ie) {
      const int nb = fn[i < u0.is ? 0 : 1];
      if (nb >= 0) prim = input_of(nb) + col + (i < u0.is ? u0.nx1 : -u0.nx1);
      prim_jm = prim - u0.sj;
      prim_jp = prim + u0.sj;
      prim_klo = prim_khi = prim;
    } else {
      if (j - 1 < u0.js && fn[2] >= 0) prim_jm = input_of(fn[2]) + col + (int64_t)(u0.nx2 - 1) * u0.sj;
      if (j + 1 > u0.je && fn[3] >= 0) prim_jp = input_of(fn[3]) + col - (int64_t)(u0.nx2 - 1) * u0.sj;
      if (fn[4] >= 0) prim_klo = input_of(fn[4]) + col + (int64_t)u0.nx3 * u0.sk;
      if (fn[5] >= 0) prim_khi = input_of(fn[5]) + col - (int64_t)u0.nx3 * u0.sk;
    }
  }
  const double area1 = b0.dx[1] * b0.dx[2], area2 = b0.dx[0] * b0.dx[2], area3 = b0.dx[0] * b0.dx[1];
  const double vol = b0.dx[0] * b0.dx[1] * b0.dx[2];
  const double upd = LEAN ? update_coefficient(sp, vol) : 0.0;
  // the march is cut into gridDim.y segments of kseg planes: 2216 full-length waves on a machine
  // with 2048 wave slots (256 VGPRs -> 2 per SIMD) would run in two rounds; many short waves
  // keep every slot busy, for one redundant x3 solve per segment
  const int s = klo + segid * kseg;
  if (s > khi) return;
  const int e = (s + kseg - 1 < khi) ? s + kseg - 1 : khi;

  // State carried from plane to plane: the previous plane's primitives stay in VGPRs; the lower
  // x3 flux and the (x1 + x2) flux difference wait in a private LDS stash (stash[slot][var][lane],
  // conflict-free, no barrier: a lane only touches its own column), which keeps the kernel under
  // 128 VGPRs = 4 waves per SIMD while an HLLD solve is in flight.
  extern __shared__ __attribute__((aligned(16))) double stash[];
  double *st_f3 = stash + lane;            // [q * 64]: permuted x3 flux at face c-1
  double *st_du = stash + NV * 64 + lane;  // [n * 64]: (x1 term + x2 term) of plane c-1, natural order
  double wprev[NV];                        // natural-order state of plane c-1
  load_input_state<FLUID, FROM_CONS>(((s - 1 < u0.ks) ? prim_klo : prim) + (int64_t)(s - 1) * u0.sk, u0.sn, sp, wprev);
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    st_f3[n * 64] = 0.0;
    st_du[n * 64] = 0.0;
  }
  // (The four Riemann problems of a cell are independent and the scheduler interleaves them; with
  // the carried state in the stash that fits in 199 VGPRs without scratch.)
  // GLM-MHD HLLD: a donor-cell state is the L state of one face and the R state of the next in every
  // direction, so its fast speed along each direction is evaluated ONCE (glmmhd_hlld_cf) -- carried to the
  // next plane for x3, handed one lane to the right for x1, shared by the two x2 solves of the cell:
  // 5 evaluations (two square roots each) per cell instead of 8, the same values bit for bit.
  constexpr bool CF = (FLUID == APK_FLUID_GLMMHD) && (RS == APK_RS_HLLD);
  auto cf_of = [&](const double (&w)[NV]) -> double {
    if constexpr (CF) return fast_speed(sp.gamma, w[IDN], w[IPR], w[IB1], w[IB2], w[IB3]);
    else return 0.0;
  };
  auto solve = [&](const double (&wl)[NV], const double (&wr)[NV], double cfl, double cfr, double (&f)[NV]) {
    if constexpr (CF) glmmhd_hlld_cf(wl, wr, sp.k, cfl, cfr, f);
    else riemann<FLUID, RS>(wl, wr, sp.k, f);
  };
  double cf3_prev = 0.0;
  if constexpr (CF) {
    double wp3[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) wp3[q] = wprev[perm<3>(q)];
    cf3_prev = cf_of(wp3);
  }
  // The next plane is requested one iteration ahead (its 9 loads have a whole iteration of four Riemann
  // solves to land: 1.69 -> 1.62 ms on 8 x 128^3) where the registers allow it: with FillDerived in the
  // kernel 243 VGPRs; without (refined meshes) the 18 extra registers would spill.
  constexpr bool PF = (EXTRA != EXTRA_NONE) && !FROM_CONS;
  double wnext[NV];
  if constexpr (PF) {
#pragma unroll
    for (int n = 0; n < NV; ++n) wnext[n] = ((s > u0.ke) ? prim_khi : prim)[n * u0.sn + (int64_t)s * u0.sk];
  }
  for (int c = s; c <= e + 1; ++c) {
    const int64_t off = (int64_t)c * u0.sk;
    double wc[NV];
    if constexpr (PF) {
#pragma unroll
      for (int n = 0; n < NV; ++n) wc[n] = wnext[n];
      if (c <= e) {
        const double *pn = (c + 1 > u0.ke) ? prim_khi : prim;  // wave-uniform choice
#pragma unroll
        for (int n = 0; n < NV; ++n) wnext[n] = pn[n * u0.sn + off + u0.sk];
      }
    } else {
      const double *pc = (c > u0.ke) ? prim_khi : prim;  // wave-uniform choice (c >= s >= ks)
      load_input_state<FLUID, FROM_CONS>(pc + off, u0.sn, sp, wc);
    }
    // ---- x3 face c (between planes c-1 and c)
    {
      double wl3[NV], wr3[NV], f3[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        wl3[q] = wprev[perm<3>(q)];
        wr3[q] = wc[perm<3>(q)];
      }
      const double cf3_c = cf_of(wr3);
      solve(wl3, wr3, cf3_prev, cf3_c, f3);
      cf3_prev = cf3_c;
      if (c >= s + 1) {
        const int64_t done = col + (int64_t)(c - 1) * u0.sk;
        double du[NV], u1v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<3>(q);
          du[n] = st_du[n * 64] + (area3 * f3[q] - area3 * st_f3[q * 64]);
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) u1v[n] = c1[n * u0.sn + done];
        if constexpr (!LEAN) {
          if (active && sp.mflux) {
            double *m = sp.mflux + ((int64_t)2 * u0.nblocks + b) * u0.sn + done;
            m[0] = st_f3[0];
            m[u0.sk] = f3[0];
          }
        }
        bool store = true;
        if constexpr (LEAN)
          store = sp.cons_store == 0 || (sp.cons_store == 1 && (shell_ij || c - 1 < u0.ks + u0.ng || c - 1 > u0.ke - u0.ng));
        if (active) finish_cell<FLUID, EXTRA, LEAN>(u0, b0, u1v, done, du, vol, sp, lane_min_dt, prim_dst, upd, store);
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) st_f3[q * 64] = f3[q];
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) wprev[n] = wc[n];
    if (c <= e) {  // wave-uniform
      // ---- x1 faces of plane c: L state from the lane on the left, upper flux from the right
      {
        double wl[NV], wr[NV], f[NV], d1[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wr[q] = wc[perm<1>(q)];
          wl[q] = wave_shr1(wr[q]);
        }
        const double cf1_c = cf_of(wr);
        solve(wl, wr, CF ? wave_shr1(cf1_c) : 0.0, cf1_c, f);
        double fup0 = 0.0;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const double fup = wave_shl1(f[q]);
          if (q == 0) fup0 = fup;
          d1[q] = (area1 * fup - area1 * f[q]);
        }
        if constexpr (!LEAN) {
          if (active && sp.mflux) {
            double *m = sp.mflux + ((int64_t)0 * u0.nblocks + b) * u0.sn + col + off;
            m[0] = f[0];
            m[1] = fup0;
          }
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) st_du[perm<1>(q) * 64] = d1[q];
      }
      // ---- x2 faces j and j+1 of this cell
      double flo[NV];
      double cf2_c = 0.0;
      {
        double wm[NV], w2[NV], wnat[NV];
        load_input_state<FLUID, FROM_CONS>(prim_jm + off, u0.sn, sp, wnat);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wm[q] = wnat[perm<2>(q)];
          w2[q] = wc[perm<2>(q)];
        }
        cf2_c = cf_of(w2);
        solve(wm, w2, cf_of(wm), cf2_c, flo);
      }
      {
        double wp[NV], w2[NV], fhi[NV], wnat[NV];
        load_input_state<FLUID, FROM_CONS>(prim_jp + off, u0.sn, sp, wnat);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wp[q] = wnat[perm<2>(q)];
          w2[q] = wc[perm<2>(q)];
        }
        solve(w2, wp, cf2_c, cf_of(wp), fhi);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<2>(q);
          st_du[n * 64] = st_du[n * 64] + (area2 * fhi[q] - area2 * flo[q]);
        }
        if constexpr (!LEAN) {
          if (active && sp.mflux) {
            double *m = sp.mflux + ((int64_t)1 * u0.nblocks + b) * u0.sn + col + off;
            m[0] = flo[0];
            m[u0.sj] = fhi[0];
          }
        }
      }
    }
  }
  if constexpr (EXTRA == EXTRA_C2P_DT) {
    double m = lane_min_dt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if (lane == 0) atomicMin(sp.dt_bits, (unsigned long long)__double_as_longlong(m));
  }
}

// ==============================================================================================
// The 3-D donor-cell stage with TWO j-rows per lane (lean stages on whole blocks with an even number of rows).
//
// fused_dc3_kernel solves four Riemann problems per cell: x1 faces are shared between neighbouring lanes (DPP), x3
// faces are carried along the march, but BOTH x2 faces are solved by the lane itself -- the lane that owns the cell
// across the face sits 130 lanes away in another wave.  Here a lane owns the cells (2m, i) and (2m + 1, i): the x2 face
// between them is solved once and used twice, 3 x2 solves for 2 cells, i.e. 3.5 solves per cell instead of 4
// (-12.5 % of a kernel that is Riemann solves and little else) and 4 fast speeds along x2 for 2 cells instead of 6.
// Same values: the shared flux is the one both cells computed for themselves before (same function of the same two
// states), the accumulation order per cell is unchanged.  The two cells are processed one after the other, so the
// working set of a solve stays what it was; the carried state doubles: 2 x 9 doubles of the previous plane in VGPRs,
// 4 x 9 rows in the LDS stash (18.4 KB per wave, inside the 20 KB two waves per SIMD leave).
// ==============================================================================================
// X1H (apk_stage_args.x1_halo): the lanes on the two ghost columns read their cells from the receive segment of that face
// -- strides of their own between variables, rows and planes -- and the lanes within `send_depth` columns of a face with
// a send segment store what they retire a second time, into the segment (see fused_m12f_kernel).
template <int FLUID, int RS, int EXTRA, bool FROM_CONS = false, bool X1H = false, int LEAN = 1>
__global__ void __launch_bounds__(64, 2)
fused_dc3r2_kernel(PackView u0, PackView u1, StageParams sp, int kseg, int wpb, int nseg, int per_xcd) {
  static_assert(LEAN == 1 || (LEAN == LEAN_PFLOOR && !FROM_CONS && !X1H && EXTRA != EXTRA_NONE), "lean forms");
  constexpr int NV = nvars<FLUID>();
  double lane_min_dt = 1.7976931348623157e308;
  const int lane = threadIdx.x;
  const int vid = (int)(blockIdx.x % 8u) * per_xcd + (int)(blockIdx.x / 8u);  // XCD-aware order (see fused_dc3_kernel)
  if (vid >= wpb * nseg * u0.nblocks) return;
  const int chunk = vid % wpb;
  const int segid = (vid / wpb) % nseg;
  const int b = vid / (wpb * nseg);
  const apk_block_desc b0 = u0.blocks[b];
  const double *c1 = u1.blocks[b].cons;
  double *prim_dst = u1.blocks[b].prim;

  const int i0 = u0.is - 1, rl = u0.nx1 + 2;
  const int64_t run = (int64_t)(u0.nx2 / 2) * rl;
  const int64_t t = (int64_t)chunk * 62 + lane - 1;
  if ((int64_t)chunk * 62 - 1 >= run) return;
  const bool in_run = (t >= 0) && (t < run);
  const int64_t tc = in_run ? t : (t < 0 ? 0 : run - 1);
  const int rowpair = (int)(tc / rl);
  const int i = i0 + (int)(tc - (int64_t)rowpair * rl);
  const bool active = in_run && (lane >= 1) && (lane <= 62) && (i >= u0.is) && (i <= u0.ie);
  const int ja = u0.js + 2 * rowpair;  // rows ja (cell A) and ja + 1 (cell B)

  const int64_t col = (int64_t)ja * u0.sj + i;
  const bool shell_i = (i < u0.is + u0.ng) || (i > u0.ie - u0.ng);
  const bool shell_ij[2] = {shell_i || (ja < u0.js + u0.ng) || (ja > u0.je - u0.ng),
                            shell_i || (ja + 1 < u0.js + u0.ng) || (ja + 1 > u0.je - u0.ng)};  // (apk_stage_args.cons_store)
  auto input_of = [&](int blk) -> const double * { return FROM_CONS ? u1.blocks[blk].cons : u0.blocks[blk].prim; };  // (see fused_dc3_kernel)
  const double *prim = input_of(b) + col;  // cell A's column; cell B's is prim + sj
  const double *prim_jm = prim - u0.sj, *prim_jp = prim + 2 * u0.sj;
  const double *prim_klo = prim, *prim_khi = prim;
  if (sp.face_nbr) {  // direct neighbour addressing, as in fused_dc3_kernel
    const int *fn = sp.face_nbr + 6 * b;
    if (i < u0.is || i > u0.ie) {
      const int nb = fn[i < u0.is ? 0 : 1];
      if (nb >= 0) prim = input_of(nb) + col + (i < u0.is ? u0.nx1 : -u0.nx1);
      prim_jm = prim - u0.sj;
      prim_jp = prim + 2 * u0.sj;
      prim_klo = prim_khi = prim;
    } else {
      if (ja - 1 < u0.js && fn[2] >= 0) prim_jm = input_of(fn[2]) + col + (int64_t)(u0.nx2 - 1) * u0.sj;
      if (ja + 2 > u0.je && fn[3] >= 0) prim_jp = input_of(fn[3]) + col + 2 * u0.sj - (int64_t)u0.nx2 * u0.sj;
      if (fn[4] >= 0) prim_klo = input_of(fn[4]) + col + (int64_t)u0.nx3 * u0.sk;
      if (fn[5] >= 0) prim_khi = input_of(fn[5]) + col - (int64_t)u0.nx3 * u0.sk;
    }
  }
  // x1_halo, receive side: [var][k][j][depth]; the ghost column next to the face is the last of the lower segment's
  // columns, the first of the upper one's.  Planes and rows outside the interior (read by every lane for the x2 / x3
  // faces of its own cells, which ghost-column lanes do not retire) repeat the nearest one: valid addresses.
  [[maybe_unused]] int64_t lsn = u0.sn;  // (X1H: per lane)
  [[maybe_unused]] int lsk = (int)u0.sk, lsj = (int)u0.sj;  // (a block's plane fits 31 bits: two_kernel_stage_applies' rule for sn)
  [[maybe_unused]] bool gseg = false;
#ifndef APK_X1H_NO_RECV
  if constexpr (X1H) {
    if (sp.x1_blocks && sp.x1_recv_depth > 0 && (i < u0.is || i > u0.ie)) {
      const int dpt = sp.x1_recv_depth, side = (i < u0.is) ? 0 : 1;
      const double *seg = sp.x1_blocks[b].recv[side];
      if (seg) {
        lsj = dpt;
        lsk = dpt * u0.nx2;
        lsn = (int64_t)lsk * u0.nx3;
        // (plane c of row ja is prim + c * lsk, like everywhere else: the segment's first plane is ks)
        prim = seg + (int64_t)(ja - u0.js) * lsj + (side ? 0 : dpt - 1) - (int64_t)u0.ks * lsk;
        prim_jm = (ja - 1 < u0.js) ? prim : prim - lsj;
        prim_jp = (ja + 2 > u0.je) ? prim + lsj : prim + 2 * lsj;
        prim_klo = prim_khi = prim;
        gseg = true;
      }
    }
  }
#endif
  auto plane = [&](int c) -> int {  // (wave-uniform candidates, chosen per lane)
    if constexpr (X1H) return gseg ? ((c < u0.ks) ? u0.ks : ((c > u0.ke) ? u0.ke : c)) : c;
    else return c;
  };
  // x1_halo, send side: cell A's place; cell B's is one row (lsj of the segment = send_depth) further
  [[maybe_unused]] X1Store xs;
#ifndef APK_X1H_NO_SEND
  if constexpr (X1H) xs = x1_store_of<false>(sp, u0, b, i, active, (int64_t)(ja - u0.js) * sp.x1_send_depth, (int64_t)sp.x1_send_depth * u0.nx2);
#endif
  const double area1 = b0.dx[1] * b0.dx[2], area2 = b0.dx[0] * b0.dx[2], area3 = b0.dx[0] * b0.dx[1];
  const double vol = b0.dx[0] * b0.dx[1] * b0.dx[2];
  const double upd = update_coefficient(sp, vol);
  const int s = u0.ks + segid * kseg;
  if (s > u0.ke) return;
  const int e = (s + kseg - 1 < u0.ke) ? s + kseg - 1 : u0.ke;

  extern __shared__ __attribute__((aligned(16))) double stash[];
  // [cell][f3 | du][var][lane]
  double *st_f3[2] = {stash + lane, stash + 2 * NV * 64 + lane};
  double *st_du[2] = {stash + NV * 64 + lane, stash + 3 * NV * 64 + lane};
  constexpr bool CF = (FLUID == APK_FLUID_GLMMHD) && (RS == APK_RS_HLLD);
  auto cf_of = [&](const double (&w)[NV]) -> double {
    if constexpr (CF) return fast_speed(sp.gamma, w[IDN], w[IPR], w[IB1], w[IB2], w[IB3]);
    else return 0.0;
  };
  auto solve = [&](const double (&wl)[NV], const double (&wr)[NV], double cfl, double cfr, double (&f)[NV]) {
    if constexpr (CF) glmmhd_hlld_cf(wl, wr, sp.k, cfl, cfr, f);
    else riemann<FLUID, RS>(wl, wr, sp.k, f);
  };
  double wprev[2][NV], cf3_prev[2] = {0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    load_input_state<FLUID, FROM_CONS, X1H>(((s - 1 < u0.ks) ? prim_klo : prim) + (int64_t)plane(s - 1) * lsk + r * lsj, lsn, sp, wprev[r]);
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      st_f3[r][n * 64] = 0.0;
      st_du[r][n * 64] = 0.0;
    }
    if constexpr (CF) {
      double wp3[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) wp3[q] = wprev[r][perm<3>(q)];
      cf3_prev[r] = cf_of(wp3);
    }
  }
  double raw[2][NV];
  auto load_raw = [&](int c) {
    const double *pc = ((c > u0.ke) ? prim_khi : prim) + (int64_t)plane(c) * lsk;  // (the choice of base is wave-uniform)
    if constexpr (X1H) {
      load_chain<NV>(pc, lsn, raw[0]);
      load_chain<NV>(pc + lsj, lsn, raw[1]);
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int n = 0; n < NV; ++n) raw[r][n] = pc[n * u0.sn + r * u0.sj];
    }
  };
  auto to_input = [&](const double (&u)[NV], double (&w)[NV]) {
    if constexpr (FROM_CONS) {
      double t[NV], di;
#pragma unroll
      for (int n = 0; n < NV; ++n) t[n] = u[n];
      (void)cons_to_prim_core<FLUID, true>(sp.eos, sp.k.eos_gm1, sp.k.vceil_sq, sp.k.pfloor_over_gm1, t, w, di);
    } else {
#pragma unroll
      for (int n = 0; n < NV; ++n) w[n] = u[n];
    }
  };
  // With FROM_CONS the conserved values of the plane just completed ARE the u1 the update reads: they stay in registers
  // (both cells: 242 VGPRs, no scratch, in the forms with ConsToPrim; the form without it has no room) instead of being
  // read a second time, 72 of the 259 B per cell the march moved -- 1.13 -> 1.05 - 1.07 ms.  (A register prefetch of the next
  // plane on top of it -- 256 VGPRs + 20 B of scratch -- measured 1.15 -> 1.17 ms in round 4 and is gone.)
  constexpr int HOLD = (FROM_CONS && EXTRA != EXTRA_NONE) ? 2 : 0;
  double held[HOLD > 0 ? HOLD : 1][NV];
  for (int c = s; c <= e + 1; ++c) {
    const int64_t off = (int64_t)plane(c) * lsk;
    double wc[2][NV];
    load_raw(c);
    to_input(raw[0], wc[0]);
    to_input(raw[1], wc[1]);
    // ---- x3 faces c of both cells; plane c-1 is complete
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      double wl3[NV], wr3[NV], f3[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        wl3[q] = wprev[r][perm<3>(q)];
        wr3[q] = wc[r][perm<3>(q)];
      }
      const double cf3_c = cf_of(wr3);
      solve(wl3, wr3, cf3_prev[r], cf3_c, f3);
      cf3_prev[r] = cf3_c;
      if (c >= s + 1) {
        const int64_t done = col + (int64_t)(c - 1) * u0.sk + r * u0.sj;
        double du[NV], u1v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<3>(q);
          du[n] = st_du[r][n * 64] + (area3 * f3[q] - area3 * st_f3[r][q * 64]);
        }
        if (r < HOLD) {
#pragma unroll
          for (int n = 0; n < NV; ++n) u1v[n] = held[r < HOLD ? r : 0][n];
        } else {
#pragma unroll
          for (int n = 0; n < NV; ++n) u1v[n] = c1[n * u0.sn + done];
        }
        const bool store = sp.cons_store == 0 || (sp.cons_store == 1 && (shell_ij[r] || c - 1 < u0.ks + u0.ng || c - 1 > u0.ke - u0.ng));
        if constexpr (X1H) {
          if (active) {
            X1Store xr = xs;
            if (r == 1 && xr.seg) xr.seg += sp.x1_send_depth;
            finish_cell<FLUID, EXTRA, LEAN>(u0, b0, u1v, done, du, vol, sp, lane_min_dt, prim_dst, upd, store, &xr, c - 1 - u0.ks);
          }
        } else {
          if (active) finish_cell<FLUID, EXTRA, LEAN>(u0, b0, u1v, done, du, vol, sp, lane_min_dt, prim_dst, upd, store);
        }
      }
      if (r < HOLD) {  // (plane c of this cell, for the next iteration)
#pragma unroll
        for (int n = 0; n < NV; ++n) held[r < HOLD ? r : 0][n] = raw[r][n];
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) st_f3[r][q * 64] = f3[q];
#pragma unroll
      for (int n = 0; n < NV; ++n) wprev[r][n] = wc[r][n];
    }
    if (c <= e) {  // wave-uniform
      // ---- x1 faces of both rows of plane c
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        double wl[NV], wr[NV], f[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wr[q] = wc[r][perm<1>(q)];
          wl[q] = wave_shr1(wr[q]);
        }
        const double cf1_c = cf_of(wr);
        solve(wl, wr, CF ? wave_shr1(cf1_c) : 0.0, cf1_c, f);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const double fup = wave_shl1(f[q]);
          st_du[r][perm<1>(q) * 64] = (area1 * fup - area1 * f[q]);
        }
      }
      // ---- x2 faces: below A, between A and B (ONE solve, both cells use it), above B
      double wa[NV], wb[NV], fmid[NV];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        wa[q] = wc[0][perm<2>(q)];
        wb[q] = wc[1][perm<2>(q)];
      }
      const double cfa = cf_of(wa), cfb = cf_of(wb);
      solve(wa, wb, cfa, cfb, fmid);
      {
        double wm[NV], flo[NV], wnat[NV];
        load_input_state<FLUID, FROM_CONS, X1H>(prim_jm + off, lsn, sp, wnat);
#pragma unroll
        for (int q = 0; q < NV; ++q) wm[q] = wnat[perm<2>(q)];
        solve(wm, wa, cf_of(wm), cfa, flo);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<2>(q);
          st_du[0][n * 64] = st_du[0][n * 64] + (area2 * fmid[q] - area2 * flo[q]);
        }
      }
      {
        double wp[NV], fhi[NV], wnat[NV];
        load_input_state<FLUID, FROM_CONS, X1H>(prim_jp + off, lsn, sp, wnat);
#pragma unroll
        for (int q = 0; q < NV; ++q) wp[q] = wnat[perm<2>(q)];
        solve(wb, wp, cfb, cf_of(wp), fhi);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int n = perm<2>(q);
          st_du[1][n * 64] = st_du[1][n * 64] + (area2 * fhi[q] - area2 * fmid[q]);
        }
      }
    }
  }
  if constexpr (EXTRA == EXTRA_C2P_DT) {
    double m = lane_min_dt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if (lane == 0) atomicMin(sp.dt_bits, (unsigned long long)__double_as_longlong(m));
  }
}

// ==============================================================================================
// Passive scalars of the fused path.  The sweeps above leave the mass flux through every face in
// sp.mflux; a scalar's flux is that mass flux times the upwind reconstructed concentration
// (src/hydro/hydro.cpp:1088-1097,1134-1143,1182-1191), so its whole update is one light kernel:
// per cell and scalar the six face values, the flux difference in the reference's order and the
// RK update.  Reads old prim (stencil), writes cons only; the new concentrations (prim) are
// written by a second kernel once every cell has been updated.
// ==============================================================================================
template <int RECON>
__global__ void __launch_bounds__(256)
fused_scalar_update_kernel(PackView u0, PackView u1, StageParams sp) {
  const int i = u0.is + blockIdx.x * 64 + threadIdx.x;
  const int j = u0.js + blockIdx.y * 4 + threadIdx.y;
  const int b = blockIdx.z / u0.nx3;
  const int k = u0.ks + blockIdx.z % u0.nx3;
  if (i > u0.ie || j > u0.je) return;
  const apk_block_desc b0 = u0.blocks[b];
  const double *c1 = u1.blocks[b].cons;
  const int64_t cell = k * u0.sk + (int64_t)j * u0.sj + i;
  const double area[3] = {b0.dx[1] * b0.dx[2], b0.dx[0] * b0.dx[2], b0.dx[0] * b0.dx[1]};
  const double vol = b0.dx[0] * b0.dx[1] * b0.dx[2];
  const int64_t st[3] = {1, u0.sj, u0.sk};
  for (int n = u0.nhydro; n < u0.nvar; ++n) {
    const double *p = b0.prim + n * u0.sn + cell;
    double du = 0.0;
    for (int d = 0; d < u0.ndim; ++d) {
      const double *m = sp.mflux + ((int64_t)d * u0.nblocks + b) * u0.sn + cell;
      double sl, sr;
      face_states_any<RECON>(p, st[d], b0.dx[d], n, sl, sr);
      const double flo = (m[0] >= 0.0) ? m[0] * sl : m[0] * sr;
      face_states_any<RECON>(p + st[d], st[d], b0.dx[d], n, sl, sr);
      const double fhi = (m[st[d]] >= 0.0) ? m[st[d]] * sl : m[st[d]] * sr;
      if (d == 0) du = (area[0] * fhi - area[0] * flo);
      else du += (area[d] * fhi - area[d] * flo);
    }
    const int64_t idx = n * u0.sn + cell;
    const double old = (sp.gam0 != 0.0) ? as_global(b0.cons)[idx] : 0.0;
    b0.cons[idx] = sp.gam0 * old + sp.gam1 * c1[idx] + sp.beta_dt * (-du / vol);
  }
}

// prim(scalar) = cons(scalar) / rho of the updated interior (adiabatic_hydro.hpp:139-141)
// (a template only so that the header can be included from several translation units)
template <int RECON>
__global__ void __launch_bounds__(256)
fused_scalar_prim_kernel(PackView u0, PackView u1, int to_u1) {
  const int i = u0.is + blockIdx.x * 64 + threadIdx.x;
  const int j = u0.js + blockIdx.y * 4 + threadIdx.y;
  const int b = blockIdx.z / u0.nx3;
  const int k = u0.ks + blockIdx.z % u0.nx3;
  if (i > u0.ie || j > u0.je) return;
  const apk_block_desc b0 = u0.blocks[b];
  double *prim = to_u1 ? u1.blocks[b].prim : b0.prim;
  const int64_t cell = k * u0.sk + (int64_t)j * u0.sj + i;
  const double di = 1.0 / b0.cons[IDN * u0.sn + cell];
  for (int n = u0.nhydro; n < u0.nvar; ++n) prim[n * u0.sn + cell] = b0.cons[n * u0.sn + cell] * di;
}

template <int RECON>
inline void launch_scalar_update(const PackView &u0, const PackView &u1, const StageParams &sp, int extra, hipStream_t s) {
  const dim3 grid((u0.nx1 + 63) / 64, (u0.nx2 + 3) / 4, u0.nx3 * u0.nblocks), block(64, 4, 1);
  hipLaunchKernelGGL((fused_scalar_update_kernel<RECON>), grid, block, 0, s, u0, u1, sp);
  if (extra != EXTRA_NONE) hipLaunchKernelGGL((fused_scalar_prim_kernel<RECON>), grid, block, 0, s, u0, u1, sp.prim_to_u1);
}

}  // namespace apk
#include "fused2_kernel.hpp"
#include "fused3_kernel.hpp"
namespace apk {

// ---- launch helpers ---------------------------------------------------------------------------
// segments per march so that the launch has at least ~2 x 2048 waves (MI355X: 1024 SIMDs x 2
// resident march waves), never shorter than 16 cells
// transverse rows per wave of a march: fill the 64 lanes when the block is narrower than a wave
inline int march_rows_per_wave(int nx1, int ntrans) {
  int rpw = (nx1 <= 16) ? 4 : ((nx1 <= 32) ? 2 : 1);
  while (rpw > 1 && rpw > ntrans) rpw /= 2;
  return rpw;
}

inline int march_segments(int64_t waves, int n_along) {
  int nseg = (int)((4096 + waves - 1) / waves);
  // (segments no shorter than 8 rows: a pack of 232 16^3 blocks is 928 march waves on 2048 slots in one piece,
  // 1856 in two -- 1.14 -> 1.09 ms per cycle of the refined MHD blast; 4-row segments lose it again)
  constexpr int min_rows = 8;
  const int max_seg = n_along / min_rows > 0 ? n_along / min_rows : 1;
  if (nseg > max_seg) nseg = max_seg;
  return nseg < 1 ? 1 : nseg;
}

template <int FLUID, int RECON, int RS, int DIR>
inline void launch_final_march(const PackView &u0, const PackView &u1, const StageParams &sp, int extra,
                               dim3 grid, int lds, hipStream_t s) {
  // (a finishing march that replaces prim in place must own its columns from end to end: the
  // next segment's stencil rows would be overwritten under it)
  const bool in_place = (extra != EXTRA_NONE) && !sp.prim_to_u1;
  const int ntrans = (DIR == 2) ? u0.nx3 : u0.nx2;
  const int rpw = march_rows_per_wave(u0.nx1, ntrans);
  grid.x = (u0.nx1 + 64 / rpw - 1) / (64 / rpw);
  grid.y = (ntrans + rpw - 1) / rpw;
  const int nseg = in_place ? 1 : march_segments((int64_t)grid.x * grid.y * grid.z, DIR == 2 ? u0.nx2 : u0.nx3);
  grid.z *= nseg;
  if (extra == EXTRA_C2P_DT)
    hipLaunchKernelGGL((fused_march_kernel<FLUID, RECON, RS, DIR, true, EXTRA_C2P_DT>), grid, dim3(64), lds, s, u0, u1, sp, nseg, rpw);
  else if (extra == EXTRA_C2P)
    hipLaunchKernelGGL((fused_march_kernel<FLUID, RECON, RS, DIR, true, EXTRA_C2P>), grid, dim3(64), lds, s, u0, u1, sp, nseg, rpw);
  else
    hipLaunchKernelGGL((fused_march_kernel<FLUID, RECON, RS, DIR, true, EXTRA_NONE>), grid, dim3(64), lds, s, u0, u1, sp, nseg, rpw);
}

template <int FLUID, int RECON, int RS>
inline int launch_fused_stage(const PackView &u0, const PackView &u1, const StageParams &sp,
                              int extra, hipStream_t s) {
  // a windowed x1 sweep (phase 1) flattens at most window_rl columns per row
  const int64_t run1 = (int64_t)u0.nx2 * (sp.window ? sp.window_rl : u0.nx1 + x1_first_lane(RECON) + 1);
  constexpr int cpw1 = x1_cells_per_wave(RECON);
  const int wpp = (int)((run1 + cpw1 - 1) / cpw1);
  const dim3 g1((wpp + 3) / 4, u0.nx3 * u0.nblocks, 1);
  if (sp.phase != 0) {
    // split stage: where the x1 sweep is its own, non-finishing kernel, or the single-kernel
    // 3-D donor-cell stage with out-of-place (or no) FillDerived
    if (u0.ndim == 1) return APK_ERR_UNSUPPORTED;
    if (RECON == APK_RC_DC && u0.ndim == 3) {
      if (!(extra == EXTRA_NONE || (sp.prim_to_u1 && extra == EXTRA_C2P))) return APK_ERR_UNSUPPORTED;
      if (sp.phase == 2) {  // the cells were all retired in phase 1; only the scalars are left
        if (sp.mflux) launch_scalar_update<RECON>(u0, u1, sp, extra, s);
        return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
      }
    }
  }
  if (sp.face_nbr) {
    // only the kernels that follow the table: the single-march donor-cell stage and the two-kernel
    // stage; nothing that reads neighbouring cells from memory by plain index arithmetic
    const bool form_ok = (RECON == APK_RC_DC) ? (extra == EXTRA_NONE || sp.prim_to_u1 != 0)
                                              : two_kernel_stage_applies(u0, RECON, extra, sp);
    if (u0.ndim != 3 || sp.mflux || sp.dedner == 2 || !form_ok) return APK_ERR_UNSUPPORTED;
  }
  // apk_stage_args.x1_halo: the lean two-row donor-cell march and the lean two-kernel stage's finishing march (phase 1 of
  // a split two-kernel stage is the x3 sweep, which reads no x1 ghost column and retires nothing: it ignores the table)
  if (sp.x1_blocks && !(RECON != APK_RC_DC && sp.phase == 1) &&
      (!x1_halo_stage_ok(u0, RECON, extra, sp) || (RECON != APK_RC_DC && single_march_stage_applies<FLUID, RECON>(u0, extra, sp))))
    return APK_ERR_UNSUPPORTED;
  if (sp.no_prim_store && !(u0.ndim == 3 && RECON != APK_RC_DC && stage_is_lean(sp) && two_kernel_stage_applies(u0, RECON, extra, sp)))
    return APK_ERR_UNSUPPORTED;
  if (sp.prim_from_cons) {
    const bool dc_ok = RECON == APK_RC_DC && sp.prim_from_cons == 1 && (extra == EXTRA_NONE || sp.prim_to_u1);
    // the two-kernel stage, whole or split (a stage that reads u0's conserved state writes its result elsewhere)
    const bool two_ok = RECON != APK_RC_DC && two_kernel_stage_applies(u0, RECON, extra, sp) && !sp.mflux &&
                        (sp.prim_from_cons == 1 || sp.out_delta != 0);
    if (!(u0.ndim == 3 && stage_is_lean(sp) && (dc_ok || two_ok))) return APK_ERR_UNSUPPORTED;
  }
  const bool do_x1 = sp.phase != 2, do_rest = sp.phase != 1;
  // timing slots: donor-cell stages (VL2 predictor) are accounted separately
  constexpr int TS = (RECON == APK_RC_DC) ? (int)APK_T_FUSED_DC_X1 : (int)APK_T_FUSED_X1;
  if (u0.ndim == 1) {
    ScopedTiming t(sp.ctx, TS + 0, s);
    hipLaunchKernelGGL((fused_x1_kernel<FLUID, RECON, RS, true>), g1, dim3(256), 0, s, u0, u1, sp, wpp);
  } else if (u0.ndim == 3) {
    constexpr int lds = march_lds_bytes<FLUID, RECON>();
    if constexpr (RECON == APK_RC_DC) {
      if (extra == EXTRA_NONE || sp.prim_to_u1) {
        // whole donor-cell stage in one march (see fused_dc3_kernel); its FillDerived is out of place
        const int64_t run3 = sp.window ? (int64_t)sp.window_rows * sp.window_rl : (int64_t)u0.nx2 * (u0.nx1 + 2);
        const int wpb = (int)((run3 + 61) / 62);
        const int lean_level = stage_lean_level(sp);
        const bool lean = lean_level == 1;
        // (the lean form with a pressure floor / the trial count: the two-row march with FillDerived from stored primitives)
        const bool lean2 = lean_level == LEAN_PFLOOR && extra != EXTRA_NONE && !sp.prim_from_cons && !sp.x1_blocks;
        // two rows per lane (fused_dc3r2_kernel): whole blocks only -- a split stage's windows keep the one-row kernel
        const bool two_rows = (lean || lean2) && !sp.window && u0.nx2 % 2 == 0 && u0.nx2 >= 4;
        const int wpb2 = (int)(((int64_t)(u0.nx2 / 2) * (u0.nx1 + 2) + 61) / 62);
        const int wpb_run = two_rows ? wpb2 : wpb;  // wave columns per block of the kernel that will run
        // measured on 8 x 128^3 (round 5, two-row march from the conserved state, ms per cycle of the headline, same box):
        // 6 planes 3.67 - 3.70, 8: 3.64 - 3.66, 12: 3.65, 16: 3.61 - 3.63, 32: 3.61 - 3.66 -- the predictor itself is
        // fastest at 8 (1.07 against 1.08 / 1.16 ms at 16 / 32), but what it leaves of the power budget is clock for the
        // two kernels after it (finishing march 1.85 -> 1.82 -> 1.76 ms): 16 is the cycle's optimum
        int kseg = (u0.nx3 >= 32) ? 16 : ((u0.nx3 >= 16) ? 8 : u0.nx3);
        if (u0.nx3 >= 16 && (int64_t)wpb_run * ((u0.nx3 + 7) / 8) * u0.nblocks < 4 * 2048) {
          // small packs (refined meshes of 16^3 blocks): a few waves per SIMD in all, so pick the segment length with
          // the fewest plane-steps on the busiest SIMD -- a segment costs its planes plus about 1.5 for the prologue, a
          // SIMD (1024 of them) works through ceil(waves / 1024) segments at the rate two resident waves share, and a
          // wave that has its SIMD to itself runs 1.4 times as fast as one of a pair, not twice.  (Round 5: counted with
          // the wave columns of the kernel that RUNS -- the two-row march has 3 per 16^3 block where the one-row march has
          // 5 -- and per SIMD instead of per round of 2048 waves.  232 blocks of 16^3, two-row march, us per launch:
          // segments of 4 planes 124, 8: 134, 6: 141, 16: 167 -- the order this estimate gives.)
          double best = 1.0e300;
          for (const int cand : {4, 6, 8, 16}) {
            if (cand > u0.nx3) continue;
            const int64_t waves = (int64_t)wpb_run * ((u0.nx3 + cand - 1) / cand) * u0.nblocks;
            const double per_simd = waves <= 1024 ? 2.0 / 1.4 : (double)((waves + 1023) / 1024);
            const double cost = per_simd * (cand + 1.5);
            if (cost < best) best = cost, kseg = cand;
          }
        }
        const int nseg = (u0.nx3 + kseg - 1) / kseg;
        const int64_t total = (int64_t)wpb * nseg * u0.nblocks;
        const int per_xcd = (int)((total + 7) / 8);
        const dim3 g((unsigned)(per_xcd * 8), 1, 1);
        constexpr int lds3 = 2 * nvars<FLUID>() * 64 * (int)sizeof(double);
        ScopedTiming t(sp.ctx, TS + 0, s);
        if (two_rows) {
          const int64_t total2 = (int64_t)wpb2 * nseg * u0.nblocks;
          const int per_xcd2 = (int)((total2 + 7) / 8);
          const dim3 g2((unsigned)(per_xcd2 * 8), 1, 1);
          constexpr int lds4 = 4 * nvars<FLUID>() * 64 * (int)sizeof(double);
#define APK_LAUNCH_DC3R2(EXTRA_, FC_) \
  hipLaunchKernelGGL((fused_dc3r2_kernel<FLUID, RS, EXTRA_, FC_>), g2, dim3(64), lds4, s, u0, u1, sp, kseg, wpb2, nseg, per_xcd2)
#define APK_LAUNCH_DC3R2_X1H(EXTRA_, FC_) \
  hipLaunchKernelGGL((fused_dc3r2_kernel<FLUID, RS, EXTRA_, FC_, true>), g2, dim3(64), lds4, s, u0, u1, sp, kseg, wpb2, nseg, per_xcd2)
          if (lean2) {
            if (extra == EXTRA_C2P_DT) hipLaunchKernelGGL((fused_dc3r2_kernel<FLUID, RS, EXTRA_C2P_DT, false, false, LEAN_PFLOOR>), g2, dim3(64), lds4, s, u0, u1, sp, kseg, wpb2, nseg, per_xcd2);
            else hipLaunchKernelGGL((fused_dc3r2_kernel<FLUID, RS, EXTRA_C2P, false, false, LEAN_PFLOOR>), g2, dim3(64), lds4, s, u0, u1, sp, kseg, wpb2, nseg, per_xcd2);
          } else if (sp.x1_blocks) {  // (apk_stage_args.x1_halo: stages without the dt estimate -- the predictor's place in a cycle)
            if (extra == EXTRA_C2P_DT) return APK_ERR_UNSUPPORTED;
            if (extra == EXTRA_C2P) {
              if (sp.prim_from_cons) APK_LAUNCH_DC3R2_X1H(EXTRA_C2P, true);
              else APK_LAUNCH_DC3R2_X1H(EXTRA_C2P, false);
            } else {
              if (sp.prim_from_cons) APK_LAUNCH_DC3R2_X1H(EXTRA_NONE, true);
              else APK_LAUNCH_DC3R2_X1H(EXTRA_NONE, false);
            }
          } else if (extra == EXTRA_C2P_DT) {
            if (sp.prim_from_cons) APK_LAUNCH_DC3R2(EXTRA_C2P_DT, true);
            else APK_LAUNCH_DC3R2(EXTRA_C2P_DT, false);
          } else if (extra == EXTRA_C2P) {
            if (sp.prim_from_cons) APK_LAUNCH_DC3R2(EXTRA_C2P, true);
            else APK_LAUNCH_DC3R2(EXTRA_C2P, false);
          } else {
            if (sp.prim_from_cons) APK_LAUNCH_DC3R2(EXTRA_NONE, true);
            else APK_LAUNCH_DC3R2(EXTRA_NONE, false);
          }
#undef APK_LAUNCH_DC3R2
#undef APK_LAUNCH_DC3R2_X1H
          return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
        }
        if (sp.x1_blocks) return APK_ERR_UNSUPPORTED;  // (x1_halo: the two-row march only)
        if (sp.prim_from_cons) {  // (one row per lane: odd row counts, the windows of a split stage)
#define APK_LAUNCH_DC3FC(EXTRA_) \
  hipLaunchKernelGGL((fused_dc3_kernel<FLUID, RS, EXTRA_, 1, true>), g, dim3(64), lds3, s, u0, u1, sp, kseg, wpb, nseg, per_xcd)
          if (extra == EXTRA_C2P_DT) APK_LAUNCH_DC3FC(EXTRA_C2P_DT);
          else if (extra == EXTRA_C2P) APK_LAUNCH_DC3FC(EXTRA_C2P);
          else APK_LAUNCH_DC3FC(EXTRA_NONE);
#undef APK_LAUNCH_DC3FC
          return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
        }
#define APK_LAUNCH_DC3(EXTRA_, LEAN_) \
  hipLaunchKernelGGL((fused_dc3_kernel<FLUID, RS, EXTRA_, LEAN_>), g, dim3(64), lds3, s, u0, u1, sp, kseg, wpb, nseg, per_xcd)
        // (the windows of a split stage and odd row counts take the one-row march in the SAME lean form the whole stage
        // takes as the two-row march: the product build's update expression is the form's, and a split stage must
        // reproduce the whole one bit for bit)
        if (extra == EXTRA_C2P_DT) {
          if (lean) APK_LAUNCH_DC3(EXTRA_C2P_DT, 1);
          else if (lean2) APK_LAUNCH_DC3(EXTRA_C2P_DT, LEAN_PFLOOR);
          else APK_LAUNCH_DC3(EXTRA_C2P_DT, 0);
        } else if (extra == EXTRA_C2P) {
          if (lean) APK_LAUNCH_DC3(EXTRA_C2P, 1);
          else if (lean2) APK_LAUNCH_DC3(EXTRA_C2P, LEAN_PFLOOR);
          else APK_LAUNCH_DC3(EXTRA_C2P, 0);
        } else {
          if (lean) APK_LAUNCH_DC3(EXTRA_NONE, 1);
          else APK_LAUNCH_DC3(EXTRA_NONE, 0);
        }
#undef APK_LAUNCH_DC3
        if (sp.mflux && sp.phase == 0) launch_scalar_update<RECON>(u0, u1, sp, extra, s);
        return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
      }
      // (with FillDerived fused into the stage) donor cell: the sweeps are HBM-bound, so x1 and x2 share ONE march over (k,i)-flattened
      // lanes (one du round trip and one prim read less).  With a high-order reconstruction
      // the sweeps are ALU-bound and the 62/64 x 128/134 lane efficiency of the flattened march
      // (and its 2216-wave grid on a 2048-wave machine) costs more than the traffic it saves
      // (measured: PPM+HLLD 3.3 ms fused vs 1.39 + 1.02 ms separate on 8 x 128^3).
      const int64_t run12 = (int64_t)u0.nx3 * u0.ni;
      const int wpb = (int)((run12 + 61) / 62);
      ScopedTiming t(sp.ctx, TS + 0, s);
      hipLaunchKernelGGL((fused_march12_kernel<FLUID, RECON, RS>), dim3(wpb, 1, u0.nblocks), dim3(64), lds, s,
                         u0, u1, sp, wpb);
    } else if (two_kernel_stage_applies(u0, RECON, extra, sp) && single_march_stage_applies<FLUID, RECON>(u0, extra, sp)) {
      // the whole stage in one march (fused3_kernel.hpp: three-point reconstructions, input from a conserved state)
      ScopedTiming t(sp.ctx, TS + 0, s);
      launch_s3<FLUID, RECON, RS>(u0, u1, sp, extra, s);
      return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
    } else if (two_kernel_stage_applies(u0, RECON, extra, sp)) {
      // two-kernel stage (fused2_kernel.hpp): the x3 sweep writes its flux difference, then one
      // march does x1 + x2 and finishes.  A split stage runs the x3 sweep on plane windows in
      // phase 1 (it reads x3 ghost zones only) and the finishing march in phase 2.
      const int du_pitch = ((u0.nx1 + 15) / 16) * 16;
      if (do_x1) {
        StageParams sp1 = sp;
        sp1.du_first = 1;
        sp1.du_pitch = du_pitch;
        const int rpw = march_rows_per_wave(u0.nx1, u0.nx2);
        dim3 g3((u0.nx1 + 64 / rpw - 1) / (64 / rpw), (u0.nx2 + rpw - 1) / rpw, u0.nblocks);
        const int nseg = march_segments((int64_t)g3.x * g3.y * g3.z, u0.nx3);
        g3.z *= nseg;
        ScopedTiming t(sp.ctx, TS + 2, s);
        if (sp.prim_from_cons)
          hipLaunchKernelGGL((fused_march_kernel<FLUID, RECON, RS, 3, false, EXTRA_NONE, true>), g3, dim3(64), lds, s, u0, u1, sp1, nseg, rpw);
        else
          hipLaunchKernelGGL((fused_march_kernel<FLUID, RECON, RS, 3, false>), g3, dim3(64), lds, s, u0, u1, sp1, nseg, rpw);
      }
      if (do_rest) {
        StageParams sp2 = sp;
        sp2.window = nullptr;
        sp2.du_pitch = du_pitch;
        ScopedTiming t(sp.ctx, TS + 0, s);
        launch_m12f<FLUID, RECON, RS>(u0, u1, sp2, extra, s);
      }
      if (sp.mflux && do_rest) launch_scalar_update<RECON>(u0, u1, sp, extra, s);
      return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
    } else {
      if (do_x1) {
        ScopedTiming t(sp.ctx, TS + 0, s);
        hipLaunchKernelGGL((fused_x1_kernel<FLUID, RECON, RS, false>), g1, dim3(256), 0, s, u0, u1, sp, wpp);
      }
      if (do_rest) {
        const int rpw = march_rows_per_wave(u0.nx1, u0.nx3);
        dim3 g2((u0.nx1 + 64 / rpw - 1) / (64 / rpw), (u0.nx3 + rpw - 1) / rpw, u0.nblocks);
        const int nseg = march_segments((int64_t)g2.x * g2.y * g2.z, u0.nx2);
        g2.z *= nseg;
        ScopedTiming t(sp.ctx, TS + 1, s);
        hipLaunchKernelGGL((fused_march_kernel<FLUID, RECON, RS, 2, false>), g2, dim3(64), lds, s, u0, u1, sp, nseg, rpw);
      }
    }
    if (do_rest) {
      const dim3 g3((u0.nx1 + 63) / 64, u0.nx2, u0.nblocks);
      ScopedTiming t(sp.ctx, TS + 2, s);
      launch_final_march<FLUID, RECON, RS, 3>(u0, u1, sp, extra, g3, lds, s);
    }
  } else {
    if (do_x1) {
      ScopedTiming t(sp.ctx, TS + 0, s);
      hipLaunchKernelGGL((fused_x1_kernel<FLUID, RECON, RS, false>), g1, dim3(256), 0, s, u0, u1, sp, wpp);
    }
    // 2-D: the (j,i)-flattened x1 sweep keeps far more waves in flight than a march over a
    // single k-plane would; the x2 march finishes the stage
    if (do_rest) {
      constexpr int lds = march_lds_bytes<FLUID, RECON>();
      const dim3 g2((u0.nx1 + 63) / 64, u0.nx3, u0.nblocks);
      ScopedTiming t(sp.ctx, TS + 1, s);
      launch_final_march<FLUID, RECON, RS, 2>(u0, u1, sp, extra, g2, lds, s);
    }
  }
  if (sp.mflux && do_rest) launch_scalar_update<RECON>(u0, u1, sp, extra, s);
  return hipGetLastError() == hipSuccess ? APK_OK : APK_ERR_DEVICE;
}

template <int FLUID, int RS>
inline int launch_fused_family(const PackView &u0, const PackView &u1, int recon,
                               const StageParams &sp, int extra, hipStream_t s) {
  switch (recon) {
  case APK_RC_DC: return launch_fused_stage<FLUID, APK_RC_DC, RS>(u0, u1, sp, extra, s);
  case APK_RC_PLM: return launch_fused_stage<FLUID, APK_RC_PLM, RS>(u0, u1, sp, extra, s);
  case APK_RC_PPM: return launch_fused_stage<FLUID, APK_RC_PPM, RS>(u0, u1, sp, extra, s);
  case APK_RC_WENOZ: return launch_fused_stage<FLUID, APK_RC_WENOZ, RS>(u0, u1, sp, extra, s);
  case APK_RC_WENO3: return launch_fused_stage<FLUID, APK_RC_WENO3, RS>(u0, u1, sp, extra, s);
  case APK_RC_LIMO3: return launch_fused_stage<FLUID, APK_RC_LIMO3, RS>(u0, u1, sp, extra, s);
  default: return APK_ERR_UNSUPPORTED;
  }
}

int launch_fused_euler_hlle(const PackView &u0, const PackView &u1, int recon, const StageParams &sp, int extra, hipStream_t s);
int launch_fused_euler_hllc(const PackView &u0, const PackView &u1, int recon, const StageParams &sp, int extra, hipStream_t s);
int launch_fused_mhd_hlle(const PackView &u0, const PackView &u1, int recon, const StageParams &sp, int extra, hipStream_t s);
int launch_fused_mhd_hlld(const PackView &u0, const PackView &u1, int recon, const StageParams &sp, int extra, hipStream_t s);

}  // namespace apk
