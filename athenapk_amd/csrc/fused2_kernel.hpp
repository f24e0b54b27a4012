// fused2_kernel.hpp -- the two-kernel form of the fused 3-D stage (no flux-difference round trips).
//
// The three-sweep schedule of fused_kernel.hpp moves `du` through HBM one and a half times and reads
// `prim` three times: 13.2 GB per PPM+HLLD stage on 8 x 128^3 against 3.6 GB of algorithmic traffic
// (profiles/r01_hbm_traffic.json), and after the instruction diet of round 2 its x2 and x3 marches
// run at the HBM rate, not the VALU rate.  Here a stage is
//
//   K1  fused_march_kernel<.., DIR = 3, FINAL = false> with StageParams.du_first: the x3 sweep alone,
//       writing its flux difference d3 = A3 (F3(k+1) - F3(k)) once            (prim in, d3 out)
//   K2  fused_m12f_kernel (this file): x1 AND x2 in one march along j that also finishes the stage,
//       du = (d1 + d2) + d3 -- the reference's accumulation order x1, x2, x3 (hydro.cpp:1070-1199;
//       floating-point addition is evaluated in that association whichever kernel ran first) --
//       then RK update, Dedner source, ConsToPrim (out of place) and the dt reduction
//                                                 (prim, d3, u1 [, u0] in; u0, prim' out)
//
// = 8.6 GB instead of 13.2 GB, and each kernel's VALU time now about equals its HBM time.
//
// K2's lanes are the (k, i)-flattened cells of one j-row of a block, 64 consecutive cells of the
// flattened run per wave (x1_cells_per_wave(RECON) of them retire, as in the x1 sweep: states and
// fluxes travel between neighbouring lanes by DPP wave shifts); the wave marches along j with the
// x2 stencil rows in its private LDS ring exactly as fused_march_kernel does.  In iteration c it
// solves the x2 face between rows c-1 and c and the x1 faces of row c-1 and retires row c-1.
//
// Scheduling: the (block, chunk) columns x nx2 rows are one global list of wave-rows, cut into
// EQUAL contiguous ranges, one per launched wave, and exactly as many waves are launched as the GPU
// holds at once (2 per SIMD).  Every wave therefore does the same amount of work in one round (a
// grid of 2256 full-length marches on 2048 slots would run in two rounds), at the price of one
// extra stencil prologue per range boundary.  Consecutive ranges go to the same XCD (the hardware
// deals workgroup ids round-robin over the 8 XCDs), so the rows a wave re-reads for the x1 stencil
// and its neighbours' overlap columns are in that XCD's L2.
// (included by fused_kernel.hpp after its kernels, before its launch helpers)
#pragma once

namespace apk {

// Neighbour-lane access of K2: value of lane (l - K) for K = 1, 2 (shr) / lane (l + K) (shl) by DPP wave shifts; lanes
// without such a neighbour receive 0 (they never retire a cell).  (Through the LDS crossbar instead -- ds_bpermute_b32, no
// VALU issue slot -- measured 3.26 ms per stage against 3.11 in round 2.)
template <int K>
APK_DEV double lane_below(double x, int) {
  double r = wave_shr1(x);
  if constexpr (K == 2) r = wave_shr1(r);
  return r;
}
template <int K>
APK_DEV double lane_above(double x, int) {
  double r = wave_shl1(x);
  if constexpr (K == 2) r = wave_shl1(r);
  return r;
}

// Lanes of a K2 wave that retire a cell.  A cell needs the fluxes of both its x1 faces; a face needs
// the reconstructed states of the two cells it separates; a state needs the cell's stencil (H lanes
// either side, fetched by wave shifts) and, with PPM, the interface value of the lane below:
//   stencil half width 2 (PPM, WENO-Z): states valid in lanes 2..61, faces 3..61, cells 3..60
//   stencil half width 1:               states valid in lanes 1..62, faces 2..62, cells 2..61
constexpr int m12_first_lane(int recon) { return recon_halfwidth(recon) + 1; }
constexpr int m12_last_lane(int recon) { return 62 - recon_halfwidth(recon); }


#ifndef APK_M12F_TIMING
// 1 (diagnostic variant build, `make variant VAR=tm VARFLAGS=-DAPK_M12F_TIMING=1`): per-phase shader-clock sums of the
// march, printed by launch_m12f.  Round 3, general PPM+HLLD stage with FillDerived + dt on 8 x 128^3: x1 reconstruction
// 28 %, x1 Riemann 8.4 %, x2 reconstruction 21.5 %, x2 Riemann 9.8 %, wait for d3 / u1 9.5 %, finish (incl. the wait for
// the old u0 the explicit s_waitcnt of this build splits off) 20 %, loop control 2.8 % -- the two HLLD solves, a third of
// the instructions, take 18 % of the time; PPM with its divergent extremum branches takes half.
#define APK_M12F_TIMING 0
#endif
#if APK_M12F_TIMING
__device__ unsigned long long g_m12f_phase[8];
#define APK_TICK(slot)                                            \
  do {                                                            \
    const unsigned long long now_ = clock64();                    \
    if (lane == 0) phase_acc[slot] += now_ - tick_;               \
    tick_ = now_;                                                 \
  } while (0)
#elif defined(APK_PHASE_MARKERS)
// (tools/ledger.sh: an assembler comment at every phase boundary, for the per-phase instruction ledger of
// tools/isa_loop_ledger.py; no instruction, but volatile asm statements pin the phase order like the ticks do)
#define APK_TICK(slot) asm volatile(";;APK_PHASE " #slot ::: "memory")
#else
#define APK_TICK(slot) do { } while (0)
#endif

// does the from-cons finishing march keep the loaded rows in a second LDS ring?  (two rings within the 20 KB a wave may
// take with two waves per SIMD, with some room)
template <int FLUID, int RECON>
constexpr bool m12f_keeps_raw_rows() {
  return 2 * (2 * recon_halfwidth(RECON) * nvars<FLUID>() * 64 * (int)sizeof(double)) <= 19 * 1024;
}

// X1H (apk_stage_args.x1_halo; lean forms that read stored primitives): lanes on x1 ghost columns behind a face with a
// receive segment load their rows from that segment -- a per-lane stride between variables and a per-lane row offset,
// where every other lane adds the wave-uniform n * sn + row * st to its pointer -- and lanes that retire a cell within
// `send_depth` of a face with a send segment store it a second time, into the segment.  A template parameter: the
// per-lane stride is a register pair the marches without it do not carry.
template <int FLUID, int RECON, int RS, int EXTRA, int LEAN, bool FC = false, bool X1H = false>
__global__ void __launch_bounds__(64, 2)
fused_m12f_kernel(PackView u0, PackView u1, StageParams sp, int wpb, int nwaves, int per_xcd,
                  long long total_rows) {
  static_assert(RECON != APK_RC_DC, "donor-cell stages have their own single-kernel form");
  static_assert(!FC || LEAN, "prim_from_cons: lean form only");
  static_assert(!X1H || LEAN == 1, "x1_halo: the lean forms");
  static_assert(LEAN != LEAN_PFLOOR || (!FC && EXTRA != EXTRA_NONE), "the lean form with a pressure floor / trial count: stages with FillDerived from stored primitives");
  constexpr int NV = nvars<FLUID>();
  constexpr int H = recon_halfwidth(RECON);
  constexpr int NS = 2 * H;
  constexpr int FIRST = m12_first_lane(RECON), LAST = m12_last_lane(RECON), CPW = LAST - FIRST + 1;
  // Where d3 / u1 of the retiring cell are requested: LOADS = 2 after the x2 solve (measured in round 2 on 8 x 128^3
  // PPM+HLLD: at the top of the iteration 3.44 ms with 148 B of scratch per lane, after the x1 phase 3.10 ms / 156 B, after
  // the x2 solve 2.36 ms / 12 B -- the earlier the loads, the more of the 256 VGPRs they hold through an HLLD solve; one
  // wave per SIMD with 512 VGPRs spills nothing but takes 2.63 ms at best).  The lean march WITHOUT ConsToPrim (the stages
  // of RK2 / RK3 that are not the last, the north-star stage benchmark) has the registers to request u1 after the x1
  // phase without scratch (LOADS = 3; 244 VGPRs): general stage 3.07 -> 2.98 ms, same box.
  // (Round 5, after the row addressing freed 10 - 17 VGPRs: u1 after the x1 phase in every lean form -- 247 VGPRs, no
  // scratch -- and d3 there as well in the forms without ConsToPrim -- 256, no scratch: both within +-0.5 % of this
  // placement on the headline, the WENOZ RK3 cycle and the general stage, same box.)
  constexpr int LOADS = (LEAN && EXTRA == EXTRA_NONE) ? 3 : 2;
  extern __shared__ __attribute__((aligned(16))) double ring[];
  // FC with room in the LDS (m12f_keeps_raw_rows: the hydro marches, PLM-class GLM-MHD): the rows are ALSO kept as loaded,
  // in a second ring -- the cell a wave retires is one of them, and its conserved value is what the update reads a second
  // time from memory otherwise (as u1 in stages with gam0 = 0, as the old u0 in the others): 40 of 160 - 200 B per cell
  // of a hydro stage.
  constexpr bool RAW = FC && m12f_keeps_raw_rows<FLUID, RECON>();
  double *const rawring = ring + NS * NV * 64;
  const int lane = threadIdx.x;
  const int w = (int)(blockIdx.x % 8u) * per_xcd + (int)(blockIdx.x / 8u);
  if (w >= nwaves) return;
  double lane_min_dt = 1.7976931348623157e308;
  const int64_t st = u0.sj;
  const int64_t run = (int64_t)u0.nx3 * u0.ni;
#if APK_M12F_TIMING
  unsigned long long phase_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tick_ = clock64();
#endif

  // Work of this wave (wave-uniform), the LOCKSTEP schedule: wave w marches WHOLE columns w, w + nwaves, ... of the
  // (block, chunk) list and then a segment of the columns left over, neighbouring waves taking neighbouring columns over
  // the same rows -- all waves of an XCD are then at (about) the same row at the same time, and the two waves that share
  // a cache line at the seam of their 58-cell chunks write it within an iteration of each other: the L2 merges the two
  // partial writes into one full line.  Written apart (one list of wave-rows cut into equal contiguous ranges that start
  // 20 rows apart: the schedule of rounds 2 - 3) every seam line goes to memory twice as a partial write.
  // tools/ubench/ubench_march_traffic.hip: the march's 27 load + 18 store streams, no arithmetic, 1.86 ms with the equal
  // split and 1.36 ms in lockstep (profiles/r04_ubench_march_traffic.jsonl).
  const int ncol = (int)(total_rows / u0.nx2);
  const int full = ncol / nwaves, left_cols = ncol - full * nwaves;
  const int per_col = (left_cols > 0) ? nwaves / left_cols : 0;  // segments per leftover column (>= 1)
  const int seg = per_col > 0 ? (u0.nx2 + per_col - 1) / per_col : 0;
  for (int piece = 0; piece <= full; ++piece) {  // wave-uniform
    // ---- this piece: rows s..e of column chunk `chunk` of block b
    int item, j0, nrows;
    if (piece < full) {
      item = piece * nwaves + w, j0 = 0, nrows = u0.nx2;
    } else {
      if (left_cols <= 0) break;
      const int sidx = w / left_cols, lc = w - sidx * left_cols;
      if (sidx >= per_col || sidx * seg >= u0.nx2) break;
      item = full * nwaves + lc, j0 = sidx * seg;
      nrows = (j0 + seg <= u0.nx2) ? seg : u0.nx2 - j0;
    }
    const int b = item / wpb;
    const int chunk = item - b * wpb;
    apk_block_desc b0 = u0.blocks[b];
    b0.cons = uniform_ptr(b0.cons);
    const double *c1 = uniform_ptr(u1.blocks[b].cons);
    double *prim_dst = (EXTRA != EXTRA_NONE && !sp.no_prim_store) ? uniform_ptr(u1.blocks[b].prim) : nullptr;
    // (the x3 sweep's flux differences: in the cells' layout, or compact -- StageParams.du_pitch)
    const int64_t d3_sn = sp.du_pitch > 0 ? (int64_t)sp.du_pitch * u0.nx2 * u0.nx3 : u0.sn;
    const double *d3 = sp.du + (int64_t)b * d3_sn * u0.nvar;

    const int64_t t = (int64_t)chunk * CPW + lane - FIRST;
    const bool in_run = (t >= 0) && (t < run);
    const int64_t tc = in_run ? t : (t < 0 ? 0 : run - 1);  // out-of-run lanes shadow a valid column
    const int krow = (int)(tc / u0.ni);
    const int i = (int)(tc - (int64_t)krow * u0.ni);
    const bool active = in_run && (lane >= FIRST) && (lane <= LAST) && (i >= u0.is) && (i <= u0.ie);
    const int s = u0.js + j0, e = s + nrows - 1;
    const int64_t base = (int64_t)(u0.ks + krow) * u0.sk + i;
    const int d3col = (i < u0.is) ? 0 : ((i > u0.ie) ? u0.nx1 - 1 : i - u0.is);
    // (row c - 1 of this column in d3: d3base + (c - 1) * d3st, like `done` in the cells' layout)
    const int64_t d3st = sp.du_pitch > 0 ? (int64_t)sp.du_pitch : st;
    const int64_t d3base = sp.du_pitch > 0 ? ((int64_t)krow * u0.nx2 - u0.js) * sp.du_pitch + d3col : base;
    // (lean forms, RowCellAt: the lane's byte offset in a row of the cell arrays / of d3; the rows are wave-uniform.  d3's
    // compact layout counts rows from js, which goes into the row part so that the lane's share stays non-negative)
    const unsigned cell_boff = (unsigned)(base * (int64_t)sizeof(double));
    const unsigned d3_boff = (unsigned)((sp.du_pitch > 0 ? (int64_t)krow * u0.nx2 * sp.du_pitch + d3col : base) * (int64_t)sizeof(double));
    const int64_t d3row0 = sp.du_pitch > 0 ? -(int64_t)u0.js * sp.du_pitch : (int64_t)0;
    // Lanes that retire no cell (overlap lanes at the ends of the wave, ghost columns: 13 % of the lanes on 128^3 blocks)
    // sit out the x2 Riemann solve -- they only have to carry their column through the ring as the x1 stencil of their
    // neighbours.  The stage runs at 94 % of the socket's 1400 W, so every lane-operation not executed counts.  (Masking
    // the x1 solve and the x2 reconstruction as well costs more in exec-mask bookkeeping than it saves: round 3; again in
    // round 5, when the cycle's clock had turned out to follow what its kernels leave of the power budget: +-0.3 %.)
    const bool need_x2 = active;
    // lanes whose x1 stencil (by wave shifts) is complete: the interface above the cell needs lanes l-1 .. l+2, the
    // cell's states l-2 .. l+2 and the interface of lane l-1
    // -- and whose values a retiring cell uses: the states of columns is-1 .. ie+1, the interfaces above is-2 .. ie+1.
    // (x2: the lanes that retire a cell, need_x2.)
    // As the threshold of PPM's extremum tests (hydro_math.hpp: kPpmNever), one register pair each.
    const double x1_thr_face = opaque((lane >= 1 && lane <= 61 && in_run && i >= u0.is - 2 && i <= u0.ie + 1) ? 0.0 : kPpmNever);
    const double x1_thr_cell = opaque((lane >= 2 && lane <= 61 && in_run && i >= u0.is - 1 && i <= u0.ie + 1) ? 0.0 : kPpmNever);
    const double x2_thr = opaque(need_x2 ? 0.0 : kPpmNever);

    const double dx1 = b0.dx[0], dx2 = b0.dx[1];
    const double area1 = to_sgpr(b0.dx[1] * b0.dx[2]);  // (per block: wave-uniform)
    const double area2 = to_sgpr(b0.dx[0] * b0.dx[2]);
    const double vol = to_sgpr(b0.dx[0] * b0.dx[1] * b0.dx[2]);
    const double upd = LEAN ? update_coefficient(sp, vol) : 0.0;
    // the stage's input: u0's primitives, or the conserved state of u1 / u0 (FC, fused_kernel.hpp: cons_row_to_prim)
    const apk_block_desc *srcb = (FC && sp.prim_from_cons != 2) ? u1.blocks : u0.blocks;  // (wave-uniform)
    auto input_of = [&](int blk) -> const double * { return FC ? srcb[blk].cons : u0.blocks[blk].prim; };
    const double *in0 = input_of(b);
    const double *prim_generic = in0 + base;
    // Direct neighbour addressing (sp.face_nbr): a lane on a ghost column reads the interior column
    // of the block behind that x1 face; the stencil rows below js / above je of an interior column
    // come from the block behind the x2 face (wave-uniform offset, not applied on ghost columns:
    // those lanes retire nothing and may read their neighbour's stale ghost rows).
    int64_t nbr_lo = 0, nbr_hi = 0;
    bool gcol = false;
    if (sp.face_nbr) {
      const int *fn = sp.face_nbr + 6 * b;
      if (fn[2] >= 0) nbr_lo = (input_of(fn[2]) - in0) + (int64_t)u0.nx2 * st;
      if (fn[3] >= 0) nbr_hi = (input_of(fn[3]) - in0) - (int64_t)u0.nx2 * st;
      gcol = (i < u0.is) || (i > u0.ie);
      if (gcol) {
        const int nb = fn[i < u0.is ? 0 : 1];
        if (nb >= 0) prim_generic = input_of(nb) + base + (i < u0.is ? u0.nx1 : -u0.nx1);
      }
    }
    // x1_halo, receive side: a ghost-column lane whose face has a segment reads its rows there -- [var][k][j][depth]
    // (rows outside the interior, which only the x2 stencil of interior columns needs, repeat the nearest row: valid
    // addresses, unused values)
    [[maybe_unused]] int64_t lsn = u0.sn;  // (X1H: per lane)
    [[maybe_unused]] int gseg = 0;       // (X1H: per lane, tested afresh at every use -- see X1Store)
#ifndef APK_X1H_NO_RECV
    if constexpr (X1H) {
      if (sp.x1_blocks && sp.x1_recv_depth > 0 && ((i < u0.is) || (i > u0.ie))) {
        const int dpt = sp.x1_recv_depth, side = (i < u0.is) ? 0 : 1;
        const int col = side ? i - (u0.ie + 1) : i - (u0.is - dpt);
        const double *seg = sp.x1_blocks[b].recv[side];
        if (seg && col >= 0 && col < dpt) {
          prim_generic = seg + (int64_t)krow * dpt * u0.nx2 + col;
          lsn = (int64_t)dpt * u0.nx2 * u0.nx3;
          gseg = 1;
        }
      }
    }
#endif
    const auto prim = as_global(prim_generic);
    auto row_off = [&](int r) -> int64_t {
      const int64_t d = (r < u0.js) ? nbr_lo : ((r > u0.je) ? nbr_hi : (int64_t)0);  // wave-uniform
      const int64_t ord = (int64_t)r * st + (gcol ? (int64_t)0 : d);
      if constexpr (X1H) {
        const int rc = (r < u0.js) ? 0 : ((r > u0.je) ? u0.nx2 - 1 : r - u0.js);  // wave-uniform
        asm volatile("" : "+v"(gseg));
        return gseg ? (int64_t)rc * sp.x1_recv_depth : ord;
      } else {
        return ord;
      }
    };
    // the NV variables of row r of this lane's column.  X1H: the stride between them is the lane's own, so the addresses
    // are a chain of per-lane additions -- pinned, or the compiler keeps the eight multiples n * lsn in registers across
    // the march (+16 VGPRs where 15 are free)
    auto load_row = [&](int r, double (&row)[NV]) {
      if constexpr (X1H) {
        auto p = prim + row_off(r);
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          row[n] = *p;
          if (n + 1 < NV) {
            p += lsn;
            asm volatile("" : "+v"(p), "+v"(lsn));
          }
        }
      } else {
#pragma unroll
        for (int n = 0; n < NV; ++n) row[n] = prim[n * u0.sn + row_off(r)];
      }
    };
    // x1_halo, send side
    [[maybe_unused]] X1Store xs;
#ifndef APK_X1H_NO_SEND
    if constexpr (X1H) xs = x1_store_of<true>(sp, u0, b, i, active, (int64_t)krow * sp.x1_send_depth * u0.nx2, sp.x1_send_depth);
#endif

    int c = s - 1;
    const int r0 = c - H;
    {
      // stencil rows of the first cell: all loads in flight together, then the ring writes
      double init[NS][NV];
#pragma unroll
      for (int m = 0; m < NS; ++m)
        if constexpr (X1H) {
          load_row(r0 + m, init[m]);
        } else {
#pragma unroll
          for (int n = 0; n < NV; ++n) init[m][n] = prim[n * u0.sn + row_off(r0 + m)];
        }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int m = 0; m < NS; ++m) {
        if constexpr (RAW) {
#pragma unroll
          for (int n = 0; n < NV; ++n) rawring[(m * NV + n) * 64 + lane] = init[m][n];
        }
        if constexpr (FC) (void)cons_row_to_prim<FLUID>(sp, init[m]);
#pragma unroll
        for (int n = 0; n < NV; ++n) ring[(m * NV + n) * 64 + lane] = init[m][n];
      }
    }
    double Pn[NV];  // row c+H (FC: as loaded until the x2 reconstruction that first uses it)
    if constexpr (X1H) {
      load_row(c + H, Pn);
    } else {
#pragma unroll
      for (int n = 0; n < NV; ++n) Pn[n] = prim[n * u0.sn + row_off(c + H)];
    }

    double wl_prev[NV], f_prev[NV];  // x2: permuted L state at face c / flux at face c-1
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      wl_prev[q] = 0.0;
      f_prev[q] = 0.0;
    }
    double face_carry[NV];  // PPM: the limited x2 interface value below cell c (see ppm_interface)
    if constexpr (RECON == APK_RC_PPM) {
#pragma unroll
      for (int n = 0; n < NV; ++n)
        face_carry[n] = ppm_interface(ring[(0 * NV + n) * 64 + lane], ring[(1 * NV + n) * 64 + lane],
                                      ring[(2 * NV + n) * 64 + lane], ring[(3 * NV + n) * 64 + lane]);
    }

    int slot0 = 0;  // slot holding row c-H
    for (; c <= e + 1; ++c) {
      const int64_t done = base + (int64_t)(c - 1) * st;  // the cell this iteration retires
      // (its place in d3; lanes that retire nothing get a valid column)
      const int64_t d3done = d3base + (int64_t)(c - 1) * d3st;
      const bool retire = (c >= s + 1);                   // wave-uniform
      APK_TICK(0);
      // ---- (1) x1 faces of row c-1 (every lane takes part in the wave shifts).  The row's own
      // values are still in the ring (slot of row c-1: c-1 = (c-H) + H-1); the lanes are
      // consecutive cells of the row, so the stencil neighbours i-2 .. i+2 come from the
      // neighbouring lanes by DPP wave shifts: no memory access at all (re-reading the row from
      // memory three iterations after it was loaded misses the L2 -- 256 waves per XCD stream ~5 MB
      // per iteration through it -- and left the waves parked on s_waitcnt for 3/4 of the kernel).
      // The x1 phase comes first so that its working set does not overlap the x2 solve's: only
      // its 9 flux differences stay live.
      double du[NV], d3v[NV], u1v[NV];
      double rawv[RAW ? NV : 1];  // the retiring cell's row as loaded
      if (retire) {
        const int slot_cm1 = (slot0 + H - 1) & (NS - 1);
        double ql1[NV], qr1[NV];
        // (the ring read of variable n+1 is issued before variable n is processed: the branchy PPM
        // code keeps the compiler from hoisting it, and with two waves per SIMD an exposed LDS round
        // trip per variable is a visible share of the iteration)
        if constexpr (RAW) {
#pragma unroll
          for (int n = 0; n < NV; ++n) rawv[n] = rawring[(slot_cm1 * NV + n) * 64 + lane];
        }
        double q0_next = ring[(slot_cm1 * NV + 0) * 64 + lane];
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          const double q0 = q0_next;
          if (n + 1 < NV) q0_next = ring[(slot_cm1 * NV + n + 1) * 64 + lane];
          const double qm1 = lane_below<1>(q0, lane), qp1 = lane_above<1>(q0, lane);
          if constexpr (RECON == APK_RC_PPM) {
            const double qm2 = lane_below<2>(q0, lane), qp2 = lane_above<2>(q0, lane);
            // (the wave shifts hand lanes 0, 1, 62, 63 zeros for the neighbours they lack; their interface values and
            // states are not used -- x1_thr_* -- and must not drag the wave through the limiter branches: with a zero
            // in the stencil nearly every variable "has an extremum" there, in every row)
            const double face_p = ppm_interface(qm1, q0, qp1, qp2, x1_thr_face);
            const double face_m = lane_below<1>(face_p, lane);
            ppm_cell(qm2, qm1, q0, qp1, qp2, face_m, face_p, ql1[n], qr1[n], x1_thr_cell);
          } else if constexpr (H >= 2) {
            const double qm2 = lane_below<2>(q0, lane), qp2 = lane_above<2>(q0, lane);
            reconstruct<RECON>(qm2, qm1, q0, qp1, qp2, dx1, n, ql1[n], qr1[n]);
          } else {
            reconstruct<RECON>(0.0, qm1, q0, qp1, 0.0, dx1, n, ql1[n], qr1[n]);
          }
          if constexpr (RECON == APK_RC_WENOZ || RECON == APK_RC_WENO3 || RECON == APK_RC_LIMO3)
            asm volatile("" : "+v"(ql1[n]), "+v"(qr1[n]));
        }
        APK_TICK(1);  // x1 reconstruction
        double wl[NV], wr[NV], f1[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          wl[q] = lane_below<1>(ql1[perm<1>(q)], lane);
          wr[q] = qr1[perm<1>(q)];
        }
        riemann<FLUID, RS>(wl, wr, sp.k, f1);
        double fup0 = 0.0;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const double fup = lane_above<1>(f1[q], lane);
          if (q == 0) fup0 = fup;
          du[perm<1>(q)] = (area1 * fup - area1 * f1[q]);
        }
        if constexpr (!LEAN) {
          if (sp.mflux && active) {
            double *m = sp.mflux + ((int64_t)0 * u0.nblocks + b) * u0.sn + done;
            m[0] = f1[0];
            m[1] = fup0;
          }
        }
        // (2) streaming operands of the cell being retired: in flight during the x2 phase
        if constexpr (LOADS == 3) {  // u1 early, d3 late
          asm volatile("" ::: "memory");
          if (RAW && sp.prim_from_cons == 1) {  // (wave-uniform: the input state IS u1, and the row is at hand)
#pragma unroll
            for (int n = 0; n < NV; ++n) u1v[n] = rawv[RAW ? n : 0];
          } else {
            if constexpr (LEAN) load_vars<NV>(c1, u0.sn, RowCellAt{(int64_t)(c - 1) * st, cell_boff}, u1v);
            else load_vars<NV>(c1, u0.sn, CellAt{done}, u1v);
          }
        }
      }
      APK_TICK(2);  // x1 Riemann + flux difference
      // ---- (3) x2: reconstruct cell c from ring rows c-H..c+H-1 and the register row c+H
      double Praw[RAW ? NV : 1];
      if constexpr (RAW) {
#pragma unroll
        for (int n = 0; n < NV; ++n) Praw[n] = Pn[n];
      }
      if constexpr (FC) (void)cons_row_to_prim<FLUID>(sp, Pn);  // (every lane: the row goes into the ring as the x1 stencil of its neighbours)
      double qln[NV], qrn[NV];
      {
      double an[NS];  // ring rows of the next variable (software-pipelined LDS reads)
#pragma unroll
      for (int m = 0; m < NS; ++m) an[m] = ring[(((slot0 + m) & (NS - 1)) * NV + 0) * 64 + lane];
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        double a[NS];
#pragma unroll
        for (int m = 0; m < NS; ++m) a[m] = an[m];
        if (n + 1 < NV) {
#pragma unroll
          for (int m = 0; m < NS; ++m) an[m] = ring[(((slot0 + m) & (NS - 1)) * NV + n + 1) * 64 + lane];
        }
        if constexpr (H == 1) {
          const double a0 = a[0], a1 = a[1];
          reconstruct<RECON>(0.0, a0, a1, Pn[n], 0.0, dx2, n, qln[n], qrn[n]);
        } else {
          const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
          if constexpr (RECON == APK_RC_PPM) {
            const double face_p = ppm_interface(a1, a2, a3, Pn[n], x2_thr);
            ppm_cell(a0, a1, a2, a3, Pn[n], face_carry[n], face_p, qln[n], qrn[n], x2_thr);
            face_carry[n] = face_p;
          } else {
            reconstruct<RECON>(a0, a1, a2, a3, Pn[n], dx2, n, qln[n], qrn[n]);
          }
        }
        if constexpr (RECON == APK_RC_WENOZ || RECON == APK_RC_WENO3 || RECON == APK_RC_LIMO3)
          asm volatile("" : "+v"(qln[n]), "+v"(qrn[n]));  // (see fused_march_kernel)
      }
      }
      const bool more = (c < e + 1);
      if (more) {
#pragma unroll
        for (int n = 0; n < NV; ++n) ring[(slot0 * NV + n) * 64 + lane] = Pn[n];
        if constexpr (RAW) {
#pragma unroll
          for (int n = 0; n < NV; ++n) rawring[(slot0 * NV + n) * 64 + lane] = Praw[n];
        }
        slot0 = (slot0 + 1) & (NS - 1);
        if constexpr (X1H) {
          load_row(c + 1 + H, Pn);
        } else {
#pragma unroll
          for (int n = 0; n < NV; ++n) Pn[n] = prim[n * u0.sn + row_off(c + 1 + H)];
        }
      }
      APK_TICK(3);  // x2 reconstruction, ring write, next-row loads issued
      // ---- (4) x2 face between rows c-1 and c; retire row c-1
      if (c >= s) {
        double f[NV];
        {
          double wr[NV];
#pragma unroll
          for (int q = 0; q < NV; ++q) wr[q] = qrn[perm<2>(q)];
          if (need_x2) riemann<FLUID, RS>(wl_prev, wr, sp.k, f);
        }
        APK_TICK(4);  // x2 Riemann
        if (retire) {
          if constexpr (LOADS == 3) {
            asm volatile("" ::: "memory");
            if (active) {
              if constexpr (LEAN) load_vars<NV>(d3, d3_sn, RowCellAt{d3row0 + (int64_t)(c - 1) * d3st, d3_boff}, d3v);
              else load_vars<NV>(d3, d3_sn, CellAt{d3done}, d3v);
            } else {
#pragma unroll
              for (int n = 0; n < NV; ++n) d3v[n] = 0.0;
            }
          }
          if constexpr (LOADS == 2) {
            asm volatile("" ::: "memory");
            // (Timing experiment: without these 18 loads a general stage takes 2.65 instead of 2.91 ms
            // -- their exposed latency is the price of not holding 36 VGPRs through the x2 solve.)
            if (active) {  // (ghost-column and overlap lanes retire nothing: 13 % of the lanes)
              if constexpr (LEAN) load_vars<NV>(d3, d3_sn, RowCellAt{d3row0 + (int64_t)(c - 1) * d3st, d3_boff}, d3v);
              else load_vars<NV>(d3, d3_sn, CellAt{d3done}, d3v);
              if (RAW && sp.prim_from_cons == 1) {  // (wave-uniform: the input state IS u1)
#pragma unroll
                for (int n = 0; n < NV; ++n) u1v[n] = rawv[RAW ? n : 0];
              } else {
                if constexpr (LEAN) load_vars<NV>(c1, u0.sn, RowCellAt{(int64_t)(c - 1) * st, cell_boff}, u1v);
                else load_vars<NV>(c1, u0.sn, CellAt{done}, u1v);
              }
            } else {
#pragma unroll
              for (int n = 0; n < NV; ++n) d3v[n] = 0.0, u1v[n] = 0.0;
            }
          }
#if APK_M12F_TIMING
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          APK_TICK(5);  // d3 / u1 requested and arrived
#endif
          // du = (x1 term + x2 term) + x3 term, the reference's accumulation order
#pragma unroll
          for (int q = 0; q < NV; ++q) {
            const int n = perm<2>(q);
            du[n] = du[n] + (area2 * f[q] - area2 * f_prev[q]);
          }
#pragma unroll
          for (int n = 0; n < NV; ++n) du[n] = du[n] + d3v[n];
          if (active) {
            if constexpr (!LEAN) {
              if (sp.mflux) {
                double *m = sp.mflux + ((int64_t)1 * u0.nblocks + b) * u0.sn + done;
                m[0] = f_prev[0];
                m[st] = f[0];
              }
            }
            if constexpr (LEAN) {
              const RowCellAt at{(int64_t)(c - 1) * st, cell_boff};
              if constexpr (RAW) {
                // (prim_from_cons = 2: the input state is the old u0 the update reads)
                if constexpr (X1H) {
                  if (sp.prim_from_cons == 2) finish_cell_old_held<FLUID, EXTRA, LEAN>(u0, b0, u1v, at, du, vol, sp, lane_min_dt, prim_dst, upd, rawv, &xs, c - 1 - u0.js);
                  else finish_cell_at<FLUID, EXTRA, LEAN>(u0, b0, u1v, at, du, vol, sp, lane_min_dt, prim_dst, upd, &xs, c - 1 - u0.js);
                } else {
                  if (sp.prim_from_cons == 2) finish_cell_old_held<FLUID, EXTRA, LEAN>(u0, b0, u1v, at, du, vol, sp, lane_min_dt, prim_dst, upd, rawv);
                  else finish_cell_at<FLUID, EXTRA, LEAN>(u0, b0, u1v, at, du, vol, sp, lane_min_dt, prim_dst, upd);
                }
              } else {
                if constexpr (X1H) finish_cell_at<FLUID, EXTRA, LEAN>(u0, b0, u1v, at, du, vol, sp, lane_min_dt, prim_dst, upd, &xs, c - 1 - u0.js);
                else finish_cell_at<FLUID, EXTRA, LEAN>(u0, b0, u1v, at, du, vol, sp, lane_min_dt, prim_dst, upd);
              }
            } else {
              finish_cell<FLUID, EXTRA, LEAN>(u0, b0, u1v, done, du, vol, sp, lane_min_dt, prim_dst, upd);
            }
          }
          APK_TICK(6);  // update, Dedner, ConsToPrim, dt, stores issued
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) f_prev[q] = f[q];
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) wl_prev[q] = qln[perm<2>(q)];
    }
  }
#if APK_M12F_TIMING
  if (lane == 0) {
    phase_acc[0] += clock64() - tick_;  // everything outside the ticked phases: prologues, loop control
#pragma unroll
    for (int p = 0; p < 8; ++p) atomicAdd(&g_m12f_phase[p], phase_acc[p]);
  }
#endif
  if constexpr (EXTRA == EXTRA_C2P_DT) {
    double m = lane_min_dt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_down(m, off, 64));
    if (lane == 0) atomicMin(sp.dt_bits, (unsigned long long)__double_as_longlong(m));
  }
}

// Round 3, measured and NOT kept (same box, general stage of 8 x 128^3, 3.39 ms with this kernel):
//  * the finish of a row software-pipelined over the loop's back edge (operands requested at the end of iteration c,
//    consumed in iteration c+1 between the x1 reconstruction and the x1 solve, behind ~600 low-pressure instructions):
//    72 more VGPRs live across the back edge than 256 hold next to an HLLD solve -- 580 B of scratch per lane, 8.4 ms;
//  * scalar-base addressing of d3 / u1 / u0 / prim' (one SGPR base per array advanced by the scalar unit + ONE 32-bit
//    per-lane byte offset: global_load_dwordx2 v, v_off, s[base:base+1] instead of a 64-bit VALU add per access): the
//    compiler emitted exactly that, but hoisted array + n * sn out of the march as loop-invariant register pairs, which
//    pushed the scalar spills from 87 to 124 and the vector spills from 6 to 36 registers -- 3.50 ms.  (Round 5: IN, as
//    RowCellAt in fused_kernel.hpp -- the scalar pointer is re-materialised after every step, so nothing is hoisted:
//    scalar spills 96 -> 75 static / 58 -> 26 v_readlane per iteration, 56 -> 10 address adds per iteration, 240 -> 230
//    VGPRs; corrector's march 1.98 -> 1.93 ms, x3 sweep 0.688 -> 0.678 ms, headline cycle +1.9 - 2.5 % on one box);
//  * the x1 stencil from the ring row at lane offsets -2 .. +2 (5 ds_read_b64 instead of 1 + 8 DPP moves per variable,
//    -72 VALU instructions per iteration): 3.40 ms, no change -- the kernel is not short of VALU issue slots alone
//    (measured again in round 5 as ds_read2_b64 pairs on top of the row addressing: 3.701 against 3.68 ms per cycle with
//    the wave shifts -- still nothing, removed again);
//  * PPM's extremum branches switched off altogether (wrong results; the ceiling of any scheme that makes them
//    cheaper): 2.97 ms, -12 %.  COMPACTING the cells that take them -- the lanes park the seven operands of
//    ppm_cell's extremum limiter in 2 KB of LDS beside the ring, item after item across the nine variables, the first
//    lanes of the wave then run the limiter once for all of them and the owners read their results back; bit-exact --
//    was built for the x3 sweep and measured: 0.809 against 0.770 ms.  Ballots, rank computation, the staging writes
//    and the second pass cost more than the seven-odd masked executions of the ~60-instruction branch they replace;
//    (Round 5, in this kernel, after the dead lanes had left the branches: the limiter of ppm_cell DEFERRED -- a lane parks
//    the seven operands of the variable that needs it under its own execution mask, eight moves in a region only entered
//    when some lane has an extremum there, and ONE masked pass per direction limits all pending items; 255 VGPRs, no
//    scratch, bit-exact: 667 -> 643 M vector instructions per launch, 49.0 -> 52.4 live lanes, but +20 % scalar
//    instructions and +44 % branches: 1.764 -> 1.748 ms on average over four same-box pairs, the cycle -0.3 %, the general
//    stage +0.3 %.  The limiter blocks that are entered for real extrema cost less than their static size suggested;
//    not kept.)
//  * PPM without divergent branches (the extremum limiters of ppm_interface / ppm_cell evaluated by every lane and
//    selected, one scheduling pin per variable so that nothing spills): x3 sweep 0.76 -> 0.93 ms, this kernel 2.31 ->
//    2.55 ms.  The masked branches are CHEAP: an instruction with one or two live lanes does not cost a full wave's
//    issue time (which is also why switching them off altogether buys 12 %, not the 30 % their instruction count suggests);
//  * a hardware fact found on the way (tools/ubench/ubench_mask.hip): an fp64 VALU instruction with 1 - 8 live lanes takes
//    16 shader cycles of its wave against 4 with 12 - 64 lanes, two waves doing such work serialise on it, and a
//    full-width wave next to them is NOT held up.  Letting lanes 0 .. 11 come along into PPM's extremum branches (so that
//    they run at the full-width rate) makes the kernels SLOWER -- x3 sweep 0.78 -> 0.82 ms, this kernel 2.40 -> 2.49 --
//    because the few-lane instructions cost their wave time but the pipe next to nothing, and padded they cost the pipe;
//  * three waves per SIMD for the HYDRO variants of this kernel (129 - 145 VGPRs, 5 - 10 KB of ring: the wave count sized
//    from hipFuncGetAttributes instead of two everywhere): PLM+HLLC 1.33 -> 1.35 ms, no gain from occupancy either;
//  * host-evaluated stage constants (hydro_math.hpp: StageConsts; no scratch left in this kernel, 14 VGPRs fewer)
//    and global_ instead of flat_ accesses (as_global): both kept, both within 1 % (same-box bench A/B);
//  * the nine per-variable offsets n * sn of d3 / u1 / u0 / prim' as ONE walking pointer (8 SGPR pairs fewer: scalar
//    spills 87 -> 72, scratch 28 -> 20 B per lane): no change (2.60 against 2.58 ms for this kernel).
// number of waves the device holds at once with two march waves per SIMD
inline int resident_march_waves() {
  static const int n = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    }
    return cus * 4 * 2;
  }();
  return n;
}

// can / should a stage take the two-kernel form?  3-D, a reconstruction with a stencil (ghost
// layers), FillDerived out of place or absent (K2's lanes read their x1 neighbours' primitives from
// memory), and rows long enough that the flattened (k, i) run keeps most lanes on interior cells
// (nx1 / (nx1 + 2 ng): 128 -> 96 %, 32 -> 84 %, 16 with nghost 4 -> 67 %).  Measured on the refined mesh of
// BASELINE config 5 (232 blocks of 16^3, MHD PPM+HLLD, nghost 4): 8.52e8 against 8.36e8 zone-cycles/s for the
// three-sweep schedule with several rows per wave, so 16-cell blocks take it too; narrower ones do not.
inline bool two_kernel_stage_applies(const PackView &u0, int recon, int extra, const StageParams &sp) {
  static const int mode = std::getenv("APK_STAGE_MODE") ? std::atoi(std::getenv("APK_STAGE_MODE")) : 2;  // A/B switch: 3 = three sweeps
  if (mode == 3) return false;
  constexpr int min_nx1 = 16;
  // (blocks narrower than 32 cells only if they are deep enough along x3 for the plane windows of a split stage --
  // 4 nghost planes: the driver's overlap rule -- so that taking this form never costs an overlapped exchange)
  const bool wide_enough = u0.nx1 >= 32 || (u0.nx1 >= min_nx1 && u0.nx3 >= 4 * u0.ng);
  // (the marches address a block's cells as scalar row pointer + 32-bit byte offset of the lane: RowCellAt)
  const bool offsets_fit = (uint64_t)u0.sn * sizeof(double) < (1ull << 32);
  return u0.ndim == 3 && recon != APK_RC_DC && wide_enough && offsets_fit && (extra == EXTRA_NONE || sp.prim_to_u1);
}

// does a stage of this form follow apk_stage_args.x1_halo?  (apk_stage_x1_halo; launch_fused_stage refuses the others)
inline bool x1_halo_stage_ok(const PackView &u0, int recon, int extra, const StageParams &sp) {
  if (u0.ndim != 3 || sp.mflux || !stage_is_lean(sp) || sp.window) return false;
  const int deepest = sp.x1_send_depth > sp.x1_recv_depth ? sp.x1_send_depth : sp.x1_recv_depth;
  if (u0.nx1 < 2 * deepest || sp.x1_send_depth < 0 || sp.x1_recv_depth < 0 || sp.x1_recv_depth > u0.ng) return false;
  if (sp.x1_send_field == 1 && extra == EXTRA_NONE) return false;  // (primitives to send: a stage that computes them)
  if (recon == APK_RC_DC)  // the two-row march, which has no form with the time-step estimate for it
    return sp.phase == 0 && (extra == EXTRA_NONE || (sp.prim_to_u1 && extra == EXTRA_C2P)) && u0.nx2 % 2 == 0 && u0.nx2 >= 4;
  // (from stored primitives or from a conserved state -- but not the stages that take the single march, fused3_kernel.hpp:
  // launch_fused_stage asks single_march_stage_applies for those and refuses)
  return two_kernel_stage_applies(u0, recon, extra, sp);
}

template <int FLUID, int RECON, int RS>
inline void launch_m12f(const PackView &u0, const PackView &u1, const StageParams &sp, int extra, hipStream_t s) {
  if constexpr (RECON != APK_RC_DC) {
    constexpr int lds = march_lds_bytes<FLUID, RECON>();
    constexpr int cpw = m12_last_lane(RECON) - m12_first_lane(RECON) + 1;
    const int64_t run = (int64_t)u0.nx3 * u0.ni;
    const int wpb = (int)((run + cpw - 1) / cpw);
    const long long total_rows = (long long)u0.nblocks * wpb * u0.nx2;
    // as many waves as the device holds, but no ranges shorter than ~16 rows
    long long nw = resident_march_waves();
    // (columns of 32 rows and more are cut down to 10-row ranges before waves are left out: Orszag-Tang 512 x 512 x 4 in
    // 128 x 128 x 4 blocks, 20480 wave-rows -- 1280 waves of 16 rows 0.215 ms, 2048 of 10 rows 0.163 ms, no further gain
    // below; the 16-row columns of a refined mesh's 16^3 blocks stay whole: cut in two they run 2 % slower)
    const int minrows = u0.nx2 >= 32 ? 10 : 16;
    if (total_rows / minrows < nw) nw = total_rows / minrows > 0 ? total_rows / minrows : 1;
    const int nwaves = (int)nw;
    const int per_xcd = (nwaves + 7) / 8;
    const dim3 g((unsigned)(per_xcd * 8), 1, 1);
    const int lean_level = stage_lean_level(sp);
    const bool lean = lean_level == 1;
    const bool lean2 = lean_level == LEAN_PFLOOR && extra != EXTRA_NONE && !sp.prim_from_cons && !sp.x1_blocks;
#define APK_LAUNCH_M12F(EXTRA_, LEAN_) \
  hipLaunchKernelGGL((fused_m12f_kernel<FLUID, RECON, RS, EXTRA_, LEAN_>), g, dim3(64), lds, s, u0, u1, sp, wpb, nwaves, per_xcd, total_rows)
    constexpr int lds_fc = m12f_keeps_raw_rows<FLUID, RECON>() ? 2 * lds : lds;  // (the rows as loaded, too)
#define APK_LAUNCH_M12F_FC(EXTRA_) \
  hipLaunchKernelGGL((fused_m12f_kernel<FLUID, RECON, RS, EXTRA_, 1, true>), g, dim3(64), lds_fc, s, u0, u1, sp, wpb, nwaves, per_xcd, total_rows)
#define APK_LAUNCH_M12F_X1H(EXTRA_) \
  hipLaunchKernelGGL((fused_m12f_kernel<FLUID, RECON, RS, EXTRA_, 1, false, true>), g, dim3(64), lds, s, u0, u1, sp, wpb, nwaves, per_xcd, total_rows)
#define APK_LAUNCH_M12F_FC_X1H(EXTRA_) \
  hipLaunchKernelGGL((fused_m12f_kernel<FLUID, RECON, RS, EXTRA_, 1, true, true>), g, dim3(64), lds_fc, s, u0, u1, sp, wpb, nwaves, per_xcd, total_rows)
    if (sp.x1_blocks && sp.prim_from_cons) {  // (the lean forms: launch_fused_stage has checked)
      if (extra == EXTRA_C2P_DT) APK_LAUNCH_M12F_FC_X1H(EXTRA_C2P_DT);
      else if (extra == EXTRA_C2P) APK_LAUNCH_M12F_FC_X1H(EXTRA_C2P);
      else APK_LAUNCH_M12F_FC_X1H(EXTRA_NONE);
    } else if (sp.x1_blocks) {
      if (extra == EXTRA_C2P_DT) APK_LAUNCH_M12F_X1H(EXTRA_C2P_DT);
      else if (extra == EXTRA_C2P) APK_LAUNCH_M12F_X1H(EXTRA_C2P);
      else APK_LAUNCH_M12F_X1H(EXTRA_NONE);
    } else if (sp.prim_from_cons) {  // (lean forms only: launch_fused_stage has checked)
      if (extra == EXTRA_C2P_DT) APK_LAUNCH_M12F_FC(EXTRA_C2P_DT);
      else if (extra == EXTRA_C2P) APK_LAUNCH_M12F_FC(EXTRA_C2P);
      else APK_LAUNCH_M12F_FC(EXTRA_NONE);
    } else if (extra == EXTRA_C2P_DT) {
      if (lean) APK_LAUNCH_M12F(EXTRA_C2P_DT, 1);
      else if (lean2) APK_LAUNCH_M12F(EXTRA_C2P_DT, LEAN_PFLOOR);
      else APK_LAUNCH_M12F(EXTRA_C2P_DT, 0);
    } else if (extra == EXTRA_C2P) {
      if (lean) APK_LAUNCH_M12F(EXTRA_C2P, 1);
      else if (lean2) APK_LAUNCH_M12F(EXTRA_C2P, LEAN_PFLOOR);
      else APK_LAUNCH_M12F(EXTRA_C2P, 0);
    } else {
      if (lean) APK_LAUNCH_M12F(EXTRA_NONE, 1);
      else APK_LAUNCH_M12F(EXTRA_NONE, 0);
    }
#undef APK_LAUNCH_M12F
#undef APK_LAUNCH_M12F_FC
#undef APK_LAUNCH_M12F_X1H
#undef APK_LAUNCH_M12F_FC_X1H
#if APK_M12F_TIMING
    {
      static int calls = 0;
      if (++calls % 8 == 0) {
        (void)hipStreamSynchronize(s);
        unsigned long long h[8] = {0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_m12f_phase), sizeof(h));
        unsigned long long z[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_m12f_phase), z, sizeof(z));
        static const char *name[8] = {"other", "x1_recon", "x1_riemann", "x2_recon", "x2_riemann", "load_wait", "finish", "-"};
        double tot = 0;
        for (int p = 0; p < 7; ++p) tot += (double)h[p];
        std::fprintf(stderr, "[m12f phases, shader clocks per wave over 8 launches, recon %d extra %d]", RECON, extra);
        for (int p = 0; p < 7; ++p) std::fprintf(stderr, " %s %.1f%% (%.3e)", name[p], 100.0 * h[p] / tot, (double)h[p] / nwaves / 8.0);
        std::fprintf(stderr, "\n");
      }
    }
#endif
  }
}

}  // namespace apk
