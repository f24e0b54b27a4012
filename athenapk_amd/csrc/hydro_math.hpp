// hydro_math.hpp -- pointwise device arithmetic of the hot path (gfx950, fp64).
//
// Reconstruction, Riemann solvers and EOS as wave64-friendly inline device functions.
// Everything is templated on the option enums so that component permutations and
// variable counts are compile-time constants: no dynamically indexed register arrays
// (which would be demoted to scratch memory), no run-time solver dispatch inside kernels.
//
// Floating-point contract: operation order and grouping follow the reference expressions
// (cited per function) so that the -ffp-contract=off build of this library reproduces
// the CPU oracle bit for bit; the default build lets the compiler contract a*b+c into
// v_fma_f64 (same expressions, <= few ulp apart, see DESIGN.md "Floating point").
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/apk_amd.h"

namespace apk {

#define APK_DEV __device__ __forceinline__

constexpr double kTiny = 1.0e-20;       // Parthenon TINY_NUMBER
constexpr double kHlldSmall = 1.0e-8;   // glmmhd_hlld.hpp:36
constexpr int NHYDRO = 5;
constexpr int NGLMMHD = 9;
// natural variable indices (src/main.hpp:19-33)
enum { IDN = 0, IM1 = 1, IM2 = 2, IM3 = 3, IEN = 4, IB1 = 5, IB2 = 6, IB3 = 7, IPS = 8 };
enum { IV1 = 1, IV2 = 2, IV3 = 3, IPR = 4 };

// x / 6.0 appears four times per PPM call (ppm_simple.hpp:55-56,78,95).  The parity build keeps
// the IEEE division; the default build multiplies by the reciprocal (<= 1 ulp apart), which
// removes ~40 fp64 instructions per call (a correctly rounded fp64 divide is ~11 instructions).
#ifdef APK_FP_STRICT
#define APK_DIV6(x) ((x) / 6.0)
#else
#define APK_DIV6(x) ((x) * (1.0 / 6.0))
#endif

APK_DEV double sqr(double x) { return x * x; }
// Field pointers come out of the block descriptors (memory), so the compiler knows them as generic pointers and emits
// flat_load / flat_store -- which count on lgkmcnt as well as vmcnt (a flat access may turn out to be LDS), so every
// s_waitcnt lgkmcnt(0) in front of a ring read also waits for the stores of the previous row to be acknowledged.
// Every field lives in global memory: say so.
template <class T>
APK_DEV __attribute__((address_space(1))) T *as_global(T *p) {
  return (__attribute__((address_space(1))) T *)p;
}
// Results of a stage kernel (new conserved state, new primitives, the x3 sweep's partial divergences): never read again by
// the kernel that writes them.  (Non-temporal stores were measured in round 4: no gain.)
template <class P>
APK_DEV void store_result(P p, double v) {
  *p = v;
}
// a wave-uniform value the vector ALU had to compute (there is no scalar fp64 unit), moved into a scalar register
// pair: it stops occupying two VGPRs of every lane for as long as it lives
APK_DEV double to_sgpr(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}

// ---- square roots and reciprocals ---------------------------------------------------------------
// The parity build evaluates the reference's IEEE operations.  In the product build hipcc expands
// an fp64 sqrt into 18 instructions around one v_rsq_f64 (range scaling for denormals, class
// checks, a Goldschmidt step and two residual corrections) and a divide into 8 around one
// v_rcp_f64, and the transcendental unit issues at a quarter of the fp64 rate (measured: 16 cycles
// per wave instruction against 4.3, tools/ubench): the 14 divides and 6 roots of one HLLD solve
// were ~40 % of its issue time.  The forms below keep the same Newton / Goldschmidt refinement
// (results within 1-2 ulp) without the range scaling -- the arguments here are densities,
// pressures and squared speeds, never denormal -- and hand out the by-products (1/sqrt(x) comes
// for free with sqrt(x); several quotients share one reciprocal).
#ifdef APK_FP_STRICT
#define APK_PLAIN_SQRT 1
APK_DEV double fsqrt(double x) { return sqrt(x); }
APK_DEV double fsqrt_pos(double x) { return sqrt(x); }
APK_DEV double fsqrt_nonneg(double x) { return sqrt(x); }
APK_DEV double frcp(double x) { return 1.0 / x; }
#else
// v_rsq_f64 / v_rcp_f64 deliver 24 good bits (measured on gfx950 over 1e-12 .. 1e12, tools/ubench/ubench_lat.hip:
// max relative error 4.6e-8 / 5.2e-8; round 2 assumed 8-10 bits and spent one quadratically converging step
// too many everywhere): 24 -> 48 -> 96 bits, i.e. TWO Newton steps for a reciprocal, one Goldschmidt step +
// ONE residual correction for a root, one Newton step for 1/sqrt from the converged pair.
APK_DEV void fsqrt_rsqrt(double x, double &root, double &inv_root) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);  // 48 bits
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);  // full
  // 1/sqrt(x) from the converged root: Newton on rs -> rs (2 - g rs), rs = 2 h has 48 bits
  double rs = h + h;
  double e = fma(-g, rs, 1.0);
  root = g;
  inv_root = fma(rs, e, rs);
}
APK_DEV double fsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  return (x == 0.0) ? 0.0 : g;  // rsq(0) = inf
}
// fsqrt without the `x == 0 ? 0 : ...` guard (a compare and two selects): for arguments that are positive whenever the
// state is valid (a squared speed).  fsqrt_nonneg clamps an argument that may be exactly zero -- a sum of squares --
// at 1e-300 instead (one v_max_f64): sqrt(0) comes out as 1e-150, far below anything it is added to.
APK_DEV double fsqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  return g;
}
APK_DEV double fsqrt_nonneg(double x) { return fsqrt_pos(fmax(x, 1.0e-300)); }
APK_DEV double frcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);  // 48 bits
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}
#endif
// a / b where nothing downstream hangs on the last bit (the weights of the WENO schemes, limiter slopes): the product
// build multiplies by frcp(b) -- v_rcp_f64 + two Newton steps + one multiply = 6 instructions, <= 1.5 ulp -- where the
// compiler's own -freciprocal-math quotient adds a residual correction (8 instructions, <= 1 ulp); WENO-Z has five
// per pencil and variable.  NOT used where a comparison against an exact tie follows (PPM's limited ratio) or where
// round-off is amplified (fast speeds next to HLLD's degenerate states, DESIGN.md "Floating point").
#ifdef APK_PLAIN_SQRT
APK_DEV double fdiv(double a, double b) { return a / b; }
#else
APK_DEV double fdiv(double a, double b) { return a * frcp(b); }
#endif
// The same with ONE Newton step on v_rcp_f64 (raw result 4.6e-8 relative, profiles/r03_clock_and_latency.json: one step
// squares it, 2.1e-15 = 19 ulp; four instructions and the multiply): the limited slope of PLM and the weights and
// normalisations of WENO-Z and WENO3, whose own truncation error is orders of magnitude above it and whose results feed no
// comparison.  Product build only; the parity build divides.
#if defined(APK_FP_STRICT) || defined(APK_PLAIN_SQRT) || defined(APK_RCP_FULL)
APK_DEV double frcp48(double x) { return frcp(x); }
APK_DEV double fdiv48(double a, double b) { return fdiv(a, b); }
#else
APK_DEV double frcp48(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  return fma(y, fma(-x, y, 1.0), y);
}
APK_DEV double fdiv48(double a, double b) { return a * frcp48(b); }
#endif
#ifdef APK_FP_STRICT
APK_DEV double min2(double a, double b) { return (b < a) ? b : a; }  // std::min
APK_DEV double max2(double a, double b) { return (a < b) ? b : a; }  // std::max
#else
// one v_min_f64 / v_max_f64 (IEEE minNum / maxNum: same value for ordered operands)
APK_DEV double min2(double a, double b) { return fmin(a, b); }
APK_DEV double max2(double a, double b) { return fmax(a, b); }
#endif
// max(|a|, |b|) / max(a, |b|) of values that come straight out of memory, LDS or a DPP move.  fmax on such operands
// makes the compiler canonicalise each of them first (v_max_f64 x, x, x: IEEE mode must quiet a signalling NaN it cannot
// rule out) -- five extra instructions in PPM's extremum limiter, whose scale is a maximum over the five stencil cells.
// v_max_f64 itself quiets its operands in IEEE mode, so the product build issues the instruction directly.
#ifdef APK_FP_STRICT
APK_DEV double max_abs2(double a, double b) { return max2(fabs(a), fabs(b)); }
APK_DEV double max_with_abs(double a, double b) { return max2(a, fabs(b)); }
APK_DEV double max_plain(double a, double b) { return max2(a, b); }
#else
APK_DEV double max_abs2(double a, double b) {
  double r;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
APK_DEV double max_with_abs(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
APK_DEV double max_plain(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#endif
// An opaque copy of x (no instruction): expressions built on it are not recognised as equal to expressions built on x,
// which keeps the compiler from hoisting code that two rarely taken branches share in front of both of them.
APK_DEV double opaque(double x) {
  asm("" : "+v"(x));
  return x;
}
// Parthenon SIGN(x) = (x < 0) ? -1 : 1 ; carried as a bool "is negative"
APK_DEV bool neg(double x) { return x < 0.0; }
// x > 0 with NaN -> false, decided on the bit pattern so that no fast-math assumption a build may
// be given can fold the NaN case away: a blown-up state must still be flagged
APK_DEV bool strictly_positive(double x) {
  // one v_cmp_class_f64: +denormal | +normal | +infinity (bits 7, 8, 9 of the class mask)
  return __builtin_amdgcn_class(x, 0x380);
}
APK_DEV double with_sign(bool negative, double mag) { return negative ? -mag : mag; }

template <int FLUID>
constexpr int nvars() {
  return FLUID == APK_FLUID_EULER ? NHYDRO : NGLMMHD;
}

// number of ghost layers a reconstruction needs (src/hydro/hydro.cpp:316-339)
constexpr int recon_nghost(int recon) {
  return recon == APK_RC_DC ? 1 : ((recon == APK_RC_PPM || recon == APK_RC_WENOZ) ? 3 : 2);
}
// stencil half width
constexpr int recon_halfwidth(int recon) {
  return recon == APK_RC_DC ? 0 : ((recon == APK_RC_PPM || recon == APK_RC_WENOZ) ? 2 : 1);
}

// ======================================================================================
// Reconstruction.  Each function reconstructs cell i and returns
//   ql = L state at face i+1/2  (the reference's ql_ip1)
//   qr = R state at face i-1/2  (the reference's qr_i)
// ======================================================================================

// src/recon/plm_simple.hpp:21-37
APK_DEV void plm(double qm1, double q0, double qp1, double &ql, double &qr) {
  const double dl = q0 - qm1;
  const double dr = qp1 - q0;
  const double prod = dl * dr;
  double slope = 0.0;
  if (prod > 0.0) slope = fdiv48(prod, (dl + dr));
  ql = q0 + slope;
  qr = q0 - slope;
}

// PPM, split so that a sweep evaluates every interface ONCE (src/recon/ppm_simple.hpp:39-162 computes
// the limited interface value q_{i+1/2} twice, as "qr_ip1 side" of cell i and as "ql_i side" of cell
// i+1, from the same four cells with the same expressions):
//   ppm_interface(q_{i-1}, q_i, q_{i+1}, q_{i+2})  steps 1 + 2a for the interface between cells i and i+1
//       (:50-98); written from the perspective of cell i ("face_p"); from the perspective of cell
//       i+1 the reference's "face_m" expressions dd_m / dd_c / d2_m / d2_c are the SAME operations
//       on the same operands (dd_m' = 0.5*da' + 0.5*(qm1'-qm2') = 0.5*db + 0.5*da = dd_c, ...), so
//       the value is bit-identical whichever cell asks for it;
//   ppm_cell(...)  steps 3 + 4 for one cell given its two limited interface values (:100-162).
// A march keeps face_p of the cell it just reconstructed as face_m of the next one; the x1 sweep
// passes it one lane to the right.
// (thr: 0.0, the reference's test.  A lane whose stencil is not valid -- the outermost lanes of a wave that gets its x1
// neighbours by wave shifts hold zeros there -- and whose value nothing uses passes kPpmNever = -inf: "product < -inf" keeps
// it out of the extremum branch, which otherwise runs for the whole wave, in nearly every pencil, on behalf of such lanes
// alone.  A per-lane constant in a register pair: the compare takes it instead of the literal 0, no instruction more --
// as a boolean ANDed into the branch condition it cost scalar registers the marches do not have, as `take && ...` a
// branch of its own per call.)
constexpr double kPpmNever = -__builtin_inf();
APK_DEV double ppm_interface(double qm1, double q0, double qp1, double qp2, double thr = 0.0) {
  // (Both builds evaluate the reference's grouping.  The algebraically equal 4-operation form
  // (7 (q_i + q_i+1) - (q_i-1 + q_i+2)) / 12 differs in the last bits, which is enough to flip the
  // extremum tests below on smooth data -- 3e-9 after two cycles of the 256^3 benchmark state, far
  // outside the 1e-12 agreement with the parity build -- so it is not used.)
  const double da = q0 - qm1;
  const double db = qp1 - q0;
  const double dd_c = 0.5 * db + 0.5 * da;
  const double dd_p = 0.5 * (qp2 - qp1) + 0.5 * db;
  const double face = 0.5 * (q0 + qp1) + APK_DIV6(dd_c - dd_p);
  // step 2a (:66-98): the limited value only replaces `face` at a local extremum (CD eq 84), so
  // the second differences and the limiter (one division) are evaluated inside that branch only
  const double below = face - q0;
  const double above = qp1 - face;
  if (below * above < thr) {
    constexpr double C2 = 1.25;
    const double d2_c = qm1 + qp1 - 2.0 * q0;
    const double d2_p = q0 + qp2 - 2.0 * qp1;
    const double d2f = 3.0 * (q0 + qp1 - 2.0 * face);
    const bool sg = neg(d2f);
    const bool agree = (sg == neg(d2_c)) && (sg == neg(d2_p));
    const double mag = min2(C2 * fabs(d2_c), min2(C2 * fabs(d2_p), fabs(d2f)));
    const double lim = agree ? with_sign(sg, mag) : 0.0;
    return 0.5 * (q0 + qp1) - APK_DIV6(lim);
  }
  return face;
}

// (the extremum limiter proper: steps 4 of ppm_simple.hpp:104-150 for a cell the extremum test has picked)
APK_DEV void ppm_cell_limited(double qm2, double qm1, double q0, double qp1, double qp2, double face_m, double face_p,
                              double dminus, double dplus, double &l, double &r) {
  constexpr double C2 = 1.25;
  // The second differences and the limited ratio are only consumed here, so they are only computed here.
  // (d2_c and d2_p are also what ppm_interface's extremum branch computes for the interface above this cell: left
  // alone, the compiler evaluates them in front of BOTH branches, i.e. in every lane of every pencil -- 4 full-width
  // instructions per variable and direction to save 4 few-lane ones where both limiters engage)
  const double q0x = opaque(q0), qp1x = opaque(qp1);
  const double d2_m = qm2 + q0x - 2.0 * qm1;
  const double d2_c = qm1 + qp1x - 2.0 * q0x;
  const double d2_p = q0x + qp2 - 2.0 * qp1x;
  const double d2_face = 6.0 * (face_m + face_p - 2.0 * q0x);
  const bool s = neg(d2_m);
  const bool agree = (s == neg(d2_c)) && (s == neg(d2_p)) && (s == neg(d2_face));
  const double mag = min2(min2(C2 * fabs(d2_m), C2 * fabs(d2_c)), min2(C2 * fabs(d2_p), fabs(d2_face)));
  const double d2lim = agree ? with_sign(neg(d2_face), mag) : 0.0;
  const double scale_lo = max_abs2(qm1, qm2);
  const double scale_hi = max_with_abs(max_abs2(q0, qp1), qp2);
  double ratio = 0.0;
  if (fabs(d2_face) > (1.0e-12) * max_plain(scale_lo, scale_hi)) ratio = d2lim / d2_face;
  l = face_p;
  r = face_m;
  if (ratio <= (1.0 - (1.0e-12))) {
    r = q0 - ratio * dminus;
    l = q0 + ratio * dplus;
  }
}
APK_DEV void ppm_cell(double qm2, double qm1, double q0, double qp1, double qp2, double face_m,
                      double face_p, double &ql, double &qr, double thr = 0.0) {
  const double dminus = q0 - face_m;
  const double dplus = face_p - q0;
  const double ext_a = dminus * dplus;
  const double ext_b = (qp1 - q0) * (q0 - qm1);
  // Local extremum: CS limiter on the parabola.  A lane whose parabola is FLAT -- both one-sided differences zero: the
  // field of an unmagnetised run, the ambient medium of a blast, a variable that does not depend on the sweep direction
  // -- passes the reference's test (0 <= 0) and would take the whole wave through the limiter to no effect: with
  // dminus = dplus = 0 both branches return q0 exactly.  Out of the extremum set with one addition and one compare
  // (|dminus| + |dplus|: a NaN stays in, as in the reference; & not &&: no branch of its own -- as `&&` and with a v_max,
  // before the outermost lanes stopped dragging every wave into the branch anyway, the same test measured as a loss).
  // Round 5, same box: headline (no flat variable) -0.5 %, general stage benchmark (two flat variables per direction)
  // 2.73 -> 2.62 ms, refined blast of config 5 (flat ambient medium) +3 %.
  const bool ext = (ext_a <= thr || ext_b <= thr) & ((fabs(dminus) + fabs(dplus)) != 0.0);

  double r = face_m, l = face_p;
  if (ext) {
    ppm_cell_limited(qm2, qm1, q0, qp1, qp2, face_m, face_p, dminus, dplus, l, r);
  } else {
    // (both corrections under the lanes' execution mask instead of selects: -36 v_cndmask, +16 moves and 36 branches
    // per iteration of the finishing march by the ledger -- not pursued)
    if (fabs(dminus) >= 2.0 * fabs(dplus)) r = q0 - 2.0 * dplus;
    if (fabs(dplus) >= 2.0 * fabs(dminus)) l = q0 + 2.0 * dminus;
  }
  ql = l;
  qr = r;
}

// src/recon/ppm_simple.hpp:39-162, one cell on its own (flux-array kernels, passive scalars)
APK_DEV void ppm(double qm2, double qm1, double q0, double qp1, double qp2, double &ql,
                 double &qr) {
  const double face_m = ppm_interface(qm2, qm1, q0, qp1);
  const double face_p = ppm_interface(qm1, q0, qp1, qp2);
  ppm_cell(qm2, qm1, q0, qp1, qp2, face_m, face_p, ql, qr);
}

// src/recon/wenoz_simple.hpp:28-81
#if defined(APK_FP_STRICT) || defined(APK_PLAIN_SQRT) || defined(APK_WENOZ_REF_FORM)
APK_DEV void wenoz(double qm2, double qm1, double q0, double qp1, double qp2, double &ql,
                   double &qr) {
  constexpr double c0 = 13. / 12., c1 = 0.25;
  const double b0 = c0 * sqr(qm2 + q0 - 2.0 * qm1) + c1 * sqr(qm2 + 3.0 * q0 - 4.0 * qm1);
  const double b1 = c0 * sqr(qm1 + qp1 - 2.0 * q0) + c1 * sqr(qm1 - qp1);
  const double b2 = c0 * sqr(qp2 + q0 - 2.0 * qp1) + c1 * sqr(qp2 + 3.0 * q0 - 4.0 * qp1);
  constexpr double eps = 1.0e-42;
  const double tau5 = fabs(b0 - b2);
#ifdef APK_WENOZ_REF_FORM
  const double p0 = b0 + eps, p1 = b1 + eps, p2 = b2 + eps;
  const double p01 = p0 * p1;
  const double tinv = tau5 * frcp48(p01 * p2);
  const double i0 = tinv * (p1 * p2);
  const double i1 = tinv * (p0 * p2);
  const double i2 = tinv * p01;
#else
  const double i0 = fdiv(tau5, (b0 + eps));
  const double i1 = fdiv(tau5, (b1 + eps));
  const double i2 = fdiv(tau5, (b2 + eps));
#endif

  double f0 = (2.0 * qm2 - 7.0 * qm1 + 11.0 * q0);
  double f1 = (-1.0 * qm1 + 5.0 * q0 + 2.0 * qp1);
  double f2 = (2.0 * q0 + 5.0 * qp1 - qp2);
  double a0 = 0.1 * (1.0 + sqr(i0));
  double a1 = 0.6 * (1.0 + sqr(i1));
  double a2 = 0.3 * (1.0 + sqr(i2));
  double asum = 6.0 * (a0 + a1 + a2);
  const double num_l = (f0 * a0 + f1 * a1 + f2 * a2), asum_l = asum;

  f0 = (2.0 * qp2 - 7.0 * qp1 + 11.0 * q0);
  f1 = (-1.0 * qp1 + 5.0 * q0 + 2.0 * qm1);
  f2 = (2.0 * q0 + 5.0 * qm1 - qm2);
  a0 = 0.1 * (1.0 + sqr(i2));
  a1 = 0.6 * (1.0 + sqr(i1));
  a2 = 0.3 * (1.0 + sqr(i0));
  asum = 6.0 * (a0 + a1 + a2);
  const double num_r = (f0 * a0 + f1 * a1 + f2 * a2);
  ql = fdiv48(num_l, asum_l);
  qr = fdiv48(num_r, asum);
}
#else
// The product build's form: the same weights up to factors that cancel in the quotients.
//  * smoothness indicators divided by 13/12 (b_k' = d2_k^2 + 3/13 d1_k^2, eps' = 12/13 eps): tau5 / (b_k + eps) unchanged;
//  * the three quotients share one reciprocal, evaluated per lane: tau5 / p_k = tau5 (p_l p_m) / (p_0 p_1 p_2) -- one
//    v_rcp_f64 instead of three (the transcendental unit issues at a quarter of the fp64 rate); p_k >= 9e-43 and
//    p_k <= ~ q^2: the product stays between 1e-126 and the square of anything a state holds;
//  * the linear weights 0.1, 0.6, 0.3 divided by 0.1, and the 1/6 of the candidate polynomials, the 6 and the 3 folded into
//    the polynomials' coefficients: q = (s_0 f_0/6 + s_1 f_1 + s_2 f_2/2) / (s_0 + 6 s_1 + 3 s_2) with s_k = 1 + i_k^2 --
//    8 multiplications fewer than alpha_k = c_k s_k, 6 sum(alpha_k);
//  * (the two weight sums keep a reciprocal each: they grow like (tau5 / eps)^2 next to a discontinuity and their
//    product can leave the range of a double.)
// ~67 instead of ~78 instructions per cell and variable; results equal to the reference's form to a few ulp
// (tests/test_gpu_parity.py compares the product build against the oracle with a tolerance, the parity build bit for bit).
APK_DEV void wenoz(double qm2, double qm1, double q0, double qp1, double qp2, double &ql,
                   double &qr) {
  constexpr double r = 3.0 / 13.0, eps = 1.0e-42 * (12.0 / 13.0);
  const double e0 = qm2 + q0 - 2.0 * qm1, g0 = qm2 + 3.0 * q0 - 4.0 * qm1;
  const double e1 = qm1 + qp1 - 2.0 * q0, g1 = qm1 - qp1;
  const double e2 = qp2 + q0 - 2.0 * qp1, g2 = qp2 + 3.0 * q0 - 4.0 * qp1;
  const double b0 = fma(e0, e0, (r * g0) * g0);
  const double b1 = fma(e1, e1, (r * g1) * g1);
  const double b2 = fma(e2, e2, (r * g2) * g2);
  const double tau5 = fabs(b0 - b2);
  const double p0 = b0 + eps, p1 = b1 + eps, p2 = b2 + eps;
  const double p01 = p0 * p1;
  const double tinv = tau5 * frcp48(p01 * p2);
  const double i0 = tinv * (p1 * p2);
  const double i1 = tinv * (p0 * p2);
  const double i2 = tinv * p01;
  const double s0 = fma(i0, i0, 1.0), s1 = fma(i1, i1, 1.0), s2 = fma(i2, i2, 1.0);
  const double s16 = 6.0 * s1;
  const double den_l = fma(3.0, s2, s16 + s0), den_r = fma(3.0, s0, s16 + s2);
  constexpr double a = 1.0 / 3.0, b = 7.0 / 6.0, c = 11.0 / 6.0;
  const double c0q = c * q0;
  const double f0l = fma(a, qm2, fma(-b, qm1, c0q)), f0r = fma(a, qp2, fma(-b, qp1, c0q));
  const double f1l = fma(2.0, qp1, fma(5.0, q0, -qm1)), f1r = fma(2.0, qm1, fma(5.0, q0, -qp1));
  const double f2l = fma(2.5, qp1, fma(-0.5, qp2, q0)), f2r = fma(2.5, qm1, fma(-0.5, qm2, q0));
  const double num_l = fma(s2, f2l, fma(s1, f1l, s0 * f0l));
  const double num_r = fma(s0, f2r, fma(s1, f1r, s2 * f0r));
  ql = fdiv48(num_l, den_l);
  qr = fdiv48(num_r, den_r);
}
#endif

// src/recon/weno3_simple.hpp:26-63
APK_DEV void weno3(double qm1, double q0, double qp1, double dx2, double &ql, double &qr) {
  const double bp = sqr(qp1 - q0);
  const double bm = sqr(q0 - qm1);
  const double tau = sqr(qp1 - 2.0 * q0 + qm1);
  const double ip = fdiv48(tau, (bp + dx2));
  const double im = fdiv48(tau, (bm + dx2));
  double f0 = q0 + qp1;
  double f1 = -qm1 + 3.0 * q0;
  double a0 = (1.0 + ip) * 2.0 / 3.0;
  double a1 = (1.0 + im) / 3.0;
  double asum = 2.0 * (a0 + a1);
  ql = fdiv48((a0 * f0 + a1 * f1), asum);
  f0 = q0 + qm1;
  f1 = -qp1 + 3.0 * q0;
  a0 = (1.0 + im) * 2.0 / 3.0;
  a1 = (1.0 + ip) / 3.0;
  asum = 2.0 * (a0 + a1);
  qr = fdiv48((a0 * f0 + a1 * f1), asum);
}

// src/hydro/diffusion/diffusion.hpp:37-47
APK_DEV double minmod(double a, double b) {
  if (a * b > 0) return (a > 0) ? min2(a, b) : max2(a, b);
  return 0.0;
}

// src/recon/limo3_simple.hpp:27-58
APK_DEV double limo3_limiter(double dvp, double dvm, double dx) {
  constexpr double r = 0.1;
  constexpr double eps = 10.0 * 2.220446049250313e-16;
  // (a plain quotient: dvp = -kTiny makes the divisor zero, where a / 0 is +-inf and the limiter's min / max chain
  // still ends in a number, but a * rcp-with-Newton(0) is NaN)
  const double theta = dvm / (dvp + kTiny);
  const double q = (2.0 + theta) / 3.0;
  const double phi =
      max2(0.0, min2(q, max2(-0.5 * theta, min2(2.0 * theta, min2(q, 1.6)))));
  double eta = r * dx;
  eta = fdiv((dvm * dvm + dvp * dvp), (eta * eta));
  if (eta <= 1.0 - eps) return q;
  if (eta >= 1.0 + eps) return phi;
  return 0.5 * ((1.0 - (eta - 1.0) / eps) * q + (1.0 + (eta - 1.0) / eps) * phi);
}

// src/recon/limo3_simple.hpp:65-78
APK_DEV void limo3(double qm1, double q0, double qp1, double dx, bool ensure_positivity,
                   double &ql, double &qr) {
  const double dqp = qp1 - q0;
  const double dqm = q0 - qm1;
  double l = q0 + 0.5 * dqp * limo3_limiter(dqp, dqm, dx);
  double r = q0 - 0.5 * dqm * limo3_limiter(dqm, dqp, dx);
  if (ensure_positivity && (l <= 0.0 || r <= 0.0)) {
    const double dmm = minmod(dqp, dqm);
    l = q0 + 0.5 * dmm;
    r = q0 - 0.5 * dmm;
  }
  ql = l;
  qr = r;
}

// Generic entry (the Reconstruct<recon,DIR> wrappers).  `var` is the variable index:
// LimO3 falls back to minmod for density and pressure only (limo3_simple.hpp:98).
template <int RECON>
APK_DEV void reconstruct(double qm2, double qm1, double q0, double qp1, double qp2, double dx,
                         int var, double &ql, double &qr) {
  if constexpr (RECON == APK_RC_DC) {
    ql = q0;
    qr = q0;
  } else if constexpr (RECON == APK_RC_PLM) {
    plm(qm1, q0, qp1, ql, qr);
  } else if constexpr (RECON == APK_RC_PPM) {
    ppm(qm2, qm1, q0, qp1, qp2, ql, qr);
  } else if constexpr (RECON == APK_RC_WENOZ) {
    wenoz(qm2, qm1, q0, qp1, qp2, ql, qr);
  } else if constexpr (RECON == APK_RC_WENO3) {
    double dx2 = dx;
    dx2 = dx2 * dx2;
    weno3(qm1, q0, qp1, dx2, ql, qr);
  } else {
    limo3(qm1, q0, qp1, dx, (var == IDN || var == IPR), ql, qr);
  }
}

// ======================================================================================
// EOS helpers
// ======================================================================================
// src/eos/adiabatic_hydro.hpp:43-45
#if defined(APK_FP_STRICT) || defined(APK_PLAIN_SQRT) || defined(APK_SOUND_SPEED_DIVIDE)
APK_DEV double sound_speed(double gamma, double d, double p) { return fsqrt(gamma * p / d); }
#else
// Product build: sqrt(x / d) = x / sqrt(x d) -- one v_rsq_f64 with its Newton steps instead of a quotient (v_rcp_f64, two
// steps, a residual correction) AND a root; within 2 ulp of the reference's form.  (Hydro only calls this; the fast speed
// of GLM-MHD keeps the reference's form in both builds: see fast_speed.)
APK_DEV double sound_speed(double gamma, double d, double p) {
  const double x = gamma * p, z = x * d;
  double root, inv_root;
  fsqrt_rsqrt(z, root, inv_root);
  return (z == 0.0) ? 0.0 : x * inv_root;  // rsq(0) = inf
}
#endif
// src/eos/adiabatic_glmmhd.hpp:47-54
APK_DEV double fast_speed(double gamma, double d, double p, double bx, double by, double bz) {
  const double asq = gamma * p;
  const double ct2 = by * by + bz * bz;
  const double qsq = bx * bx + ct2 + asq;
  const double tmp = bx * bx + ct2 - asq;
  // (Kept in the reference's form in both builds, each operation rounded like IEEE's: the HLLD star
  // states divide by rho (s - v)(s - s*) - Bx^2, which vanishes where the fast speed equals the Alfven
  // speed (By = Bz = 0), and next to such faces a last-bit difference in c_f is amplified up to the
  // 1e-8 of the solver's degeneracy threshold.  sqrt(x / d) = x rsqrt(x d) saves a divide per call
  // and agrees to 6e-16, but moved the 256^3 benchmark state by 1e-10 per cycle against the parity
  // build along the planes where By and Bz change sign.)
  // (the inner argument is a sum of squares -- zero only where Bx^2 = gamma p and By = Bz = 0 exactly --, the outer
  // one a squared speed, positive whenever rho and p are: the product build drops fsqrt's zero guards, see fsqrt_pos)
  return fsqrt_pos(0.5 * (qsq + fsqrt_nonneg(tmp * tmp + 4.0 * asq * ct2)) / d);
}

// ======================================================================================
// Riemann solvers.  States are direction-permuted: w[IV1] is the normal velocity,
// w[IB1] the normal field.  f[] is returned in the same permuted order.
// ======================================================================================
struct Cons1D {  // glmmhd_hlld.hpp:32-34
  double d, mx, my, mz, e, by, bz;
};

// src/hydro/rsolvers/hydro_hlle.hpp:40-138
APK_DEV void hydro_hlle(const double (&wl)[NHYDRO], const double (&wr)[NHYDRO], double gamma,
                        double (&f)[NHYDRO]) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0 / gm1;
  const double sdl = fsqrt(wl[IDN]);
  const double sdr = fsqrt(wr[IDN]);
  const double isum = 1.0 / (sdl + sdr);
  const double roe_v1 = (sdl * wl[IV1] + sdr * wr[IV1]) * isum;
  const double roe_v2 = (sdl * wl[IV2] + sdr * wr[IV2]) * isum;
  const double roe_v3 = (sdl * wl[IV3] + sdr * wr[IV3]) * isum;
  const double el = wl[IPR] * igm1 + 0.5 * wl[IDN] * (sqr(wl[IV1]) + sqr(wl[IV2]) + sqr(wl[IV3]));
  const double er = wr[IPR] * igm1 + 0.5 * wr[IDN] * (sqr(wr[IV1]) + sqr(wr[IV2]) + sqr(wr[IV3]));
  const double hroe = ((el + wl[IPR]) / sdl + (er + wr[IPR]) / sdr) * isum;
  const double cl = sound_speed(gamma, wl[IDN], wl[IPR]);
  const double cr = sound_speed(gamma, wr[IDN], wr[IPR]);
  const double q = hroe - 0.5 * (sqr(roe_v1) + sqr(roe_v2) + sqr(roe_v3));
  const double a = (q < 0.0) ? 0.0 : fsqrt(gm1 * q);
  const double al = min2((roe_v1 - a), (wl[IV1] - cl));
  const double ar = max2((roe_v1 + a), (wr[IV1] + cr));
  const double bp = ar > 0.0 ? ar : kTiny;
  const double bm = al < 0.0 ? al : kTiny;  // hydro HLLE: +TINY (:97-98)
  const double vxl = wl[IV1] - bm;
  const double vxr = wr[IV1] - bp;
  double fl[NHYDRO], fr[NHYDRO];
  fl[IDN] = wl[IDN] * vxl;
  fr[IDN] = wr[IDN] * vxr;
  fl[IV1] = wl[IDN] * wl[IV1] * vxl;
  fr[IV1] = wr[IDN] * wr[IV1] * vxr;
  fl[IV2] = wl[IDN] * wl[IV2] * vxl;
  fr[IV2] = wr[IDN] * wr[IV2] * vxr;
  fl[IV3] = wl[IDN] * wl[IV3] * vxl;
  fr[IV3] = wr[IDN] * wr[IV3] * vxr;
  fl[IV1] += wl[IPR];
  fr[IV1] += wr[IPR];
  fl[IEN] = el * vxl + wl[IPR] * wl[IV1];
  fr[IEN] = er * vxr + wr[IPR] * wr[IV1];
  double tmp = 0.0;
  if (bp != bm) tmp = 0.5 * (bp + bm) / (bp - bm);
#pragma unroll
  for (int n = 0; n < NHYDRO; ++n) f[n] = 0.5 * (fl[n] + fr[n]) + (fl[n] - fr[n]) * tmp;
}

// src/hydro/rsolvers/hydro_hllc.hpp:32-157
APK_DEV void hydro_hllc(const double (&wl)[NHYDRO], const double (&wr)[NHYDRO], double gamma,
                        double (&f)[NHYDRO]) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0 / gm1;
  const double cl = sound_speed(gamma, wl[IDN], wl[IPR]);
  const double cr = sound_speed(gamma, wr[IDN], wr[IPR]);
  const double el = wl[IPR] * igm1 + 0.5 * wl[IDN] * (sqr(wl[IV1]) + sqr(wl[IV2]) + sqr(wl[IV3]));
  const double er = wr[IPR] * igm1 + 0.5 * wr[IDN] * (sqr(wr[IV1]) + sqr(wr[IV2]) + sqr(wr[IV3]));
  const double rhoa = .5 * (wl[IDN] + wr[IDN]);
  const double ca = .5 * (cl + cr);
  const double pmid = .5 * (wl[IPR] + wr[IPR] + (wl[IV1] - wr[IV1]) * rhoa * ca);
  const double ql = (pmid <= wl[IPR])
                        ? 1.0
                        : fsqrt(1.0 + (gamma + 1) / (2 * gamma) * (pmid / wl[IPR] - 1.0));
  const double qr = (pmid <= wr[IPR])
                        ? 1.0
                        : fsqrt(1.0 + (gamma + 1) / (2 * gamma) * (pmid / wr[IPR] - 1.0));
  const double al = wl[IV1] - cl * ql;
  const double ar = wr[IV1] + cr * qr;
  const double bp = ar > 0.0 ? ar : (kTiny);
  const double bm = al < 0.0 ? al : -(kTiny);
  double vxl = wl[IV1] - al;
  double vxr = wr[IV1] - ar;
  const double tl = wl[IPR] + vxl * wl[IDN] * wl[IV1];
  const double tr = wr[IPR] + vxr * wr[IDN] * wr[IV1];
  const double ml = wl[IDN] * vxl;
  const double mr = -(wr[IDN] * vxr);
#if defined(APK_FP_STRICT) || defined(APK_PLAIN_SQRT) || defined(APK_HLLC_REF_FORM)
  const double am = (tl - tr) / (ml + mr);
  double cp = (ml * tr + mr * tl) / (ml + mr);
#else
  // (product build: the two quotients share their reciprocal, and so do the two of the branch below -- the denominator is
  // picked first; three v_rcp_f64 sequences a face fewer, results within 2 ulp of the reference's form)
  const double inv_m = frcp(ml + mr);
  const double am = (tl - tr) * inv_m;
  double cp = (ml * tr + mr * tl) * inv_m;
#endif
  cp = cp > 0.0 ? cp : 0.0;
  vxl = wl[IV1] - bm;
  vxr = wr[IV1] - bp;
  double fl[NHYDRO], fr[NHYDRO];
  fl[IDN] = wl[IDN] * vxl;
  fr[IDN] = wr[IDN] * vxr;
  fl[IV1] = wl[IDN] * wl[IV1] * vxl + wl[IPR];
  fr[IV1] = wr[IDN] * wr[IV1] * vxr + wr[IPR];
  fl[IV2] = wl[IDN] * wl[IV2] * vxl;
  fr[IV2] = wr[IDN] * wr[IV2] * vxr;
  fl[IV3] = wl[IDN] * wl[IV3] * vxl;
  fr[IV3] = wr[IDN] * wr[IV3] * vxr;
  fl[IEN] = el * vxl + wl[IPR] * wl[IV1];
  fr[IEN] = er * vxr + wr[IPR] * wr[IV1];
  double sl, sr, sm;
#if defined(APK_FP_STRICT) || defined(APK_PLAIN_SQRT) || defined(APK_HLLC_REF_FORM)
  if (am >= 0.0) {
    sl = am / (am - bm);
    sr = 0.0;
    sm = -bm / (am - bm);
  } else {
    sl = 0.0;
    sr = -am / (bp - am);
    sm = bp / (bp - am);
  }
#else
  {
    const bool pos = am >= 0.0;
    const double inv = frcp(pos ? (am - bm) : (bp - am));
    sl = pos ? am * inv : 0.0;
    sr = pos ? 0.0 : -am * inv;
    sm = pos ? -bm * inv : bp * inv;
  }
#endif
  f[IDN] = sl * fl[IDN] + sr * fr[IDN];
  f[IV1] = sl * fl[IV1] + sr * fr[IV1] + sm * cp;
  f[IV2] = sl * fl[IV2] + sr * fr[IV2];
  f[IV3] = sl * fl[IV3] + sr * fr[IV3];
  f[IEN] = sl * fl[IEN] + sr * fr[IEN] + sm * cp * am;
}

// src/hydro/rsolvers/hydro_dc_llf.hpp:43-142 (wl/wr are the first-order states)
APK_DEV void hydro_llf(const double (&wl)[NHYDRO], const double (&wr)[NHYDRO], double gamma,
                       double (&f)[NHYDRO]) {
  const double igm1 = 1.0 / (gamma - 1.0);
  double qa = wl[IDN] * wl[IV1];
  double qb = wr[IDN] * wr[IV1];
  const double fs_d = qa + qb;
  double fs_mx = qa * wl[IV1] + qb * wr[IV1];
  const double fs_my = qa * wl[IV2] + qb * wr[IV2];
  const double fs_mz = qa * wl[IV3] + qb * wr[IV3];
  const double el = wl[IPR] * igm1 + 0.5 * wl[IDN] * (sqr(wl[IV1]) + sqr(wl[IV2]) + sqr(wl[IV3]));
  const double er = wr[IPR] * igm1 + 0.5 * wr[IDN] * (sqr(wr[IV1]) + sqr(wr[IV2]) + sqr(wr[IV3]));
  fs_mx += (wl[IPR] + wr[IPR]);
  const double fs_e = (el + wl[IPR]) * wl[IV1] + (er + wr[IPR]) * wr[IV1];
  qa = sound_speed(gamma, wl[IDN], wl[IPR]);
  qb = sound_speed(gamma, wr[IDN], wr[IPR]);
  const double a = fmax((fabs(wl[IV1]) + qa), (fabs(wr[IV1]) + qb));
  const double du_d = a * (wr[IDN] - wl[IDN]);
  const double du_mx = a * (wr[IDN] * wr[IV1] - wl[IDN] * wl[IV1]);
  const double du_my = a * (wr[IDN] * wr[IV2] - wl[IDN] * wl[IV2]);
  const double du_mz = a * (wr[IDN] * wr[IV3] - wl[IDN] * wl[IV3]);
  const double du_e = a * (er - el);
  f[IDN] = 0.5 * (fs_d - du_d);
  f[IV1] = 0.5 * (fs_mx - du_mx);
  f[IV2] = 0.5 * (fs_my - du_my);
  f[IV3] = 0.5 * (fs_mz - du_mz);
  f[IEN] = 0.5 * (fs_e - du_e);
}

// GLM 2x2 interface state shared by the MHD solvers (glmmhd_hlld.hpp:88-92)
APK_DEV void glm_interface(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD],
                           double c_h, double &bxi, double &psii) {
  bxi = 0.5 * (wl[IB1] + wr[IB1]) - 0.5 / c_h * (wr[IPS] - wl[IPS]);
  psii = 0.5 * (wl[IPS] + wr[IPS]) - 0.5 * c_h * (wr[IB1] - wl[IB1]);
}

// Wave-uniform constants of a fused stage that the pointwise functions derive from gamma, c_h and the EOS limits.
// There is no scalar fp64 unit: left to the kernels, 1 / (gamma - 1), 0.5 / c_h, c_h^2, vceil^2 ... are computed by
// the vector ALU in the prologue and then HELD IN VECTOR REGISTERS across the marches -- seven values = 14 VGPRs of
// kernels that sit at the 256-register limit (the finishing march spilled two of them to scratch and re-read them
// three times per iteration, each time behind an s_waitcnt vmcnt(0) that also waits for the iteration's stores).
// The host evaluates the same IEEE expressions once per launch (make_stage_consts, fused_dispatch.hip); as kernel
// arguments they live in SGPRs and fp64 instructions take them as scalar operands.  Bit-identical in the parity
// build (same correctly rounded operations); the product build's in-kernel reciprocals were within an ulp of these.
struct StageConsts {
  double gamma, c_h;
  double gm1, igm1;                  // gamma - 1, 1 / (gamma - 1)
  double half_over_ch, half_ch, ch_sq;  // 0.5 / c_h, 0.5 * c_h, c_h^2 (glm_interface, the psi flux)
  double eos_gm1, vceil_sq, pfloor_over_gm1;  // ConsToPrim: eos.gamma - 1, eos.vceil^2, eos.pfloor / (eos.gamma - 1)
};
inline __host__ __device__ StageConsts make_stage_consts(double gamma, double c_h, const apk_eos &eos) {
  StageConsts k;
  k.gamma = gamma;
  k.c_h = c_h;
  k.gm1 = gamma - 1.0;
  k.igm1 = 1.0 / k.gm1;
  k.half_over_ch = 0.5 / c_h;
  k.half_ch = 0.5 * c_h;
  k.ch_sq = c_h * c_h;
  k.eos_gm1 = eos.gamma - 1.0;
  k.vceil_sq = eos.vceil * eos.vceil;
  k.pfloor_over_gm1 = eos.pfloor / k.eos_gm1;
  return k;
}
APK_DEV void glm_interface(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD], const StageConsts &k,
                           double &bxi, double &psii) {
  bxi = 0.5 * (wl[IB1] + wr[IB1]) - k.half_over_ch * (wr[IPS] - wl[IPS]);
  psii = 0.5 * (wl[IPS] + wr[IPS]) - k.half_ch * (wr[IB1] - wl[IB1]);
}

// src/hydro/rsolvers/glmmhd_hlle.hpp:27-192
APK_DEV void glmmhd_hlle(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD],
                         double gamma, double c_h, double (&f)[NGLMMHD]) {
  const double gm1 = gamma - 1.0;
  double bxi, psii;
  glm_interface(wl, wr, c_h, bxi, psii);
  f[IB1] = psii;
  f[IPS] = sqr(c_h) * bxi;

  const double sdl = fsqrt(wl[IDN]);
  const double sdr = fsqrt(wr[IDN]);
  const double isum = 1.0 / (sdl + sdr);
  const double roe_d = sdl * sdr;
  const double roe_v1 = (sdl * wl[IV1] + sdr * wr[IV1]) * isum;
  const double roe_v2 = (sdl * wl[IV2] + sdr * wr[IV2]) * isum;
  const double roe_v3 = (sdl * wl[IV3] + sdr * wr[IV3]) * isum;
  const double roe_b2 = (sdr * wl[IB2] + sdl * wr[IB2]) * isum;
  const double roe_b3 = (sdr * wl[IB3] + sdl * wr[IB3]) * isum;
  const double x = 0.5 * (sqr(wl[IB2] - wr[IB2]) + sqr(wl[IB3] - wr[IB3])) / (sqr(sdl + sdr));
  const double y = 0.5 * (wl[IDN] + wr[IDN]) / roe_d;
  const double pbl = 0.5 * (bxi * bxi + sqr(wl[IB2]) + sqr(wl[IB3]));
  const double pbr = 0.5 * (bxi * bxi + sqr(wr[IB2]) + sqr(wr[IB3]));
  const double el =
      wl[IPR] / gm1 + 0.5 * wl[IDN] * (sqr(wl[IV1]) + sqr(wl[IV2]) + sqr(wl[IV3])) + pbl;
  const double er =
      wr[IPR] / gm1 + 0.5 * wr[IDN] * (sqr(wr[IV1]) + sqr(wr[IV2]) + sqr(wr[IV3])) + pbr;
  const double hroe = ((el + wl[IPR] + pbl) / sdl + (er + wr[IPR] + pbr) / sdr) * isum;
  const double cl = fast_speed(gamma, wl[IDN], wl[IPR], wl[IB1], wl[IB2], wl[IB3]);
  const double cr = fast_speed(gamma, wr[IDN], wr[IPR], wr[IB1], wr[IB2], wr[IB3]);
  const double btsq = sqr(roe_b2) + sqr(roe_b3);
  const double vaxsq = bxi * bxi / roe_d;
  const double bt_starsq = (gm1 - (gm1 - 1.0) * y) * btsq;
  const double hp = hroe - (vaxsq + btsq / roe_d);
  const double vsq = sqr(roe_v1) + sqr(roe_v2) + sqr(roe_v3);
  const double twid_asq = max2((gm1 * (hp - 0.5 * vsq) - (gm1 - 1.0) * x), 0.0);
  const double ct2 = bt_starsq / roe_d;
  const double tsum = vaxsq + ct2 + twid_asq;
  const double tdif = vaxsq + ct2 - twid_asq;
  const double cf2_cs2 = fsqrt(tdif * tdif + 4.0 * twid_asq * ct2);
  const double cfsq = 0.5 * (tsum + cf2_cs2);
  const double a = fsqrt(cfsq);
  const double al = min2((roe_v1 - a), (wl[IV1] - cl));
  const double ar = max2((roe_v1 + a), (wr[IV1] + cr));
  const double bp = ar > 0.0 ? ar : 0.0;  // MHD HLLE: 0.0 (:134-135)
  const double bm = al < 0.0 ? al : 0.0;
  const double vxl = wl[IV1] - bm;
  const double vxr = wr[IV1] - bp;
  double fl[NGLMMHD], fr[NGLMMHD];
  fl[IDN] = wl[IDN] * vxl;
  fr[IDN] = wr[IDN] * vxr;
  fl[IV1] = wl[IDN] * wl[IV1] * vxl + pbl - sqr(bxi);
  fr[IV1] = wr[IDN] * wr[IV1] * vxr + pbr - sqr(bxi);
  fl[IV2] = wl[IDN] * wl[IV2] * vxl - bxi * wl[IB2];
  fr[IV2] = wr[IDN] * wr[IV2] * vxr - bxi * wr[IB2];
  fl[IV3] = wl[IDN] * wl[IV3] * vxl - bxi * wl[IB3];
  fr[IV3] = wr[IDN] * wr[IV3] * vxr - bxi * wr[IB3];
  fl[IV1] += wl[IPR];
  fr[IV1] += wr[IPR];
  fl[IEN] = el * vxl + wl[IV1] * (wl[IPR] + pbl - bxi * bxi);
  fr[IEN] = er * vxr + wr[IV1] * (wr[IPR] + pbr - bxi * bxi);
  fl[IEN] -= bxi * (wl[IB2] * wl[IV2] + wl[IB3] * wl[IV3]);
  fr[IEN] -= bxi * (wr[IB2] * wr[IV2] + wr[IB3] * wr[IV3]);
  fl[IB2] = wl[IB2] * vxl - bxi * wl[IV2];
  fr[IB2] = wr[IB2] * vxr - bxi * wr[IV2];
  fl[IB3] = wl[IB3] * vxl - bxi * wl[IV3];
  fr[IB3] = wr[IB3] * vxr - bxi * wr[IV3];
  double tmp = 0.0;
  if (bp != bm) tmp = 0.5 * (bp + bm) / (bp - bm);
#define APK_HLLE_MIX(n) f[n] = 0.5 * (fl[n] + fr[n]) + (fl[n] - fr[n]) * tmp
  APK_HLLE_MIX(IDN);
  APK_HLLE_MIX(IV1);
  APK_HLLE_MIX(IV2);
  APK_HLLE_MIX(IV3);
  APK_HLLE_MIX(IEN);
  APK_HLLE_MIX(IB2);
  APK_HLLE_MIX(IB3);
#undef APK_HLLE_MIX
}

// ---- HLLD (src/hydro/rsolvers/glmmhd_hlld.hpp:39-396) -----------------------------------
// conserved 1-D state and magnetic pressure of one side (:95-118)
APK_DEV void hlld_side_state(const double (&w)[NGLMMHD], double igm1, double bxsq, Cons1D &u,
                             double &pb) {
  pb = 0.5 * (bxsq + (sqr(w[IB2]) + sqr(w[IB3])));
  const double ke = 0.5 * w[IDN] * (sqr(w[IV1]) + (sqr(w[IV2]) + sqr(w[IV3])));
  u.d = w[IDN];
  u.mx = w[IV1] * u.d;
  u.my = w[IV2] * u.d;
  u.mz = w[IV3] * u.d;
  u.e = w[IPR] * igm1 + ke + pb;
  u.by = w[IB2];
  u.bz = w[IB3];
}
// physical flux of one side (:141-155)
APK_DEV void hlld_side_flux(const double (&w)[NGLMMHD], const Cons1D &u, double pt, double bxi,
                            double bxsq, Cons1D &fx) {
  fx.d = u.mx;
  fx.mx = u.mx * w[IV1] + pt - bxsq;
  fx.my = u.my * w[IV1] - bxi * u.by;
  fx.mz = u.mz * w[IV1] - bxi * u.bz;
  fx.e = w[IV1] * (u.e + pt - bxsq) - bxi * (w[IV2] * u.by + w[IV3] * u.bz);
  fx.by = u.by * w[IV1] - bxi * w[IV2];
  fx.bz = u.bz * w[IV1] - bxi * w[IV3];
}
// true if the predicate holds in any active lane of the wave (scalar branch condition)
APK_DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// transverse components of the star state of one side, eqns (44)-(47) of M&K (:193-211, :225-243);
// ust.d is set by the caller
APK_DEV void hlld_star_transverse(const double (&w)[NGLMMHD], const Cons1D &u, double sd, double sdm,
                                  double ptst, double bxi, double bxsq, Cons1D &ust) {
  const double denom = u.d * sd * sdm - bxsq;
#ifdef APK_PLAIN_SQRT
  const double t1 = bxi * (sd - sdm) / denom;
  const double t2 = (u.d * sqr(sd) - bxsq) / denom;
#else
  const double inv = frcp(denom);
  const double t1 = bxi * (sd - sdm) * inv;
  const double t2 = (u.d * sqr(sd) - bxsq) * inv;
#endif
  const bool degenerate = fabs(denom) < (kHlldSmall)*ptst;
#ifdef APK_FP_STRICT
  ust.my = degenerate ? ust.d * w[IV2] : ust.d * (w[IV2] - u.by * t1);
  ust.mz = degenerate ? ust.d * w[IV3] : ust.d * (w[IV3] - u.bz * t1);
  ust.by = degenerate ? u.by : u.by * t2;
  ust.bz = degenerate ? u.bz : u.bz * t2;
#else
  // the degenerate case selected on the two factors instead of on the four results (w - B * 0 = w and B * 1 = B to
  // the bit for finite B): 4 selects instead of 8
  const double t1s = degenerate ? 0.0 : t1, t2s = degenerate ? 1.0 : t2;
  ust.my = ust.d * (w[IV2] - u.by * t1s);
  ust.mz = ust.d * (w[IV3] - u.bz * t1s);
  ust.by = u.by * t2s;
  ust.bz = u.bz * t2s;
#endif
}
// a <- s * (a - b)  (:297-327)
APK_DEV void hlld_jump(Cons1D &a, const Cons1D &b, double s) {
  a.d = s * (a.d - b.d);
  a.mx = s * (a.mx - b.mx);
  a.my = s * (a.my - b.my);
  a.mz = s * (a.mz - b.mz);
  a.e = s * (a.e - b.e);
  a.by = s * (a.by - b.by);
  a.bz = s * (a.bz - b.bz);
}
APK_DEV double pick(bool left, double l, double r) { return left ? l : r; }

// HLLD (glmmhd_hlld.hpp:39-396).  The reference evaluates both star energies, both physical
// fluxes and all four wave jumps for every face and then picks one of six sums.  A lane's flux
// is built from ONE side of the contact only (left: F_L [+ jump over s0 [+ jump over s1]],
// right: F_R [+ jump over s4 [+ jump over s3]]), so the side is chosen first -- from the same
// comparisons in the same order -- the operands of that side are selected per lane, and the
// side-specific part (physical flux, star energy, two jumps) is evaluated once.  What both sides
// need (wave speeds, transverse star components: the double-star states mix them) is computed for
// both.  Every value that reaches the result comes from the reference's expressions in the
// reference's order: bit-identical in the parity build.
// glmmhd_hlld with the fast speeds of the two states handed in (cfl, cfr = fast_speed of wl / wr along the
// face normal): a donor-cell sweep solves two faces per cell with the SAME cell-centre state on one side, so
// the caller evaluates that state's fast speed once (same function of the same values: bit-identical).
APK_DEV void glmmhd_hlld_cf(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD], double gamma, double c_h,
                            double cfl, double cfr, double (&f)[NGLMMHD]);

APK_DEV void glmmhd_hlld(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD],
                         double gamma, double c_h, double (&f)[NGLMMHD]) {
  // fast speeds from the RECONSTRUCTED normal field, not bxi (:122-125)
  const double cfl = fast_speed(gamma, wl[IDN], wl[IPR], wl[IB1], wl[IB2], wl[IB3]);
  const double cfr = fast_speed(gamma, wr[IDN], wr[IPR], wr[IB1], wr[IB2], wr[IB3]);
  glmmhd_hlld_cf(wl, wr, gamma, c_h, cfl, cfr, f);
}

// (the solver proper: igm1 = 1 / (gamma - 1), the GLM interface state and c_h^2 handed in.  MASKED_PICK: the operands
// of the lane's side as the right side's values overwritten under the execution mask of the lanes on the left -- one
// v_mov_b64 per value -- instead of two v_cndmask_b32 each: the marches, -80 + 44 vector instructions per iteration of the
// finishing march, its time -1.8 %, the x3 sweep -1.4 %, WENOZ RK3 cycle -1.2 %, same box; the donor-cell predictor
// with its 3.5 interleaved solves per cell is 1.8 % SLOWER with it and keeps the selects)
template <bool MASKED_PICK>
APK_DEV void glmmhd_hlld_core(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD], double igm1, double bxi,
                              double psii, double ch_sq, double cfl, double cfr, double (&f)[NGLMMHD]);

APK_DEV void glmmhd_hlld_cf(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD], double gamma, double c_h,
                            double cfl, double cfr, double (&f)[NGLMMHD]) {
  const double gm1 = gamma - 1.0;
  const double igm1 = 1.0 / gm1;
  double bxi, psii;
  glm_interface(wl, wr, c_h, bxi, psii);
  glmmhd_hlld_core<false>(wl, wr, igm1, bxi, psii, sqr(c_h), cfl, cfr, f);
}
// the same with the stage's constants from the host (StageConsts)
APK_DEV void glmmhd_hlld_cf(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD], const StageConsts &k,
                            double cfl, double cfr, double (&f)[NGLMMHD]) {
  double bxi, psii;
  glm_interface(wl, wr, k, bxi, psii);
  glmmhd_hlld_core<false>(wl, wr, k.igm1, bxi, psii, k.ch_sq, cfl, cfr, f);
}
// (the marches' entry: one face per call)
APK_DEV void glmmhd_hlld(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD], const StageConsts &k,
                         double (&f)[NGLMMHD]) {
  const double cfl = fast_speed(k.gamma, wl[IDN], wl[IPR], wl[IB1], wl[IB2], wl[IB3]);
  const double cfr = fast_speed(k.gamma, wr[IDN], wr[IPR], wr[IB1], wr[IB2], wr[IB3]);
  double bxi, psii;
  glm_interface(wl, wr, k, bxi, psii);
  glmmhd_hlld_core<true>(wl, wr, k.igm1, bxi, psii, k.ch_sq, cfl, cfr, f);
}

template <bool MASKED_PICK>
APK_DEV void glmmhd_hlld_core(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD], double igm1, double bxi,
                              double psii, double ch_sq, double cfl, double cfr, double (&f)[NGLMMHD]) {
  f[IB1] = psii;
  f[IPS] = ch_sq * bxi;
  const double bxsq = bxi * bxi;

  Cons1D ul, ur;
  double pbl, pbr;
  hlld_side_state(wl, igm1, bxsq, ul, pbl);
  hlld_side_state(wr, igm1, bxsq, ur, pbr);

  const double s0 = min2(wl[IV1] - cfl, wr[IV1] - cfr);
  const double s4 = max2(wl[IV1] + cfl, wr[IV1] + cfr);

  const double ptl = wl[IPR] + pbl;
  const double ptr = wr[IPR] + pbr;

  const double sdl = s0 - wl[IV1];
  const double sdr = s4 - wr[IV1];
  const double s2 = (sdr * ur.mx - sdl * ul.mx + (ptl - ptr)) / (sdr * ur.d - sdl * ul.d);
  const double sdml = s0 - s2;
  const double sdmr = s4 - s2;
  Cons1D ulst, urst;
#ifdef APK_PLAIN_SQRT
  const double sdml_inv = 1.0 / sdml;
  const double sdmr_inv = 1.0 / sdmr;
  ulst.d = ul.d * sdl * sdml_inv;
  urst.d = ur.d * sdr * sdmr_inv;
  const double ulst_d_inv = 1.0 / ulst.d;
  const double urst_d_inv = 1.0 / urst.d;
  const double sqrtdl = sqrt(ulst.d);
  const double sqrtdr = sqrt(urst.d);
  const double s1 = s2 - fabs(bxi) / sqrtdl;
  const double s3 = s2 + fabs(bxi) / sqrtdr;
#else
  // one reciprocal for both 1/(s0 - s2) and 1/(s4 - s2); sqrt(d*), 1/sqrt(d*) and 1/d* from one rsq
  const double inv_prod = frcp(sdml * sdmr);
  const double sdml_inv = sdmr * inv_prod;
  const double sdmr_inv = sdml * inv_prod;
  ulst.d = ul.d * sdl * sdml_inv;
  urst.d = ur.d * sdr * sdmr_inv;
  double sqrtdl, sqrtdr, rsl, rsr;
  fsqrt_rsqrt(ulst.d, sqrtdl, rsl);
  fsqrt_rsqrt(urst.d, sqrtdr, rsr);
  const double ulst_d_inv = rsl * rsl;
  const double urst_d_inv = rsr * rsr;
  const double s1 = s2 - fabs(bxi) * rsl;
  const double s3 = s2 + fabs(bxi) * rsr;
#endif

  const double ptstl = ptl + ul.d * sdl * (s2 - wl[IV1]);
  const double ptstr = ptr + ur.d * sdr * (s2 - wr[IV1]);
  const double ptst = 0.5 * (ptstr + ptstl);

  // which of the six sums the lane returns (:330-387), decided before anything side-specific:
  //   s0 >= 0: F_L | s4 <= 0: F_R | s1 >= 0: F_L + j0 | s2 >= 0: F_L + j0 + j1 | s3 > 0: F_R + j4 + j3 | F_R + j4
  const bool c0 = s0 >= 0.0, c4 = s4 <= 0.0, c1 = s1 >= 0.0, c2 = s2 >= 0.0, c3 = s3 > 0.0;
  const bool fan = !c0 && !c4;
  const bool left = c0 || (fan && (c1 || c2));
  const bool outer = c0 || c4;                        // physical flux alone
  const bool with_dstar = fan && !c1 && (c2 || c3);   // the jump across the Alfven wave takes part

  hlld_star_transverse(wl, ul, sdl, sdml, ptst, bxi, bxsq, ulst);
  hlld_star_transverse(wr, ur, sdr, sdmr, ptst, bxi, bxsq, urst);

  // ---- operands of the lane's side
  double w1, w2, w3, pt, sd, sdm_inv, ust_d_inv, s_outer, s_inner;
  Cons1D u, ust;
  if constexpr (MASKED_PICK) {
    // (the empty asm keeps the compiler from turning the block back into selects)
    w1 = wr[IV1], w2 = wr[IV2], w3 = wr[IV3];
    u = ur;
    ust.d = urst.d, ust.my = urst.my, ust.mz = urst.mz, ust.by = urst.by, ust.bz = urst.bz;
    pt = ptr, sd = sdr, sdm_inv = sdmr_inv, ust_d_inv = urst_d_inv, s_outer = s4, s_inner = s3;
    if (left) {
      asm volatile("" ::: );
      w1 = wl[IV1], w2 = wl[IV2], w3 = wl[IV3];
      u.d = ul.d, u.mx = ul.mx, u.my = ul.my, u.mz = ul.mz, u.e = ul.e, u.by = ul.by, u.bz = ul.bz;
      ust.d = ulst.d, ust.my = ulst.my, ust.mz = ulst.mz, ust.by = ulst.by, ust.bz = ulst.bz;
      pt = ptl, sd = sdl, sdm_inv = sdml_inv, ust_d_inv = ulst_d_inv, s_outer = s0, s_inner = s1;
    }
  } else {
    w1 = pick(left, wl[IV1], wr[IV1]);
    w2 = pick(left, wl[IV2], wr[IV2]);
    w3 = pick(left, wl[IV3], wr[IV3]);
    u.d = pick(left, ul.d, ur.d);
    u.mx = pick(left, ul.mx, ur.mx);
    u.my = pick(left, ul.my, ur.my);
    u.mz = pick(left, ul.mz, ur.mz);
    u.e = pick(left, ul.e, ur.e);
    u.by = pick(left, ul.by, ur.by);
    u.bz = pick(left, ul.bz, ur.bz);
    ust.d = pick(left, ulst.d, urst.d);
    ust.my = pick(left, ulst.my, urst.my);
    ust.mz = pick(left, ulst.mz, urst.mz);
    ust.by = pick(left, ulst.by, urst.by);
    ust.bz = pick(left, ulst.bz, urst.bz);
    pt = pick(left, ptl, ptr);
    sd = pick(left, sdl, sdr);
    sdm_inv = pick(left, sdml_inv, sdmr_inv);
    ust_d_inv = pick(left, ulst_d_inv, urst_d_inv);
    s_outer = pick(left, s0, s4);
    s_inner = pick(left, s1, s3);
  }

  // physical flux of that side (:141-155)
  Cons1D fx;
  fx.d = u.mx;
  fx.mx = u.mx * w1 + pt - bxsq;
  fx.my = u.my * w1 - bxi * u.by;
  fx.mz = u.mz * w1 - bxi * u.bz;
  fx.e = w1 * (u.e + pt - bxsq) - bxi * (w2 * u.by + w3 * u.bz);
  fx.by = u.by * w1 - bxi * w2;
  fx.bz = u.bz * w1 - bxi * w3;

  // star state of that side: eqns (39), (48) (:192, :212-219)
  ust.mx = ust.d * s2;
  const double vbst = (ust.mx * bxi + (ust.my * ust.by + ust.mz * ust.bz)) * ust_d_inv;
  ust.e = (sd * u.e - pt * w1 + ptst * s2 + bxi * (w1 * bxi + (w2 * u.by + w3 * u.bz) - vbst)) * sdm_inv;

  // double-star state of that side (:252-294); same as the star state where Bx is near zero.
  // Only lanes between the two Alfven waves use it: waves without such a lane skip the block.
  Cons1D udst = ust;
  if (wave_any(with_dstar)) {
    const bool dst_degenerate = 0.5 * bxsq < (kHlldSmall)*ptst;
#ifdef APK_PLAIN_SQRT
    const double invsumd = 1.0 / (sqrtdl + sqrtdr);
#else
    const double invsumd = frcp(sqrtdl + sqrtdr);
#endif
    const double bxsig = (bxi > 0.0 ? 1.0 : -1.0);
    const double tmy = invsumd * (sqrtdl * (ulst.my * ulst_d_inv) + sqrtdr * (urst.my * urst_d_inv) +
                                  bxsig * (urst.by - ulst.by));
    const double tmz = invsumd * (sqrtdl * (ulst.mz * ulst_d_inv) + sqrtdr * (urst.mz * urst_d_inv) +
                                  bxsig * (urst.bz - ulst.bz));
    const double tby = invsumd * (sqrtdl * urst.by + sqrtdr * ulst.by +
                                  bxsig * sqrtdl * sqrtdr * ((urst.my * urst_d_inv) - (ulst.my * ulst_d_inv)));
    const double tbz = invsumd * (sqrtdl * urst.bz + sqrtdr * ulst.bz +
                                  bxsig * sqrtdl * sqrtdr * ((urst.mz * urst_d_inv) - (ulst.mz * ulst_d_inv)));
    // eqn (63): the bracket is formed with the LEFT double-star momenta on both sides (:289)
    const double dl_my = ulst.d * tmy, dl_mz = ulst.d * tmz;
#ifdef APK_PLAIN_SQRT
    const double tmp_e = s2 * bxi + (dl_my * tby + dl_mz * tbz) / ulst.d;
#else
    const double tmp_e = s2 * bxi + (dl_my * tby + dl_mz * tbz) * ulst_d_inv;
#endif
    const double sq_sig = pick(left, sqrtdl, sqrtdr) * bxsig;
    const double e_l = ust.e - sq_sig * (vbst - tmp_e);
    const double e_r = ust.e + sq_sig * (vbst - tmp_e);
    if (!dst_degenerate) {
      udst.my = ust.d * tmy;
      udst.mz = ust.d * tmz;
      udst.by = tby;
      udst.bz = tbz;
      udst.e = left ? e_l : e_r;
    }
  }

  hlld_jump(udst, ust, s_inner);  // double-star before star: the star state is overwritten next
  hlld_jump(ust, u, s_outer);
  // F, F + j_outer or (F + j_outer) + j_inner: the additions run under the lanes' execution mask instead of selecting
  // among three sums afterwards -- 14 masked v_add_f64 and a few scalar instructions instead of 14 additions + 28
  // v_cndmask_b32.  (The empty volatile asm statements keep the compiler from turning the branches back into
  // selects; the association (F + j_outer) + j_inner is the reference's, glmmhd_hlld.hpp:340-383.)
  Cons1D fl = fx;
  if (!outer) {
    asm volatile("" ::: );
    fl.d += ust.d;
    fl.mx += ust.mx;
    fl.my += ust.my;
    fl.mz += ust.mz;
    fl.e += ust.e;
    fl.by += ust.by;
    fl.bz += ust.bz;
    if (with_dstar) {
      asm volatile("" ::: );
      fl.d += udst.d;
      fl.mx += udst.mx;
      fl.my += udst.my;
      fl.mz += udst.mz;
      fl.e += udst.e;
      fl.by += udst.by;
      fl.bz += udst.bz;
    }
  }
  f[IDN] = fl.d;
  f[IV1] = fl.mx;
  f[IV2] = fl.my;
  f[IV3] = fl.mz;
  f[IEN] = fl.e;
  f[IB2] = fl.by;
  f[IB3] = fl.bz;
}

// src/hydro/rsolvers/glmmhd_dc_llf.hpp:46-179
APK_DEV void glmmhd_llf(const double (&wl)[NGLMMHD], const double (&wr)[NGLMMHD], double gamma,
                        double c_h, double (&f)[NGLMMHD]) {
  const double igm1 = 1.0 / (gamma - 1.0);
  double bxi, psii;
  glm_interface(wl, wr, c_h, bxi, psii);
  double qa = wl[IDN] * wl[IV1];
  double qb = wr[IDN] * wr[IV1];
  const double qc = 0.5 * (sqr(wl[IB2]) + sqr(wl[IB3]) - sqr(bxi));
  const double qd = 0.5 * (sqr(wr[IB2]) + sqr(wr[IB3]) - sqr(bxi));
  const double fs_d = qa + qb;
  double fs_mx = qa * wl[IV1] + qb * wr[IV1] + qc + qd;
  const double fs_my = qa * wl[IV2] + qb * wr[IV2] - bxi * (wl[IB2] + wr[IB2]);
  const double fs_mz = qa * wl[IV3] + qb * wr[IV3] - bxi * (wl[IB3] + wr[IB3]);
  const double fs_by = wl[IB2] * wl[IV1] + wr[IB2] * wr[IV1] - bxi * (wl[IV2] + wr[IV2]);
  const double fs_bz = wl[IB3] * wl[IV1] + wr[IB3] * wr[IV1] - bxi * (wl[IV3] + wr[IV3]);
  const double el = wl[IPR] * igm1 +
                    0.5 * wl[IDN] * (sqr(wl[IV1]) + sqr(wl[IV2]) + sqr(wl[IV3])) + qc + sqr(bxi);
  const double er = wr[IPR] * igm1 +
                    0.5 * wr[IDN] * (sqr(wr[IV1]) + sqr(wr[IV2]) + sqr(wr[IV3])) + qd + sqr(bxi);
  fs_mx += (wl[IPR] + wr[IPR]);
  double fs_e = (el + wl[IPR] + qc) * wl[IV1] + (er + wr[IPR] + qd) * wr[IV1];
  fs_e -= bxi * (wl[IB2] * wl[IV2] + wl[IB3] * wl[IV3]);
  fs_e -= bxi * (wr[IB2] * wr[IV2] + wr[IB3] * wr[IV3]);
  qa = fast_speed(gamma, wl[IDN], wl[IPR], wl[IB1], wl[IB2], wl[IB3]);
  qb = fast_speed(gamma, wr[IDN], wr[IPR], wr[IB1], wr[IB2], wr[IB3]);
  const double a = fmax((fabs(wl[IV1]) + qa), (fabs(wr[IV1]) + qb));
  const double du_d = a * (wr[IDN] - wl[IDN]);
  const double du_mx = a * (wr[IDN] * wr[IV1] - wl[IDN] * wl[IV1]);
  const double du_my = a * (wr[IDN] * wr[IV2] - wl[IDN] * wl[IV2]);
  const double du_mz = a * (wr[IDN] * wr[IV3] - wl[IDN] * wl[IV3]);
  const double du_e = a * (er - el);
  const double du_by = a * (wr[IB2] - wl[IB2]);
  const double du_bz = a * (wr[IB3] - wl[IB3]);
  f[IDN] = 0.5 * (fs_d - du_d);
  f[IV1] = 0.5 * (fs_mx - du_mx);
  f[IV2] = 0.5 * (fs_my - du_my);
  f[IV3] = 0.5 * (fs_mz - du_mz);
  f[IEN] = 0.5 * (fs_e - du_e);
  f[IB1] = psii;
  f[IB2] = 0.5 * (fs_by - du_by);
  f[IB3] = 0.5 * (fs_bz - du_bz);
  f[IPS] = sqr(c_h) * bxi;
}

// compile-time dispatch: Riemann<fluid, rsolver>::Solve for one face
template <int FLUID, int RS>
APK_DEV void riemann(const double (&wl)[nvars<FLUID>()], const double (&wr)[nvars<FLUID>()],
                     double gamma, double c_h, double (&f)[nvars<FLUID>()]) {
  if constexpr (RS == APK_RS_NONE) {  // rsolvers.hpp:35-63
#pragma unroll
    for (int n = 0; n < nvars<FLUID>(); ++n) f[n] = 0.0;
  } else if constexpr (FLUID == APK_FLUID_EULER) {
    if constexpr (RS == APK_RS_HLLE) hydro_hlle(wl, wr, gamma, f);
    else if constexpr (RS == APK_RS_HLLC) hydro_hllc(wl, wr, gamma, f);
    else hydro_llf(wl, wr, gamma, f);
  } else {
    if constexpr (RS == APK_RS_HLLE) glmmhd_hlle(wl, wr, gamma, c_h, f);
    else if constexpr (RS == APK_RS_HLLD) glmmhd_hlld(wl, wr, gamma, c_h, f);
    else glmmhd_llf(wl, wr, gamma, c_h, f);
  }
}

// the same for the fused stage kernels, which carry the stage's derived constants in scalar registers
template <int FLUID, int RS>
APK_DEV void riemann(const double (&wl)[nvars<FLUID>()], const double (&wr)[nvars<FLUID>()], const StageConsts &k,
                     double (&f)[nvars<FLUID>()]) {
  if constexpr (FLUID == APK_FLUID_GLMMHD && RS == APK_RS_HLLD) glmmhd_hlld(wl, wr, k, f);
  else riemann<FLUID, RS>(wl, wr, k.gamma, k.c_h, f);
}

// Direction permutation (e.g. glmmhd_hlld.hpp:45-49): natural index of the permuted slot.
// DIR = 1,2,3 ; slot in {IDN,IV1,IV2,IV3,IPR/IEN,IB1,IB2,IB3,IPS}
template <int DIR>
constexpr int perm(int slot) {
  constexpr int vx = DIR, vy = IV1 + ((DIR - IV1) + 1) % 3, vz = IV1 + ((DIR - IV1) + 2) % 3;
  return slot == IV1   ? vx
         : slot == IV2 ? vy
         : slot == IV3 ? vz
         : slot == IB1 ? vx - 1 + NHYDRO
         : slot == IB2 ? vy - 1 + NHYDRO
         : slot == IB3 ? vz - 1 + NHYDRO
                       : slot;
}

// ======================================================================================
// cons -> prim for one cell (src/eos/adiabatic_hydro.hpp:52-142, adiabatic_glmmhd.hpp:62-167)
// u/w hold the NH hydro/MHD variables; returns APK_FLAG_* bits.  u may be modified.
// ======================================================================================
// LEAN = 1: the caller guarantees eos.vceil = eos.eceil = +inf and eos.pfloor <= 0 (eos_is_lean: the defaults of
// hydro.cpp:507-537, no velocity ceiling, no pressure floor, no internal-energy ceiling), so the three blocks those
// parameters guard can never act and are not compiled: same results, no registers for their constants, and `u` is
// only ever modified by the density floor and the internal-energy floor.
// LEAN = 2: the same with the pressure floor's block compiled (eos_is_lean_but_pfloor: inputs/orszag_tang.in sets one).
// LEAN = 0: everything.
constexpr int LEAN_PFLOOR = 2;
template <int FLUID, int LEAN = 0>
APK_DEV unsigned cons_to_prim_core(const apk_eos &eos, double gm1, double vceil_sq, double pfloor_over_gm1,
                                   double (&u)[nvars<FLUID>()], double (&w)[nvars<FLUID>()], double &di_out);
inline __host__ __device__ bool eos_is_lean(const apk_eos &eos) {
  return eos.vceil == __builtin_inf() && eos.eceil == __builtin_inf() && eos.pfloor <= 0.0;
}
inline __host__ __device__ bool eos_is_lean_but_pfloor(const apk_eos &eos) {
  return eos.vceil == __builtin_inf() && eos.eceil == __builtin_inf();
}
template <int FLUID>
APK_DEV unsigned cons_to_prim_cell(const apk_eos &eos, double (&u)[nvars<FLUID>()],
                                   double (&w)[nvars<FLUID>()], double &di_out) {
  const double gm1 = eos.gamma - 1.0;
  return cons_to_prim_core<FLUID>(eos, gm1, sqr(eos.vceil), eos.pfloor / gm1, u, w, di_out);
}
// with the stage's constants from the host (StageConsts)
template <int FLUID, int LEAN = 0>
APK_DEV unsigned cons_to_prim_cell(const apk_eos &eos, const StageConsts &k, double (&u)[nvars<FLUID>()],
                                   double (&w)[nvars<FLUID>()], double &di_out) {
  return cons_to_prim_core<FLUID, LEAN>(eos, k.eos_gm1, k.vceil_sq, k.pfloor_over_gm1, u, w, di_out);
}
template <int FLUID, int LEAN>
APK_DEV unsigned cons_to_prim_core(const apk_eos &eos, double gm1, double vceil_sq, double pfloor_over_gm1,
                                   double (&u)[nvars<FLUID>()], double (&w)[nvars<FLUID>()], double &di_out) {
  // No contraction in here, in the product build either: the primitives of a cell must not depend on which kernel this
  // function was inlined into.  Stages that derive their input from the conserved state convert the same cell in several
  // kernels -- on a refined mesh the stage kernels AND the kernel of the flux correction's boundary planes, whose flux
  // through a coarse-fine face has to be the one the stage applied: with primitives that differed in the last bit between
  // the two, PPM's limiters now and then decided differently, the correction subtracted a flux the stage had not used, and
  // the mass of the refined MHD blast drifted by 1e-6 in 1500 cycles (round 6, tools/soak_r06.py; 2e-16 with this).
  // (frcp's Newton steps are explicit fma calls: the same everywhere.)
#pragma clang fp contract(off)
  constexpr bool mhd = (FLUID == APK_FLUID_GLMMHD);
  unsigned flags = 0;
  if (!(strictly_positive(u[IDN]) || eos.dfloor > 0.0)) flags |= APK_FLAG_NEG_DENSITY;
  u[IDN] = (u[IDN] > eos.dfloor) ? u[IDN] : eos.dfloor;
  w[IDN] = u[IDN];
#ifdef APK_PLAIN_SQRT
  const double di = 1.0 / u[IDN];
#else
  const double di = frcp(u[IDN]);  // (the compiler's own reciprocal-math quotient takes one Newton step more)
#endif
  di_out = di;
  w[IV1] = u[IM1] * di;
  w[IV2] = u[IM2] * di;
  w[IV3] = u[IM3] * di;
  double e_B = 0.0;
  double e_k = 0.5 * di * (sqr(u[IM1]) + sqr(u[IM2]) + sqr(u[IM3]));
  if constexpr (mhd) {
    w[IB1] = u[IB1];
    w[IB2] = u[IB2];
    w[IB3] = u[IB3];
    w[IPS] = u[IPS];
    e_B = 0.5 * (sqr(u[IB1]) + sqr(u[IB2]) + sqr(u[IB3]));
    w[IPR] = gm1 * (u[IEN] - e_k - e_B);
  } else {
    w[IPR] = gm1 * (u[IEN] - e_k);
  }
  if constexpr (!LEAN) {
  const double v2 = sqr(w[IV1]) + sqr(w[IV2]) + sqr(w[IV3]);
  if (v2 > vceil_sq) {
    const double v = sqrt(v2);
    w[IV1] *= eos.vceil / v;
    w[IV2] *= eos.vceil / v;
    w[IV3] *= eos.vceil / v;
    u[IM1] *= eos.vceil / v;
    u[IM2] *= eos.vceil / v;
    u[IM3] *= eos.vceil / v;
    const double e_k_new = 0.5 * u[IDN] * vceil_sq;
    u[IEN] -= e_k - e_k_new;
    e_k = e_k_new;
  }
  }
  if constexpr (LEAN == 1) {
    if (!(strictly_positive(w[IPR]) || eos.efloor > 0.0)) flags |= APK_FLAG_NEG_PRESSURE;
  } else {
    if (!(strictly_positive(w[IPR]) || eos.pfloor > 0.0 || eos.efloor > 0.0)) flags |= APK_FLAG_NEG_PRESSURE;
    if ((eos.pfloor > 0.0) && (w[IPR] < eos.pfloor)) {
      if constexpr (mhd) u[IEN] = pfloor_over_gm1 + e_k + e_B;
      else u[IEN] = pfloor_over_gm1 + e_k;
      w[IPR] = eos.pfloor;
    }
  }
  const double eff_floor = gm1 * u[IDN] * eos.efloor;
  if (w[IPR] < eff_floor) {
    if constexpr (mhd) u[IEN] = (u[IDN] * eos.efloor) + e_k + e_B;
    else u[IEN] = (u[IDN] * eos.efloor) + e_k;
    w[IPR] = eff_floor;
  }
  if constexpr (!LEAN) {
    const double eff_ceil = gm1 * u[IDN] * eos.eceil;
    if (w[IPR] > eff_ceil) {
      if constexpr (mhd) u[IEN] = (u[IDN] * eos.eceil) + e_k + e_B;
      else u[IEN] = (u[IDN] * eos.eceil) + e_k;
      w[IPR] = eff_ceil;
    }
  }
  return flags;
}

// ======================================================================================
// EstimateHyperbolicTimestep of one cell (src/hydro/hydro.cpp:845-895): min over the active directions of
// dx_d / (|v_d| + c_d), c_d the sound speed / the fast speed along d.  `di` = 1 / w[IDN] as ConsToPrim left it.
// The parity build evaluates the reference's expressions (three independent fast speeds).  In the product build the
// three fast speeds share what does not depend on the direction -- gamma p, |B|^2, the sum and the difference of the
// two, their square, and the reciprocal of the density ConsToPrim already has: 2 roots + 5 operations per direction
// instead of 2 roots + a divide + 12 -- and the quotients dx / s multiply by a Newton reciprocal (fdiv).  The result
// agrees with the parity build's to a few ulp (a time step: every cell of the mesh sees the same dt).
// ======================================================================================
template <int FLUID>
APK_DEV double cell_dt_hyp(double gamma, const double (&w)[nvars<FLUID>()], double di, int ndim, double dx0, double dx1,
                           double dx2) {
#ifdef APK_FP_STRICT
  double lx, ly = 0.0, lz = 0.0;
  if constexpr (FLUID == APK_FLUID_EULER) {
    lx = sound_speed(gamma, w[IDN], w[IPR]);
    ly = lx;
    lz = lx;
  } else {
    lx = fast_speed(gamma, w[IDN], w[IPR], w[IB1], w[IB2], w[IB3]);
    if (ndim > 1) ly = fast_speed(gamma, w[IDN], w[IPR], w[IB2], w[IB3], w[IB1]);
    if (ndim > 2) lz = fast_speed(gamma, w[IDN], w[IPR], w[IB3], w[IB1], w[IB2]);
  }
  double m = dx0 / (fabs(w[IV1]) + lx);
  if (ndim > 1) m = fmin(m, dx1 / (fabs(w[IV2]) + ly));
  if (ndim > 2) m = fmin(m, dx2 / (fabs(w[IV3]) + lz));
  return m;
#else
  const double asq = gamma * w[IPR];
  if constexpr (FLUID == APK_FLUID_EULER) {
    const double cs = fsqrt(asq * di);
    double m = fdiv(dx0, fabs(w[IV1]) + cs);
    if (ndim > 1) m = fmin(m, fdiv(dx1, fabs(w[IV2]) + cs));
    if (ndim > 2) m = fmin(m, fdiv(dx2, fabs(w[IV3]) + cs));
    return m;
  } else {
    const double bsq = w[IB1] * w[IB1] + (w[IB2] * w[IB2] + w[IB3] * w[IB3]);
    const double qsq = bsq + asq, dif = bsq - asq;
    const double dif2 = dif * dif, a4 = 4.0 * asq, hdi = 0.5 * di;
    // c_f^2 = (q + sqrt(dif^2 + 4 a^2 (|B|^2 - B_d^2))) / (2 rho); the transverse field energy is clamped at zero
    // (|B|^2 - B_d^2 may round to -1 ulp of |B|^2 when the field is along d, which would put a negative number under
    // the root where dif = 0)
    auto cf = [&](double bn) { return fsqrt((qsq + fsqrt(fma(a4, fmax(fma(-bn, bn, bsq), 0.0), dif2))) * hdi); };
    double m = fdiv(dx0, fabs(w[IV1]) + cf(w[IB1]));
    if (ndim > 1) m = fmin(m, fdiv(dx1, fabs(w[IV2]) + cf(w[IB2])));
    if (ndim > 2) m = fmin(m, fdiv(dx2, fabs(w[IV3]) + cf(w[IB3])));
    return m;
  }
#endif
}

}  // namespace apk
