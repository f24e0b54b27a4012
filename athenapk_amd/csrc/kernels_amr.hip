// kernels_amr.hip -- mesh-refinement operators (prolongation, restriction of cells and of face
// fluxes) as batched index-box plans, and block tagging.  All of it is HBM-bound streaming work:
// one thread per coarse cell and variable, i fastest, so reads of the coarse buffer and the
// paired (fi, fi+1) writes of the fine array coalesce; one launch covers every box of the plan
// (blockIdx.x = box), which is what matters for AMR meshes made of many 16^3 blocks.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "apk_internal.hpp"
#include "hydro_math.hpp"

// (merged plans, apk_flux_fix_plan_create_merged: the regions of all directions in ONE launch)
struct FixRegionBox {
  int lo[3];  // first cell of the region, as indices of the destination block's array
  int dir;    // direction of the face the region corrects
  // the other regions of the plan that share cells with this one, lower directions first; their boxes ride along so that
  // the test "is this cell theirs too" reads nothing else (a second record per partner made the launch latency-bound)
  struct Partner {
    int idx, dir;  // idx < 0: none
    int lo[3], ext[3];
  } partner[8];
};
struct apk_flux_fix_plan {
  apk_flux_fix_region *d_regions = nullptr;
  int n = 0;
  int64_t max_items = 0;
  FixRegionBox *d_boxes = nullptr;  // non-null: a merged plan
};

struct apk_refine_plan {
  apk_refine_geom geom{};
  int nvar = 0, nops = 0;
  int64_t max_items = 0;
  apk_refine_op *d_ops = nullptr;
  // work list: the boxes cut into chunks of kRefineChunkItems (cell, variable) items, one workgroup
  // each, one item per thread (a plan mixes 16 x 16 x 4 faces with 4^3 corners; several items per
  // thread measured slower for these operators)
  apk_copy_chunk *d_chunks = nullptr;
  int nchunks = 0;
};

namespace apk {
namespace {

constexpr int kRefineChunkItems = 256;

struct RefineDims {
  int DIM;
  int fn[3], cn[3];  // cell-centred array extents
  int fs[3], cs[3];  // first interior index
};

__host__ __device__ inline RefineDims refine_dims(const apk_refine_geom &g) {
  RefineDims r;
  r.DIM = (g.nx[2] > 1) ? 3 : ((g.nx[1] > 1) ? 2 : 1);
  for (int d = 0; d < 3; ++d) {
    const bool act = (d == 0) || g.nx[d] > 1;
    r.fn[d] = act ? g.nx[d] + 2 * g.ng : 1;
    r.cn[d] = act ? g.nx[d] / 2 + 2 * g.cng : 1;
    r.fs[d] = (d < r.DIM) ? g.ng : 0;
    r.cs[d] = (d < r.DIM) ? g.cng : 0;
  }
  return r;
}

APK_DEV double sign_of(double a) { return (a < 0.) ? -1. : 1.; }
// UniformCartesian cell centre of an index space whose first interior index is s
APK_DEV double xc_of(double xmin, double dx, int s, int idx) { return (xmin - s * dx) + (idx + 0.5) * dx; }

struct Spacing {
  double dxm, dxp, dxfm, dxfp;
};
// Parthenon refinement_ops::util::GetGridSpacings<DIM, CC>
APK_DEV Spacing spacings(double xmin, double dx, int cs, int fs, int ci, int fi) {
  const double cdx = 2.0 * dx;
  const double xm = xc_of(xmin, cdx, cs, ci - 1);
  const double xc = xc_of(xmin, cdx, cs, ci);
  const double xp = xc_of(xmin, cdx, cs, ci + 1);
  const double fxm = xc_of(xmin, dx, fs, fi);
  const double fxp = xc_of(xmin, dx, fs, fi + 1);
  return {xc - xm, xp - xc, xc - fxm, fxp - xc};
}
// Parthenon refinement_ops::util::GradMinMod
APK_DEV double grad_minmod(double fc, double fm, double fp, double dxm, double dxp) {
  const double gxm = (fc - fm) / dxm;
  const double gxp = (fp - fc) / dxp;
  return 0.5 * (sign_of(gxm) + sign_of(gxp)) * fmin(fabs(gxm), fabs(gxp));
}

// custom_ops.hpp:60-183 (el == CC)
template <int DIM>
APK_DEV void prolongate_cell(const apk_refine_geom &g, const RefineDims &r, const apk_refine_op &op, int v, int k,
                             int j, int i) {
  const int64_t csj = r.cn[0], csk = (int64_t)r.cn[0] * r.cn[1], csn = csk * r.cn[2];
  const int64_t fsj = r.fn[0], fsk = (int64_t)r.fn[0] * r.fn[1], fsn = fsk * r.fn[2];
  const double *c = op.src + v * csn + k * csk + j * csj + i;
  const int fi = (i - r.cs[0]) * 2 + r.fs[0];
  const int fj = (DIM > 1) ? (j - r.cs[1]) * 2 + r.fs[1] : r.fs[1];
  const int fk = (DIM > 2) ? (k - r.cs[2]) * 2 + r.fs[2] : r.fs[2];
  double *f = op.dst + v * fsn + fk * fsk + fj * fsj + fi;
  const double fc = c[0];
  const Spacing s1 = spacings(op.xmin[0], g.dx[0], r.cs[0], r.fs[0], i, fi);
  double gx1c = grad_minmod(fc, c[-1], c[1], s1.dxm, s1.dxp);
  Spacing s2{0, 0, 0, 0}, s3{0, 0, 0, 0};
  double gx2c = 0, gx3c = 0;
  if constexpr (DIM > 1) {
    s2 = spacings(op.xmin[1], g.dx[1], r.cs[1], r.fs[1], j, fj);
    gx2c = grad_minmod(fc, c[-csj], c[csj], s2.dxm, s2.dxp);
  }
  if constexpr (DIM > 2) {
    s3 = spacings(op.xmin[2], g.dx[2], r.cs[2], r.fs[2], k, fk);
    gx3c = grad_minmod(fc, c[-csk], c[csk], s3.dxm, s3.dxp);
  }
  double dqmax = fabs(gx1c) * fmax(s1.dxfm, s1.dxfp);
  if constexpr (DIM > 1) dqmax += fabs(gx2c) * fmax(s2.dxfm, s2.dxfp);
  if constexpr (DIM > 2) dqmax += fabs(gx3c) * fmax(s3.dxfm, s3.dxfp);
  constexpr int jlim = (DIM > 1) ? 1 : 0, klim = (DIM > 2) ? 1 : 0;
  double qmin = fc, qmax = fc;
#pragma unroll
  for (int koff = -klim; koff <= klim; koff++)
#pragma unroll
    for (int joff = -jlim; joff <= jlim; joff++)
#pragma unroll
      for (int ioff = -1; ioff <= 1; ioff++) {
        const double q = c[koff * csk + joff * csj + ioff];
        qmin = fmin(qmin, q);
        qmax = fmax(qmax, q);
      }
  double alpha = 1.0;
  if (dqmax * alpha > (qmax - fc)) alpha = (qmax - fc) / dqmax;
  if (dqmax * alpha > (fc - qmin)) alpha = (fc - qmin) / dqmax;
  gx1c *= alpha;
  gx2c *= alpha;
  gx3c *= alpha;
  const double dx1fm = s1.dxfm, dx1fp = s1.dxfp, dx2fm = s2.dxfm, dx2fp = s2.dxfp, dx3fm = s3.dxfm, dx3fp = s3.dxfp;
  f[0] = fc - (gx1c * dx1fm + gx2c * dx2fm + gx3c * dx3fm);
  f[1] = fc + (gx1c * dx1fp - gx2c * dx2fm - gx3c * dx3fm);
  if constexpr (DIM > 1) {
    f[fsj] = fc - (gx1c * dx1fm - gx2c * dx2fp + gx3c * dx3fm);
    f[fsj + 1] = fc + (gx1c * dx1fp + gx2c * dx2fp - gx3c * dx3fm);
  }
  if constexpr (DIM > 2) {
    f[fsk] = fc - (gx1c * dx1fm + gx2c * dx2fm - gx3c * dx3fp);
    f[fsk + 1] = fc + (gx1c * dx1fp - gx2c * dx2fm + gx3c * dx3fp);
    f[fsk + fsj] = fc - (gx1c * dx1fm - gx2c * dx2fp - gx3c * dx3fp);
    f[fsk + fsj + 1] = fc + (gx1c * dx1fp + gx2c * dx2fp + gx3c * dx3fp);
  }
}

// RestrictAverage for cells (el = 0) and faces (el = 1..3): uniform weights, pairwise sums
// (face_geom: the arrays have one more entry along the face direction, Parthenon's face fields;
// otherwise both are cell-shaped and face i is the left face of cell i, apk_block_desc.flux)
APK_DEV void restrict_cell(const apk_refine_geom &g, const RefineDims &r, const apk_refine_op &op, int el, bool face_geom,
                           int v, int k, int j, int i) {
  int fn[3] = {r.fn[0], r.fn[1], r.fn[2]}, cn[3] = {r.cn[0], r.cn[1], r.cn[2]};
  if (el >= 1 && face_geom) {
    fn[el - 1] += 1;
    cn[el - 1] += 1;
  }
  const int64_t csj = cn[0], csk = (int64_t)cn[0] * cn[1], csn = csk * cn[2];
  const int64_t fsj = fn[0], fsk = (int64_t)fn[0] * fn[1], fsn = fsk * fn[2];
  const int fi = (i - r.cs[0]) * 2 + r.fs[0];
  const int fj = (r.DIM > 1) ? (j - r.cs[1]) * 2 + r.fs[1] : 0;
  const int fk = (r.DIM > 2) ? (k - r.cs[2]) * 2 + r.fs[2] : 0;
  const int oi1 = (el != 1) ? 1 : 0;
  const int oj1 = (r.DIM > 1 && el != 2) ? 1 : 0;
  const int ok1 = (r.DIM > 2 && el != 3) ? 1 : 0;
  double w = 1.0;
  if (el != 1) w *= g.dx[0];
  if (el != 2) w *= g.dx[1];
  if (el != 3) w *= g.dx[2];
  const double *f = op.src + v * fsn + fk * fsk + fj * fsj + fi;
  double vol[2][2][2], t[2][2][2];
#pragma unroll
  for (int ok = 0; ok < 2; ++ok)
#pragma unroll
    for (int oj = 0; oj < 2; ++oj)
#pragma unroll
      for (int oi = 0; oi < 2; ++oi) {
        const bool in = !(ok > ok1 || oj > oj1 || oi > oi1);
        vol[ok][oj][oi] = in ? w : 0.0;
        t[ok][oj][oi] = in ? w * f[ok * fsk + oj * fsj + oi] : 0.0;
      }
  const double tvol = ((vol[0][0][0] + vol[0][1][0]) + (vol[0][0][1] + vol[0][1][1])) +
                      ((vol[1][0][0] + vol[1][1][0]) + (vol[1][0][1] + vol[1][1][1]));
  op.dst[v * csn + k * csk + j * csj + i] = (((t[0][0][0] + t[0][1][0]) + (t[0][0][1] + t[0][1][1])) +
                                             ((t[1][0][0] + t[1][1][0]) + (t[1][0][1] + t[1][1][1]))) /
                                            tvol;
}

__global__ void __launch_bounds__(256) refine_ops_kernel(apk_refine_geom g, int nvar, const apk_refine_op *ops,
                                                         const apk_copy_chunk *chunks) {
  const apk_copy_chunk ch = chunks[blockIdx.x];
  const apk_refine_op op = ops[ch.region];
  if (op.dx[0] > 0.0) {  // the box brings its own cell widths (another refinement level)
    g.dx[0] = op.dx[0];
    g.dx[1] = op.dx[1];
    g.dx[2] = op.dx[2];
  }
  const RefineDims r = refine_dims(g);
  const int e0 = op.hi[0] - op.lo[0] + 1, e1 = op.hi[1] - op.lo[1] + 1, e2 = op.hi[2] - op.lo[2] + 1;
  // (32-bit index arithmetic: apk_refine_plan_create refuses boxes of 2^31 items or more, and the three 64-bit
  // divisions per item made this kernel compute-bound on the 8^3 boxes of 16^3 blocks)
  const unsigned cells = (unsigned)e0 * (unsigned)e1 * (unsigned)e2, items = cells * (unsigned)nvar;
  const unsigned end = ((unsigned)ch.first + (unsigned)kRefineChunkItems < items) ? (unsigned)ch.first + (unsigned)kRefineChunkItems : items;
  for (unsigned t = (unsigned)ch.first + threadIdx.x; t < end; t += 256u) {
    const int v = (int)(t / cells);
    unsigned c = t - (unsigned)v * cells;
    const int i = op.lo[0] + (int)(c % (unsigned)e0);
    c /= (unsigned)e0;
    const int j = op.lo[1] + (int)(c % (unsigned)e1);
    const int k = op.lo[2] + (int)(c / (unsigned)e1);
    if (op.kind == APK_RO_PROLONGATE) {
      if (r.DIM == 3) prolongate_cell<3>(g, r, op, v, k, j, i);
      else if (r.DIM == 2) prolongate_cell<2>(g, r, op, v, k, j, i);
      else prolongate_cell<1>(g, r, op, v, k, j, i);
    } else {
      const bool flux_shaped = op.kind >= APK_RO_RESTRICT_FLUX1;
      restrict_cell(g, r, op, flux_shaped ? op.kind - APK_RO_RESTRICT_FLUX1 + 1 : op.kind - APK_RO_RESTRICT_CELL,
                    !flux_shaped, v, k, j, i);
    }
  }
}

// the correction of element (i, j, k, v) of a region: beta_dt * scale * (fine average - coarse flux) [* psi_factor]
template <class R>
APK_DEV double flux_fix_term(const R &r, int i, int j, int k, int v, int64_t d, double beta_dt, int psi_var, double psi_factor) {
    const int64_t so = i * r.src_stride[0] + j * r.src_stride[1] + k * r.src_stride[2] + v * r.src_stride[3];
    double avg;
    if (r.average == 0) {
      avg = r.fine_avg[so];
    } else {  // restrict_cell(el = r.average, cell-shaped arrays) on the fine block's flux array, value for value
      const double *f = r.fine_avg + so;
      const int el = r.average;
      const int oi1 = (el != 1) ? 1 : 0;
      const int oj1 = (r.ndim > 1 && el != 2) ? 1 : 0;
      const int ok1 = (r.ndim > 2 && el != 3) ? 1 : 0;
      const double w = r.fine_area;
      double vol[2][2][2], tt[2][2][2];
#pragma unroll
      for (int ok = 0; ok < 2; ++ok)
#pragma unroll
        for (int oj = 0; oj < 2; ++oj)
#pragma unroll
          for (int oi = 0; oi < 2; ++oi) {
            const bool in = !(ok > ok1 || oj > oj1 || oi > oi1);
            vol[ok][oj][oi] = in ? w : 0.0;
            tt[ok][oj][oi] = in ? w * f[ok * r.fine_stride[2] + oj * r.fine_stride[1] + oi * r.fine_stride[0]] : 0.0;
          }
      const double tvol = ((vol[0][0][0] + vol[0][1][0]) + (vol[0][0][1] + vol[0][1][1])) +
                          ((vol[1][0][0] + vol[1][1][0]) + (vol[1][0][1] + vol[1][1][1]));
      avg = (((tt[0][0][0] + tt[0][1][0]) + (tt[0][0][1] + tt[0][1][1])) + ((tt[1][0][0] + tt[1][1][0]) + (tt[1][0][1] + tt[1][1][1]))) /
            tvol;
    }
    double dv = (beta_dt * r.scale) * (avg - r.coarse_flux[d]);
    if (v == psi_var) dv *= psi_factor;
    return dv;
}

// flux correction after a fused stage: one thread per (face cell, variable) of a region
__global__ void __launch_bounds__(256) flux_fix_kernel(const apk_flux_fix_region *regions, double beta_dt, int psi_var,
                                                       double psi_factor) {
  const apk_flux_fix_region r = regions[blockIdx.x];
  const int64_t cells = (int64_t)r.ext[0] * r.ext[1] * r.ext[2], items = cells * r.nvar;
  for (int64_t t = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; t < items; t += (int64_t)gridDim.y * blockDim.x) {
    const int v = (int)(t / cells);
    int64_t c = t - (int64_t)v * cells;
    const int i = (int)(c % r.ext[0]);
    c /= r.ext[0];
    const int j = (int)(c % r.ext[1]);
    const int k = (int)(c / r.ext[1]);
    const int64_t d = i * r.dst_stride[0] + j * r.dst_stride[1] + k * r.dst_stride[2] + v * r.dst_stride[3];
    r.cons[d] += flux_fix_term(r, i, j, k, v, d, beta_dt, psi_var, psi_factor);
  }
}

// The same for the regions of ALL directions in one launch.  A coarse cell on an edge of its block can lie next to two or
// three coarse-fine faces; the launches per direction corrected it direction by direction, ((u + d1) + d2) + d3.  Here the
// region of the LOWEST direction that holds the cell owns it: it applies its own term and then the terms of the higher
// directions, in order (the regions that share cells with it are listed in its box record) -- same additions, same
// order, no two threads on one cell.
__global__ void __launch_bounds__(256) flux_fix_merged_kernel(const apk_flux_fix_region *regions, const FixRegionBox *boxes,
                                                              double beta_dt, int psi_var, double psi_factor) {
  const apk_flux_fix_region r = regions[blockIdx.x];
  const FixRegionBox &bx = boxes[blockIdx.x];
  // (32-bit index arithmetic: a merged plan's regions are faces of meshblocks, apk_flux_fix_plan_create_merged checks)
  const unsigned cells = (unsigned)r.ext[0] * (unsigned)r.ext[1] * (unsigned)r.ext[2], items = cells * (unsigned)r.nvar;
  for (unsigned t = blockIdx.y * blockDim.x + threadIdx.x; t < items; t += gridDim.y * blockDim.x) {
    const int v = (int)(t / cells);
    unsigned c = t - (unsigned)v * cells;
    const int i = (int)(c % (unsigned)r.ext[0]);
    c /= (unsigned)r.ext[0];
    const int j = (int)(c % (unsigned)r.ext[1]);
    const int k = (int)(c / (unsigned)r.ext[1]);
    const int64_t d = i * r.dst_stride[0] + j * r.dst_stride[1] + k * r.dst_stride[2] + v * r.dst_stride[3];
    const int ci = bx.lo[0] + i, cj = bx.lo[1] + j, ck = bx.lo[2] + k;  // the cell in its block
    bool owner = true;
    double later[2];
    int nlater = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const FixRegionBox::Partner &pb = bx.partner[q];  // (block-uniform; sorted by direction)
      if (pb.idx < 0) break;
      const int li = ci - pb.lo[0], lj = cj - pb.lo[1], lk = ck - pb.lo[2];
      if (li < 0 || lj < 0 || lk < 0 || li >= pb.ext[0] || lj >= pb.ext[1] || lk >= pb.ext[2]) continue;
      if (pb.dir < bx.dir) {
        owner = false;  // the region of the lower direction applies this region's term too
        break;
      }
      const apk_flux_fix_region *pr = regions + pb.idx;
      const int64_t pd = li * pr->dst_stride[0] + lj * pr->dst_stride[1] + lk * pr->dst_stride[2] + v * pr->dst_stride[3];
      if (nlater < 2) later[nlater++] = flux_fix_term(*pr, li, lj, lk, v, pd, beta_dt, psi_var, psi_factor);
    }
    if (!owner) continue;
    double u = r.cons[d] + flux_fix_term(r, i, j, k, v, d, beta_dt, psi_var, psi_factor);
    for (int q = 0; q < nlater; ++q) u += later[q];
    r.cons[d] = u;
  }
}

// ---- tagging ---------------------------------------------------------------------------------
// one workgroup row per (block, k-plane chunk); the criteria are non-negative, so their bit
// patterns order like unsigned integers and one atomicMax per workgroup suffices
// face_nbr (apk_stage_args.face_neighbor's table, or null): a ghost cell straight behind a face whose entry is >= 0 is
// read from that neighbour's interior instead -- the exchange in front of the check then leaves those zones out, as the
// stage loop's does (same values: the cells the copy would have brought).
template <int CRIT>
__global__ void __launch_bounds__(256) tag_kernel(PackView pv, unsigned long long *block_max, int kchunks, const int *face_nbr) {
  const int b = blockIdx.z / kchunks;
  const int chunk = blockIdx.z - b * kchunks;
  const apk_block_desc blk = pv.blocks[b];
  const int ndim = (pv.nx3 > 1) ? 3 : ((pv.nx2 > 1) ? 2 : 1);
  // extents per criterion (gradient.cpp:33-36,45-46,79-81; other.cpp:31-32)
  int il = pv.is, iu = pv.ie, jl = pv.js, ju = pv.je, kl = pv.ks, ku = pv.ke;
  if (CRIT == APK_TAG_PRESSURE_GRADIENT) {
    il -= 1, iu += 1, jl -= 1, ju += 1;
    if (ndim == 3) kl -= 1, ku += 1;
  } else if (CRIT == APK_TAG_VELOCITY_GRADIENT) {
    il -= 1, iu += 1, jl -= 1, ju += 1;
  } else {
    iu += 1;
  }
  double m = 0.0;
  int io, jo;
  const bool inside = rect_ij(pv.nx1 + 2, pv.nx2 + 2, io, jo);  // (the launch covers the widest extent)
  const int i = il + io, j = jl + jo;
  // (the k range in kchunks pieces, one workgroup each: a pack of a few hundred narrow blocks is too few workgroups
  // of 18 dependent plane-steps otherwise; the block's maximum is an atomicMax either way)
  const int klen = (ku - kl + kchunks) / kchunks;
  const int k0 = kl + chunk * klen, k1 = (k0 + klen - 1 < ku) ? k0 + klen - 1 : ku;
  // (the six neighbours' arrays once per thread: a descriptor load per access made the kernel twice as slow)
  const double *nbp0 = nullptr, *nbp1 = nullptr, *nbp2 = nullptr, *nbp3 = nullptr, *nbp4 = nullptr, *nbp5 = nullptr;
  if (face_nbr) {
    const int *fn = face_nbr + 6 * b;
    if (fn[0] >= 0) nbp0 = pv.blocks[fn[0]].prim;
    if (fn[1] >= 0) nbp1 = pv.blocks[fn[1]].prim;
    if (fn[2] >= 0) nbp2 = pv.blocks[fn[2]].prim;
    if (fn[3] >= 0) nbp3 = pv.blocks[fn[3]].prim;
    if (fn[4] >= 0) nbp4 = pv.blocks[fn[4]].prim;
    if (fn[5] >= 0) nbp5 = pv.blocks[fn[5]].prim;
  }
  // primitive `var` of cell (kk, jj, ii): the block's own array, or the interior of the block behind the one face the
  // cell lies behind
  auto at = [&](int var, int kk, int jj, int ii) -> double {
    const double *base = blk.prim;
    if (face_nbr) {
      const int gi = (ii < pv.is) ? 1 : ((ii > pv.ie) ? 2 : 0), gj = (jj < pv.js) ? 1 : ((jj > pv.je) ? 2 : 0),
                gk = (kk < pv.ks) ? 1 : ((kk > pv.ke) ? 2 : 0);
      if ((gi != 0) + (gj != 0) + (gk != 0) == 1) {
        const int f = gi ? gi - 1 : (gj ? 1 + gj : 3 + gk);
        const double *nb = (f == 0) ? nbp0 : (f == 1) ? nbp1 : (f == 2) ? nbp2 : (f == 3) ? nbp3 : (f == 4) ? nbp4 : nbp5;
        if (nb) {
          base = nb;
          if (gi) ii += (gi == 1) ? pv.nx1 : -pv.nx1;
          if (gj) jj += (gj == 1) ? pv.nx2 : -pv.nx2;
          if (gk) kk += (gk == 1) ? pv.nx3 : -pv.nx3;
        }
      }
    }
    return base[var * pv.sn + kk * pv.sk + jj * pv.sj + ii];
  };
  if (inside && i <= iu && j <= ju) {
    for (int k = k0; k <= k1; ++k) {
      const int64_t c = k * pv.sk + j * pv.sj + i;
      // (cells whose stencil stays inside the block's own interior: plain pointer arithmetic)
      const bool deep = !face_nbr || (i > pv.is && i < pv.ie && j > pv.js && j < pv.je && (ndim < 3 || (k > pv.ks && k < pv.ke)));
      if (CRIT == APK_TAG_PRESSURE_GRADIENT) {
        double a, bb, cc = 0.0, p0;
        if (deep) {
          const double *p = blk.prim + IPR * pv.sn + c;
          a = 0.5 * (p[1] - p[-1]), bb = 0.5 * (p[pv.sj] - p[-pv.sj]);
          if (ndim == 3) cc = 0.5 * (p[pv.sk] - p[-pv.sk]);
          p0 = p[0];
        } else {
          a = 0.5 * (at(IPR, k, j, i + 1) - at(IPR, k, j, i - 1)), bb = 0.5 * (at(IPR, k, j + 1, i) - at(IPR, k, j - 1, i));
          if (ndim == 3) cc = 0.5 * (at(IPR, k + 1, j, i) - at(IPR, k - 1, j, i));
          p0 = at(IPR, k, j, i);
        }
        double eps;
        if (ndim == 3) {
          eps = sqrt(sqr(a) + sqr(bb) + sqr(cc)) / p0;
        } else {
          eps = sqrt(sqr(a) + sqr(bb)) / p0;
        }
        m = fmax(m, eps);
      } else if (CRIT == APK_TAG_VELOCITY_GRADIENT) {
        double vgy, vgx;
        if (deep) {
          const double *v1 = blk.prim + IV1 * pv.sn + c, *v2 = blk.prim + IV2 * pv.sn + c;
          vgy = fabs(v2[1] - v2[-1]) * 0.5;
          vgx = fabs(v1[pv.sj] - v1[-pv.sj]) * 0.5;
        } else {
          vgy = fabs(at(IV2, k, j, i + 1) - at(IV2, k, j, i - 1)) * 0.5;
          vgx = fabs(at(IV1, k, j + 1, i) - at(IV1, k, j - 1, i)) * 0.5;
        }
        const double vg = sqrt(vgx * vgx + vgy * vgy);
        if (vg > m) m = vg;
      } else {
        m = fmax(m, deep ? blk.prim[IDN * pv.sn + c] : at(IDN, k, j, i));
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
  __shared__ double part[4];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  if ((tid & 63) == 0) part[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    const double mm = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
    atomicMax(block_max + b, (unsigned long long)__double_as_longlong(mm));
  }
}

}  // namespace
}  // namespace apk

using namespace apk;

extern "C" {

int apk_refine_plan_create(apk_ctx *ctx, const apk_refine_geom *geom, int nvar, const apk_refine_op *ops, int nops,
                           apk_refine_plan **out) {
  if (!ctx || !geom || !ops || !out || nvar <= 0 || nops <= 0) return set_err(ctx, APK_ERR_INVALID, "apk_refine_plan_create: bad argument");
  *out = nullptr;
  if (geom->ng < 1 || geom->cng < 1) return set_err(ctx, APK_ERR_NGHOST, "apk_refine_plan_create: ng, cng must be >= 1");
  for (int d = 0; d < 3; ++d)
    if (geom->nx[d] < 1 || (geom->nx[d] > 1 && (geom->nx[d] & 1)))
      return set_err(ctx, APK_ERR_INVALID, "apk_refine_plan_create: active dimensions need an even number of cells");
  const RefineDims r = refine_dims(*geom);
  int64_t max_items = 0;
  for (int n = 0; n < nops; ++n) {
    const apk_refine_op &op = ops[n];
    if (op.kind < APK_RO_PROLONGATE || op.kind > APK_RO_RESTRICT_FLUX3 || !op.src || !op.dst)
      return set_err(ctx, APK_ERR_INVALID, "apk_refine_plan_create: bad op");
    const bool flux_shaped = op.kind >= APK_RO_RESTRICT_FLUX1;
    const int fdir = flux_shaped ? op.kind - APK_RO_RESTRICT_FLUX1 + 1 : 0;
    const int el = (op.kind >= APK_RO_RESTRICT_FACE1 && !flux_shaped) ? op.kind - APK_RO_RESTRICT_CELL : 0;
    if (el > r.DIM || fdir > r.DIM) return set_err(ctx, APK_ERR_INVALID, "apk_refine_plan_create: face direction is collapsed");
    int64_t cells = 1;
    for (int d = 0; d < 3; ++d) {
      // prolongation reads one coarse cell either side in active dimensions
      const int halo = (op.kind == APK_RO_PROLONGATE && d < r.DIM) ? 1 : 0;
      const int top = r.cn[d] + ((el == d + 1) ? 1 : 0) - 1;
      if (op.lo[d] > op.hi[d] || op.lo[d] - halo < 0 || op.hi[d] + halo > top)
        return set_err(ctx, APK_ERR_INVALID, "apk_refine_plan_create: index box outside of the coarse buffer");
      cells *= op.hi[d] - op.lo[d] + 1;
    }
    max_items = cells * nvar > max_items ? cells * nvar : max_items;
  }
  if (max_items >= ((int64_t)1 << 31)) return set_err(ctx, APK_ERR_UNSUPPORTED, "apk_refine_plan_create: a box of 2^31 items or more");
  apk_refine_plan *p = new (std::nothrow) apk_refine_plan();
  if (!p) return APK_ERR_INVALID;
  p->geom = *geom;
  p->nvar = nvar;
  p->nops = nops;
  p->max_items = max_items;
  std::vector<apk_copy_chunk> chunks;
  for (int q = 0; q < nops; ++q) {
    int64_t items = nvar;
    for (int d = 0; d < 3; ++d) items *= ops[q].hi[d] - ops[q].lo[d] + 1;
    for (int64_t f = 0; f < items; f += apk::kRefineChunkItems) chunks.push_back({q, (int)f});
  }
  p->nchunks = (int)chunks.size();
  hipError_t e = hipMalloc(&p->d_ops, sizeof(apk_refine_op) * nops);
  if (e == hipSuccess) e = hipMemcpy(p->d_ops, ops, sizeof(apk_refine_op) * nops, hipMemcpyHostToDevice);
  if (e == hipSuccess && !chunks.empty()) {
    e = hipMalloc(&p->d_chunks, sizeof(apk_copy_chunk) * chunks.size());
    if (e == hipSuccess) e = hipMemcpy(p->d_chunks, chunks.data(), sizeof(apk_copy_chunk) * chunks.size(), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    apk_refine_plan_destroy(p);
    return set_err(ctx, APK_ERR_DEVICE, "apk_refine_plan_create", e);
  }
  *out = p;
  return APK_OK;
}

void apk_refine_plan_destroy(apk_refine_plan *p) {
  if (!p) return;
  if (p->d_ops) (void)hipFree(p->d_ops);
  if (p->d_chunks) (void)hipFree(p->d_chunks);
  delete p;
}

int apk_refine_plan_run(apk_ctx *ctx, const apk_refine_plan *p, apk_stream_t stream) {
  if (!ctx || !p) return set_err(ctx, APK_ERR_INVALID, "apk_refine_plan_run: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p->nchunks <= 0) return APK_OK;
  hipLaunchKernelGGL(refine_ops_kernel, dim3((unsigned)p->nchunks), dim3(256), 0, s, p->geom, p->nvar, p->d_ops, p->d_chunks);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? APK_OK : set_err(ctx, APK_ERR_DEVICE, "refine_ops launch", e);
}

int apk_flux_fix_plan_create(apk_ctx *ctx, const apk_flux_fix_region *regions, int n, apk_flux_fix_plan **out) {
  if (!ctx || !out || n < 0 || (n > 0 && !regions)) return set_err(ctx, APK_ERR_INVALID, "apk_flux_fix_plan_create: bad argument");
  *out = nullptr;
  apk_flux_fix_plan *p = new (std::nothrow) apk_flux_fix_plan();
  if (!p) return APK_ERR_INVALID;
  p->n = n;
  for (int q = 0; q < n; ++q) {
    const apk_flux_fix_region &r = regions[q];
    if (!r.fine_avg || !r.coarse_flux || !r.cons || r.nvar <= 0 || r.ext[0] <= 0 || r.ext[1] <= 0 || r.ext[2] <= 0 ||
        r.average < 0 || r.average > 3 || (r.average != 0 && (r.ndim < 1 || r.ndim > 3 || r.average > r.ndim || !(r.fine_area > 0.0)))) {
      delete p;
      return set_err(ctx, APK_ERR_INVALID, "apk_flux_fix_plan_create: bad region");
    }
    const int64_t items = (int64_t)r.ext[0] * r.ext[1] * r.ext[2] * r.nvar;
    p->max_items = items > p->max_items ? items : p->max_items;
  }
  if (n > 0) {
    hipError_t e = hipMalloc(&p->d_regions, sizeof(apk_flux_fix_region) * n);
    if (e == hipSuccess) e = hipMemcpy(p->d_regions, regions, sizeof(apk_flux_fix_region) * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      apk_flux_fix_plan_destroy(p);
      return set_err(ctx, APK_ERR_DEVICE, "apk_flux_fix_plan_create", e);
    }
  }
  *out = p;
  return APK_OK;
}

// The regions of up to three directions (direction 0's first, then 1's, then 2's: n_by_dir) as ONE plan that runs in one
// launch (flux_fix_merged_kernel).  field_base / block_elems locate a region's cons pointer in its block: the regions'
// destinations are cells of cell-shaped arrays [block][var][k][j][i] that start at field_base, block_elems apart, with
// the strides the regions carry (dst_stride[0] = 1).  Regions that share cells (the edges of a coarse block with finer
// blocks behind two or three of its faces) are linked, at most 8 per region; a mesh that needs more is refused and the
// caller keeps one plan per direction.
int apk_flux_fix_plan_create_merged(apk_ctx *ctx, const apk_flux_fix_region *regions, const int n_by_dir[3], const double *field_base,
                                    int64_t block_elems, apk_flux_fix_plan **out) {
  if (!ctx || !out || !n_by_dir || !field_base || block_elems <= 0)
    return set_err(ctx, APK_ERR_INVALID, "apk_flux_fix_plan_create_merged: bad argument");
  const int n = n_by_dir[0] + n_by_dir[1] + n_by_dir[2];
  int rc = apk_flux_fix_plan_create(ctx, regions, n, out);
  if (rc != APK_OK || n == 0) return rc;
  apk_flux_fix_plan *p = *out;
  if (p->max_items >= ((int64_t)1 << 31)) {
    apk_flux_fix_plan_destroy(p);
    *out = nullptr;
    return set_err(ctx, APK_ERR_UNSUPPORTED, "apk_flux_fix_plan_create_merged: a region of 2^31 items or more");
  }
  std::vector<FixRegionBox> boxes((size_t)n);
  std::vector<int64_t> blk((size_t)n);
  for (int q = 0; q < n; ++q) {
    const apk_flux_fix_region &r = regions[q];
    FixRegionBox &b = boxes[(size_t)q];
    b.dir = q < n_by_dir[0] ? 0 : (q < n_by_dir[0] + n_by_dir[1] ? 1 : 2);
    for (int m = 0; m < 8; ++m) b.partner[m] = FixRegionBox::Partner{-1, 0, {0, 0, 0}, {0, 0, 0}};
    // (cells of variable 0 of the block: offset = k * sk + j * sj + i)
    const int64_t off = r.cons - field_base;
    const int64_t sj = r.dst_stride[1], sk = r.dst_stride[2];
    const bool ok = off >= 0 && r.dst_stride[0] == 1 && sj > 0 && (sk > 0 ? sk % sj == 0 : r.ext[2] == 1);
    blk[(size_t)q] = ok ? off / block_elems : -1;
    int64_t in = ok ? off % block_elems : 0;
    b.lo[2] = sk > 0 ? (int)(in / sk) : 0;
    in -= (int64_t)b.lo[2] * (sk > 0 ? sk : 0);
    b.lo[1] = sj > 0 ? (int)(in / sj) : 0;
    b.lo[0] = (int)(in - (int64_t)b.lo[1] * sj);
    if (!ok) {
      apk_flux_fix_plan_destroy(p);
      *out = nullptr;
      return set_err(ctx, APK_ERR_UNSUPPORTED, "apk_flux_fix_plan_create_merged: region layout");
    }
  }
  // regions of one block, different directions, overlapping boxes
  std::vector<int> order((size_t)n);
  for (int q = 0; q < n; ++q) order[(size_t)q] = q;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return blk[(size_t)a] != blk[(size_t)b] ? blk[(size_t)a] < blk[(size_t)b] : a < b; });
  auto overlap = [&](int a, int b) {
    for (int d = 0; d < 3; ++d) {
      const int alo = boxes[(size_t)a].lo[d], ahi = alo + regions[a].ext[d] - 1;
      const int blo = boxes[(size_t)b].lo[d], bhi = blo + regions[b].ext[d] - 1;
      if (ahi < blo || bhi < alo) return false;
    }
    return true;
  };
  for (size_t x = 0; x < order.size();) {
    size_t y = x;
    while (y < order.size() && blk[(size_t)order[y]] == blk[(size_t)order[x]]) ++y;
    for (size_t a = x; a < y; ++a) {
      const int qa = order[a];
      int np = 0;
      // (ascending region index = ascending direction: the kernel meets the lower directions first)
      for (size_t bq = x; bq < y; ++bq) {
        const int qb = order[bq];
        if (qb == qa || boxes[(size_t)qb].dir == boxes[(size_t)qa].dir || !overlap(qa, qb)) continue;
        if (np == 8) {
          apk_flux_fix_plan_destroy(p);
          *out = nullptr;
          return set_err(ctx, APK_ERR_UNSUPPORTED, "apk_flux_fix_plan_create_merged: more than 8 regions share cells with one");
        }
        FixRegionBox::Partner &pp = boxes[(size_t)qa].partner[np++];
        pp.idx = qb;
        pp.dir = boxes[(size_t)qb].dir;
        for (int d = 0; d < 3; ++d) {
          pp.lo[d] = boxes[(size_t)qb].lo[d];
          pp.ext[d] = regions[qb].ext[d];
        }
      }
    }
    x = y;
  }
  hipError_t e = hipMalloc(&p->d_boxes, sizeof(FixRegionBox) * (size_t)n);
  if (e == hipSuccess) e = hipMemcpy(p->d_boxes, boxes.data(), sizeof(FixRegionBox) * (size_t)n, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    apk_flux_fix_plan_destroy(p);
    *out = nullptr;
    return set_err(ctx, APK_ERR_DEVICE, "apk_flux_fix_plan_create_merged", e);
  }
  return APK_OK;
}

void apk_flux_fix_plan_destroy(apk_flux_fix_plan *p) {
  if (!p) return;
  if (p->d_regions) (void)hipFree(p->d_regions);
  if (p->d_boxes) (void)hipFree(p->d_boxes);
  delete p;
}

int apk_flux_fix_plan_run(apk_ctx *ctx, const apk_flux_fix_plan *p, double beta_dt, int psi_var, double psi_factor,
                          apk_stream_t stream) {
  if (!ctx || !p) return set_err(ctx, APK_ERR_INVALID, "apk_flux_fix_plan_run: bad argument");
  if (p->n == 0) return APK_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int64_t gy = (p->max_items + 255) / 256;
  if (gy > 1024) gy = 1024;
  for (int off = 0; off < p->n; off += 1 << 20) {  // (gridDim.x is ample; chunked for symmetry with the copy plans)
    const int m = (p->n - off < (1 << 20)) ? p->n - off : (1 << 20);
    if (p->d_boxes) {  // (a merged plan is one launch: the partner indices are plan-wide)
      hipLaunchKernelGGL(flux_fix_merged_kernel, dim3((unsigned)p->n, (unsigned)gy), dim3(256), 0, s, p->d_regions, p->d_boxes, beta_dt,
                         psi_var, psi_factor);
      break;
    }
    hipLaunchKernelGGL(flux_fix_kernel, dim3((unsigned)m, (unsigned)gy), dim3(256), 0, s, p->d_regions + off, beta_dt, psi_var,
                       psi_factor);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? APK_OK : set_err(ctx, APK_ERR_DEVICE, "flux_fix launch", e);
}

// The launch half of apk_tag_blocks: reduce the criterion per block and request the read-back into
// pinned host memory; nothing is waited for.  *pending = 0 when there is nothing to read (1-D
// pressure gradient: "same" everywhere).
int apk_tag_blocks_begin(apk_ctx *ctx, const apk_pack *md, int criterion, int *pending, apk_stream_t stream) {
  return apk_tag_blocks_begin_skip(ctx, md, criterion, nullptr, pending, stream);
}

namespace {
// the per-block maxima' words and their pinned host copy, cleared for a new reduction
int tag_words_prepare(apk_ctx *ctx, int nb, hipStream_t s) {
  if (ctx->tagmax_cap < (size_t)nb) {
    if (ctx->d_tagmax) (void)hipFree(ctx->d_tagmax);
    ctx->d_tagmax = nullptr;
    ctx->tagmax_cap = 0;
    ctx->tag_words_clean = 0;
    APK_HIP_TRY(ctx, hipMalloc(&ctx->d_tagmax, sizeof(unsigned long long) * (nb + 64)));
    ctx->tagmax_cap = nb + 64;
  }
  if (ctx->h_partial_cap < (size_t)nb) {  // (pinned: a pageable destination costs a staging copy every cycle)
    if (ctx->h_partial) (void)hipHostFree(ctx->h_partial);
    ctx->h_partial = nullptr;
    ctx->h_partial_dev = nullptr;
    ctx->h_partial_cap = 0;
    APK_HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_partial), sizeof(double) * (nb + 64), hipHostMallocDefault));
    ctx->h_partial_cap = nb + 64;
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ctx->h_partial, 0) == hipSuccess) ctx->h_partial_dev = static_cast<double *>(dev);
    else (void)hipGetLastError();
  }
  // (the gather at the end of the previous cycle left the words at zero: apk_ctx::tag_words_clean)
  if (ctx->tag_words_clean < nb || ctx->clean_stream != s) APK_HIP_TRY(ctx, hipMemsetAsync(ctx->d_tagmax, 0, sizeof(unsigned long long) * nb, s));
  ctx->tag_words_clean = 0;
  return APK_OK;
}
// the read-back of a reduction just launched
int tag_words_request(apk_ctx *ctx, int nb, hipStream_t s, int *pending) {
  if (ctx->h_pinned_dev && ctx->h_partial_dev) {
    // the criteria ride to the host with the time-step word (apk_stage_dt_flags_read's gather kernel) if the caller
    // reads that next -- the driver does --, else apk_tag_blocks_end fetches them
    ctx->tags_pending = nb;
  } else {
    APK_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_partial, ctx->d_tagmax, sizeof(double) * nb, hipMemcpyDeviceToHost, s));
    ctx->tags_pending = 0;
  }
  *pending = 1;
  return APK_OK;
}
}  // namespace

int apk_tag_blocks_begin_skip(apk_ctx *ctx, const apk_pack *md, int criterion, const int *face_neighbor, int *pending,
                              apk_stream_t stream) {
  if (!ctx || !md || !pending || criterion < APK_TAG_PRESSURE_GRADIENT || criterion > APK_TAG_MAX_DENSITY)
    return set_err(ctx, APK_ERR_INVALID, "apk_tag_blocks: bad argument");
  const PackView &pv = md->view;
  if (pv.ng < 1) return set_err(ctx, APK_ERR_NGHOST, "apk_tag_blocks needs one ghost cell");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nb = pv.nblocks;
  const int ndim = (pv.nx3 > 1) ? 3 : ((pv.nx2 > 1) ? 2 : 1);
  *pending = 0;
  if (criterion == APK_TAG_PRESSURE_GRADIENT && ndim == 1) return APK_OK;  // gradient.cpp:56-58: AmrTag::same
  if (criterion == APK_TAG_VELOCITY_GRADIENT && ndim == 1)
    return set_err(ctx, APK_ERR_UNSUPPORTED, "xyvelocity_gradient needs at least two dimensions");
  const int rc = tag_words_prepare(ctx, nb, s);
  if (rc != APK_OK) return rc;
  unsigned long long *d_max = ctx->d_tagmax;
  const int kchunks = (pv.nx3 >= 12 && nb < 4096) ? 3 : 1;
  const dim3 grid = rect_grid(pv.nx1 + 2, pv.nx2 + 2, nb * kchunks), block(64, 4, 1);
  if (criterion == APK_TAG_PRESSURE_GRADIENT)
    hipLaunchKernelGGL(tag_kernel<APK_TAG_PRESSURE_GRADIENT>, grid, block, 0, s, pv, d_max, kchunks, face_neighbor);
  else if (criterion == APK_TAG_VELOCITY_GRADIENT)
    hipLaunchKernelGGL(tag_kernel<APK_TAG_VELOCITY_GRADIENT>, grid, block, 0, s, pv, d_max, kchunks, face_neighbor);
  else
    hipLaunchKernelGGL(tag_kernel<APK_TAG_MAX_DENSITY>, grid, block, 0, s, pv, d_max, kchunks, face_neighbor);
  return tag_words_request(ctx, nb, s, pending);
}

int apk_tag_blocks_dt_from_cons(apk_ctx *ctx, const apk_pack *md, int fluid, const apk_eos *eos, const int *face_neighbor, int *pending,
                                apk_stream_t stream) {
  if (!ctx || !md || !pending || !eos || (fluid != APK_FLUID_EULER && fluid != APK_FLUID_GLMMHD) ||
      md->view.nhydro != ((fluid == APK_FLUID_EULER) ? 5 : 9))
    return set_err(ctx, APK_ERR_INVALID, "apk_tag_blocks_dt_from_cons: bad argument");
  const PackView &pv = md->view;
  // (the lean ConsToPrim in registers, nothing written back; the pressure tile of a few planes in LDS)
  if (!eos_is_lean(*eos) || eos->dfloor > 0.0 || eos->efloor > 0.0 || pv.nvar != pv.nhydro || apk::tag_from_cons_kchunks(pv) <= 0)
    return set_err(ctx, APK_ERR_UNSUPPORTED, "apk_tag_blocks_dt_from_cons: 3-D packs of blocks whose planes fit the LDS, two ghost layers, no floors / ceilings, no passive scalars");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nb = pv.nblocks;
  *pending = 0;
  int rc = tag_words_prepare(ctx, nb, s);
  if (rc != APK_OK) return rc;
  if (apk::prepare_dt_word(ctx, s) != APK_OK) return set_err(ctx, APK_ERR_DEVICE, "time-step word reset", hipGetLastError());
  rc = apk::launch_tag_pgrad_from_cons(pv, fluid, *eos, ctx->d_flags, ctx->d_u64 + 4, ctx->d_tagmax, face_neighbor, s);
  if (rc != APK_OK) return set_err(ctx, rc, "tag_pgrad_from_cons kernel launch failed", hipGetLastError());
  return tag_words_request(ctx, nb, s, pending);
}

// The read half: waits for the stream (a no-op if the caller synchronised it meanwhile, e.g. by reading
// the time-step estimate enqueued after apk_tag_blocks_begin) and turns the criteria into tags.
int apk_tag_blocks_end(apk_ctx *ctx, int nblocks, int criterion, int pending, double p0, double p1, int *tags, double *crit,
                       apk_stream_t stream) {
  if (!ctx || !tags || nblocks < 0) return set_err(ctx, APK_ERR_INVALID, "apk_tag_blocks: bad argument");
  if (!pending) {
    for (int b = 0; b < nblocks; ++b) {
      tags[b] = 0;
      if (crit) crit[b] = 0.0;
    }
    return APK_OK;
  }
  if ((size_t)nblocks > ctx->h_partial_cap) return set_err(ctx, APK_ERR_INVALID, "apk_tag_blocks_end without apk_tag_blocks_begin");
  if (ctx->tags_pending > 0) {  // (nobody gathered them meanwhile)
    APK_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_partial, ctx->d_tagmax, sizeof(double) * ctx->tags_pending, hipMemcpyDeviceToHost,
                                    reinterpret_cast<hipStream_t>(stream)));
    ctx->tags_pending = 0;
  }
  APK_HIP_TRY(ctx, hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  const double *h = ctx->h_partial;
  const double refine_above = p0;
  const double deref_below = (criterion == APK_TAG_PRESSURE_GRADIENT) ? 0.25 * p0
                             : (criterion == APK_TAG_VELOCITY_GRADIENT) ? 0.5 * p0 : p1;
  for (int b = 0; b < nblocks; ++b) {
    tags[b] = (h[b] > refine_above) ? 1 : ((h[b] < deref_below) ? -1 : 0);
    if (crit) crit[b] = h[b];
  }
  return APK_OK;
}

int apk_tag_blocks(apk_ctx *ctx, const apk_pack *md, int criterion, double p0, double p1, int *tags, double *crit,
                   apk_stream_t stream) {
  if (!tags) return set_err(ctx, APK_ERR_INVALID, "apk_tag_blocks: bad argument");
  int pending = 0;
  const int rc = apk_tag_blocks_begin(ctx, md, criterion, &pending, stream);
  if (rc != APK_OK) return rc;
  return apk_tag_blocks_end(ctx, md->view.nblocks, criterion, pending, p0, p1, tags, crit, stream);
}

}  // extern "C"
