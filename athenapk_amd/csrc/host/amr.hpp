// amr.hpp -- block-structured mesh refinement on the host: the forest of octrees (quadtrees in
// 2-D) over the root grid of meshblocks, 2:1 balance, neighbour classification, and the index-box
// plans of everything that moves data between blocks of different levels (pure host logic; no HIP).
// Stands in for the parts of Parthenon's Mesh / MeshBlockTree / bvals-in-one / flux correction
// that BASELINE config 5 (inputs/blast_3d_amr.in) needs; Parthenon is un-vendored upstream, so the
// scheme below is a restatement of its published behaviour:
//   * every block owns a coarse buffer (nx/2 interior cells + cng ghosts) holding its restricted
//     interior;
//   * ghost zones facing a same-level block are copied, those facing finer blocks are copied from
//     the finer blocks' coarse buffers (= RestrictAverage of their interiors), those facing a
//     coarser block are prolongated (ProlongateCellMinModMultiD, custom_ops.hpp:49-186) from the
//     block's own coarse buffer, whose ghost zones are filled from the coarser block's interior,
//     from same-level neighbours' coarse buffers and by the physical boundary conditions;
//   * at a coarse-fine face the coarse block's face flux is replaced by the area average of the
//     fine fluxes (hydro_driver.cpp:527-531) before the flux divergence is taken.
#pragma once

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "mesh.hpp"

namespace apk {

enum AmrRegionKind {  // extends RegionKind: which array of a block a plan entry addresses
  RK_COARSE = 3,      // the block's coarse buffer
  RK_FLUX1 = 4, RK_FLUX2 = 5, RK_FLUX3 = 6,
  RK_OLD_BLOCK = 7,   // regridding: cons / coarse buffer of the mesh being replaced
  RK_OLD_COARSE = 8
};
enum AmrNeighborKind { NB_PHYSICAL = 0, NB_SAME = 1, NB_COARSER = 2, NB_FINER = 3 };

struct AmrLeaf {
  int level = 0;
  int lx[3] = {0, 0, 0};  // logical location at its level
  int deref_count = 0;    // consecutive derefinement requests (parthenon/mesh/derefine_count)
};

struct AmrRefOp {  // one apk_refine_op in terms of block indices
  int kind = 0, level = 0;
  int src_kind = 0, src_block = 0, dst_kind = 0, dst_block = 0;
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  int geom_block = 0;  // block whose lower corner / cell widths the operator uses
  int corner = 0;      // prolongation into a ghost zone behind an edge or a corner (see BoxRegion::corner)
};

struct AmrPlans {
  std::vector<AmrRefOp> restrict_own;        // interior -> coarse buffer, every block of level >= 1
  std::vector<BoxRegion> fill;               // all copies into fine / coarse-buffer ghost zones
  std::vector<BoxRegion> coarse_bc[3];       // physical boundaries of the coarse buffers
  std::vector<AmrRefOp> prolongate;          // coarse buffer -> ghost zones facing coarser blocks
  std::vector<BoxRegion> fine_bc[3];         // physical boundaries of the blocks
  std::vector<AmrRefOp> flux_restrict[3];    // fine face flux -> coarse buffer plane
  std::vector<BoxRegion> flux_copy[3];       // coarse buffer plane -> coarse block's face flux
};

struct AmrTree {
  int nrb[3] = {1, 1, 1};  // root grid of meshblocks
  bool act[3] = {true, false, false};
  int ndim = 1, max_level = 0;
  int bc_in[3] = {0, 0, 0}, bc_out[3] = {0, 0, 0};
  // ground truth: the set of leaves and of internal nodes; `leaves` (Z-order) and `index` are
  // rebuilt from it by Reindex() after the tree has been modified
  std::unordered_map<uint64_t, AmrLeaf> leafmap;
  std::unordered_set<uint64_t> internal;
  std::vector<AmrLeaf> leaves;
  std::unordered_map<uint64_t, int> index;
  long long modifications = 0;  // blocks split or merged so far (an unchanged count = an unchanged forest)

  static uint64_t Key(int level, const int lx[3]) {
    return ((uint64_t)level << 57) | ((uint64_t)lx[2] << 38) | ((uint64_t)lx[1] << 19) | (uint64_t)lx[0];
  }
  int NumChildren() const { return 1 << ndim; }

  void InitRoot() {
    leafmap.clear();
    internal.clear();
    for (int z = 0; z < nrb[2]; ++z)
      for (int y = 0; y < nrb[1]; ++y)
        for (int x = 0; x < nrb[0]; ++x) {
          AmrLeaf l;
          l.lx[0] = x, l.lx[1] = y, l.lx[2] = z;
          leafmap[Key(0, l.lx)] = l;
        }
    Reindex();
  }

  // depth-first Z-order of the forest = Morton order of the block origins at the finest level
  void Reindex() {
    leaves.clear();
    for (const auto &kv : leafmap) leaves.push_back(kv.second);
    auto zkey = [&](const AmrLeaf &l) {
      const int sh = max_level - l.level;
      return Mesh::Morton((unsigned)(l.lx[0] << sh), (unsigned)(l.lx[1] << sh), (unsigned)(l.lx[2] << sh));
    };
    std::sort(leaves.begin(), leaves.end(), [&](const AmrLeaf &a, const AmrLeaf &b) { return zkey(a) < zkey(b); });
    index.clear();
    for (int n = 0; n < (int)leaves.size(); ++n) index[Key(leaves[n].level, leaves[n].lx)] = n;
  }
  void SetDerefCount(int lb, int n) {
    leaves[lb].deref_count = n;
    leafmap[Key(leaves[lb].level, leaves[lb].lx)].deref_count = n;
  }

  // periodic wrap of a block position at `level`; false beyond a non-periodic boundary
  bool Wrap(int level, const int pos[3], int out[3]) const {
    for (int d = 0; d < 3; ++d) {
      const int n = act[d] ? (nrb[d] << level) : 1;
      int c = pos[d];
      if (c < 0) {
        if (bc_in[d] != BC_PERIODIC) return false;
        c += n;
      } else if (c >= n) {
        if (bc_out[d] != BC_PERIODIC) return false;
        c -= n;
      }
      out[d] = c;
    }
    return true;
  }

  // who covers the block-sized slot at (level, pos)?  NB_SAME / NB_COARSER: *leaf = its index
  int Classify(int level, const int pos[3], int *leaf) const {
    int w[3];
    *leaf = -1;
    if (!Wrap(level, pos, w)) return NB_PHYSICAL;
    auto it = index.find(Key(level, w));
    if (it != index.end()) {
      *leaf = it->second;
      return NB_SAME;
    }
    if (internal.count(Key(level, w))) return NB_FINER;
    if (level > 0) {
      const int p[3] = {w[0] >> 1, w[1] >> 1, w[2] >> 1};
      it = index.find(Key(level - 1, p));
      if (it != index.end()) {
        *leaf = it->second;
        return NB_COARSER;
      }
    }
    throw std::runtime_error("mesh tree violates the 2:1 level balance");
  }

  template <class F>
  void ForEachOffset(F &&f) const {
    for (int oz = -1; oz <= 1; ++oz)
      for (int oy = -1; oy <= 1; ++oy)
        for (int ox = -1; ox <= 1; ++ox) {
          if (ox == 0 && oy == 0 && oz == 0) continue;
          if ((!act[1] && oy != 0) || (!act[2] && oz != 0)) continue;
          const int o[3] = {ox, oy, oz};
          f(o);
        }
  }
  template <class F>
  void ForEachChild(const int plx[3], F &&f) const {
    for (int cz = 0; cz <= (act[2] ? 1 : 0); ++cz)
      for (int cy = 0; cy <= (act[1] ? 1 : 0); ++cy)
        for (int cx = 0; cx <= 1; ++cx) {
          const int c[3] = {cx, cy, cz};
          const int cl[3] = {2 * plx[0] + cx, act[1] ? 2 * plx[1] + cy : 0, act[2] ? 2 * plx[2] + cz : 0};
          f(c, cl);
        }
  }

  // refine (level, lx) and, first, every coarser neighbour (recursively): keeps the 2:1 balance.
  // `leaves` / `index` are stale until the next Reindex()
  void RefineBalanced(int level, const int lx[3]) {
    const uint64_t key = Key(level, lx);
    if (!leafmap.count(key)) return;  // already split
    if (level >= max_level) return;
    ForEachOffset([&](const int o[3]) {
      const int pos[3] = {lx[0] + o[0], lx[1] + o[1], lx[2] + o[2]};
      int w[3];
      if (!Wrap(level, pos, w)) return;
      if (leafmap.count(Key(level, w)) || internal.count(Key(level, w))) return;
      if (level > 0) {
        const int p[3] = {w[0] >> 1, w[1] >> 1, w[2] >> 1};
        RefineBalanced(level - 1, p);
      }
    });
    leafmap.erase(key);
    internal.insert(key);
    ++modifications;
    ForEachChild(lx, [&](const int *, const int cl[3]) {
      AmrLeaf c;
      c.level = level + 1;
      for (int d = 0; d < 3; ++d) c.lx[d] = cl[d];
      leafmap[Key(c.level, c.lx)] = c;
    });
  }

  // are all children of (level, plx) leaves, and can they be merged without breaking the balance?
  bool CanMerge(int level, const int plx[3]) const {
    if (!internal.count(Key(level, plx))) return false;
    bool ok = true;
    ForEachChild(plx, [&](const int *, const int cl[3]) {
      if (!leafmap.count(Key(level + 1, cl))) ok = false;
    });
    ForEachOffset([&](const int o[3]) {
      const int pos[3] = {plx[0] + o[0], plx[1] + o[1], plx[2] + o[2]};
      int w[3];
      if (!ok || !Wrap(level, pos, w)) return;
      if (!internal.count(Key(level, w))) return;
      // the children of that node which touch the merged block must be leaves
      ForEachChild(w, [&](const int c[3], const int cl[3]) {
        for (int d = 0; d < 3; ++d)
          if (act[d] && o[d] != 0 && c[d] != (o[d] < 0 ? 1 : 0)) return;
        if (internal.count(Key(level + 1, cl))) ok = false;
      });
    });
    return ok;
  }

  void Merge(int level, const int plx[3]) {
    ++modifications;
    internal.erase(Key(level, plx));
    ForEachChild(plx, [&](const int *, const int cl[3]) { leafmap.erase(Key(level + 1, cl)); });
    AmrLeaf p;
    p.level = level;
    for (int d = 0; d < 3; ++d) p.lx[d] = plx[d];
    leafmap[Key(level, plx)] = p;
  }
};

// ---- index-box plans ----------------------------------------------------------------------------
struct AmrGeom {
  int mb[3] = {1, 1, 1}, ng = 2, cng = 2, nvar = 5;
  bool act[3] = {true, false, false};
  // fine arrays
  int fn[3] = {1, 1, 1}, fs[3] = {0, 0, 0}, fe[3] = {0, 0, 0};
  int64_t fst[4] = {1, 1, 1, 1};
  // coarse buffers (cell-centred use)
  int cn[3] = {1, 1, 1}, cs[3] = {0, 0, 0}, ce[3] = {0, 0, 0};
  int64_t cst[4] = {1, 1, 1, 1};
  int64_t coarse_doubles = 0;  // allocation per block: room for one more entry per dimension

  void Build() {
    for (int d = 0; d < 3; ++d) {
      fn[d] = act[d] ? mb[d] + 2 * ng : 1;
      fs[d] = act[d] ? ng : 0;
      fe[d] = fs[d] + mb[d] - 1;
      cn[d] = act[d] ? mb[d] / 2 + 2 * cng : 1;
      cs[d] = act[d] ? cng : 0;
      ce[d] = act[d] ? cs[d] + mb[d] / 2 - 1 : 0;
    }
    fst[0] = 1, fst[1] = fn[0], fst[2] = (int64_t)fn[0] * fn[1], fst[3] = fst[2] * fn[2];
    cst[0] = 1, cst[1] = cn[0], cst[2] = (int64_t)cn[0] * cn[1], cst[3] = cst[2] * cn[2];
    coarse_doubles = (int64_t)nvar * (cn[0] + 1) * (cn[1] + (act[1] ? 1 : 0)) * (cn[2] + (act[2] ? 1 : 0));
  }
  // strides of the coarse buffer when it holds the restricted flux of direction dir (same extents as
  // the cell-centred use: face index ce+1 lies in the ghost layer)
};

inline void amr_box_region(BoxRegion &r, const int64_t sst[4], const int slo[3], const int64_t dst_st[4], const int dlo[3],
                           const int ext[3], int nvar) {
  r.nvar = nvar;
  r.src_off = r.dst_off = 0;
  for (int d = 0; d < 3; ++d) {
    r.ext[d] = ext[d];
    r.src_off += slo[d] * sst[d];
    r.dst_off += dlo[d] * dst_st[d];
  }
  for (int q = 0; q < 4; ++q) {
    r.src_stride[q] = sst[q];
    r.dst_stride[q] = dst_st[q];
  }
}

// physical-boundary regions of one array family (fine blocks or coarse buffers)
inline void amr_bc_regions(const AmrTree &t, int lb, int kind, const int n[3], const int s3[3], const int e3[3], int nghost,
                           const int64_t st[4], int nvar, std::vector<BoxRegion> out[3]) {
  const AmrLeaf &l = t.leaves[lb];
  for (int d = 0; d < 3; ++d) {
    if (!t.act[d]) continue;
    for (int side = 0; side < 2; ++side) {
      const bool edge = side ? (l.lx[d] == (t.nrb[d] << l.level) - 1) : (l.lx[d] == 0);
      const int bk = side ? t.bc_out[d] : t.bc_in[d];
      if (!edge || bk == BC_PERIODIC) continue;
      BoxRegion r;
      r.src_kind = r.dst_kind = kind;
      r.src_block = r.dst_block = lb;
      r.nvar = nvar;
      for (int q = 0; q < 4; ++q) r.src_stride[q] = r.dst_stride[q] = st[q];
      for (int q = 0; q < 3; ++q) r.ext[q] = n[q];
      r.ext[d] = nghost;
      const int g0 = side ? e3[d] + 1 : s3[d] - nghost;
      r.dst_off = g0 * st[d];
      if (bk == BC_OUTFLOW) {
        r.src_off = (side ? e3[d] : s3[d]) * st[d];
        r.src_stride[d] = 0;
      } else {  // reflecting: ghost g mirrors 2 s - 1 - g (inner) / 2 e + 1 - g (outer)
        const int first = side ? (2 * e3[d] + 1 - g0) : (2 * s3[d] - 1 - g0);
        r.src_off = first * st[d];
        r.src_stride[d] = -st[d];
        r.flip_var = 1 + d;
      }
      out[d].push_back(r);
    }
  }
}

// fill_depth > 0: the fills of the blocks' own ghost zones (copies, restricted fine data, prolongation, physical
// boundaries) reach only that many layers deep -- the "shell" exchange that precedes a refinement check when the
// first stage of the next cycle needs no more (the gradient criteria read 2 layers, edges and corners included).
// Coarse-buffer fills are always complete.  Even, at most nghost.
inline void BuildAmrPlans(const AmrTree &t, const AmrGeom &g, AmrPlans &p, int fill_depth = 0) {
  p = AmrPlans();

  const int nb = (int)t.leaves.size();
  for (int d = 0; d < 3; ++d)
    if (g.act[d] && (g.mb[d] % 2 != 0 || g.mb[d] / 2 < g.cng || g.mb[d] / 2 < g.ng))
      throw std::runtime_error("mesh refinement needs even meshblock sizes of at least 2 * nghost cells per dimension");
  if (g.ng % 2 != 0) throw std::runtime_error("mesh refinement needs an even number of ghost cells (use nghost = 4 with ppm / wenoz)");
  const int gd = (fill_depth > 0 && fill_depth < g.ng) ? fill_depth : g.ng;
  if (gd % 2 != 0) throw std::runtime_error("the ghost fill depth must be even");
  // pass 1: which blocks face a coarser one
  std::vector<char> has_coarser(nb, 0);
  for (int lb = 0; lb < nb; ++lb) {
    const AmrLeaf &l = t.leaves[lb];
    t.ForEachOffset([&](const int o[3]) {
      const int pos[3] = {l.lx[0] + o[0], l.lx[1] + o[1], l.lx[2] + o[2]};
      int leaf;
      if (t.Classify(l.level, pos, &leaf) == NB_COARSER) has_coarser[lb] = 1;
    });
  }
  for (int lb = 0; lb < nb; ++lb) {
    const AmrLeaf &l = t.leaves[lb];
    if (l.level >= 1) {
      AmrRefOp op;
      op.kind = 1;  // APK_RO_RESTRICT_CELL
      op.level = l.level;
      op.src_kind = RK_BLOCK, op.src_block = lb, op.dst_kind = RK_COARSE, op.dst_block = lb, op.geom_block = lb;
      for (int d = 0; d < 3; ++d) op.lo[d] = g.cs[d], op.hi[d] = g.ce[d];
      p.restrict_own.push_back(op);
    }
    t.ForEachOffset([&](const int o[3]) {
      const int pos[3] = {l.lx[0] + o[0], l.lx[1] + o[1], l.lx[2] + o[2]};
      int nbr;
      const int kind = t.Classify(l.level, pos, &nbr);
      if (kind == NB_PHYSICAL) return;
      int slo[3], dlo[3], ext[3];
      // The unsplit sweeps and the flux correction read ghost cells straight behind a FACE only: whatever
      // fills the block's ghost zone behind an edge or a corner is marked, and the stage loop runs the plans
      // without those boxes (coarse-buffer fills are always complete: the prolongation stencil reaches
      // sideways).  The tagging criteria DO read behind edges and corners (gradient.cpp:33-36): a cycle that
      // ends with a refinement check exchanges in full after its last stage (regrid_check_follows).
      const int behind_corner = ((o[0] != 0) + (o[1] != 0) + (o[2] != 0)) != 1;
      if (kind == NB_SAME) {
        // fine ghost zone <- neighbour interior
        for (int d = 0; d < 3; ++d) {
          if (!g.act[d] || o[d] == 0) {
            dlo[d] = slo[d] = g.fs[d], ext[d] = g.act[d] ? g.mb[d] : 1;
          } else {
            ext[d] = gd;
            dlo[d] = (o[d] < 0) ? g.fs[d] - gd : g.fe[d] + 1;
            slo[d] = (o[d] < 0) ? g.fe[d] - gd + 1 : g.fs[d];
          }
        }
        BoxRegion r;
        r.src_kind = RK_BLOCK, r.src_block = nbr, r.dst_kind = RK_BLOCK, r.dst_block = lb;
        amr_box_region(r, g.fst, slo, g.fst, dlo, ext, g.nvar);
        r.corner = behind_corner;
        r.same_face = !behind_corner;
        p.fill.push_back(r);
        if (has_coarser[lb]) {  // coarse-buffer ghost zone <- neighbour's restricted interior
          for (int d = 0; d < 3; ++d) {
            if (!g.act[d] || o[d] == 0) {
              dlo[d] = slo[d] = g.cs[d], ext[d] = g.act[d] ? g.mb[d] / 2 : 1;
            } else {
              ext[d] = g.cng;
              dlo[d] = (o[d] < 0) ? g.cs[d] - g.cng : g.ce[d] + 1;
              slo[d] = (o[d] < 0) ? g.ce[d] - g.cng + 1 : g.cs[d];
            }
          }
          BoxRegion c;
          c.src_kind = RK_COARSE, c.src_block = nbr, c.dst_kind = RK_COARSE, c.dst_block = lb;
          amr_box_region(c, g.cst, slo, g.cst, dlo, ext, g.nvar);
          p.fill.push_back(c);
        }
      } else if (kind == NB_FINER) {
        int w[3];
        t.Wrap(l.level, pos, w);
        int nface = 0, fdir = 0;
        for (int d = 0; d < 3; ++d)
          if (o[d] != 0) ++nface, fdir = d;
        for (int cz = 0; cz <= (t.act[2] ? 1 : 0); ++cz)
          for (int cy = 0; cy <= (t.act[1] ? 1 : 0); ++cy)
            for (int cx = 0; cx <= 1; ++cx) {
              const int c[3] = {cx, cy, cz};
              bool touches = true;
              for (int d = 0; d < 3; ++d)
                if (t.act[d] && o[d] != 0 && c[d] != (o[d] < 0 ? 1 : 0)) touches = false;
              if (!touches) continue;
              const int cl[3] = {2 * w[0] + cx, t.act[1] ? 2 * w[1] + cy : 0, t.act[2] ? 2 * w[2] + cz : 0};
              auto it = t.index.find(AmrTree::Key(l.level + 1, cl));
              if (it == t.index.end()) throw std::runtime_error("mesh tree violates the 2:1 level balance");
              const int fb = it->second;
              // my ghost zone (the half / quarter this child covers) <- its restricted interior
              for (int d = 0; d < 3; ++d) {
                if (!g.act[d]) {
                  dlo[d] = slo[d] = 0, ext[d] = 1;
                } else if (o[d] == 0) {
                  ext[d] = g.mb[d] / 2;
                  dlo[d] = g.fs[d] + c[d] * (g.mb[d] / 2);
                  slo[d] = g.cs[d];
                } else {
                  ext[d] = gd;
                  dlo[d] = (o[d] < 0) ? g.fs[d] - gd : g.fe[d] + 1;
                  slo[d] = (o[d] < 0) ? g.ce[d] - gd + 1 : g.cs[d];
                }
              }
              BoxRegion r;
              r.src_kind = RK_COARSE, r.src_block = fb, r.dst_kind = RK_BLOCK, r.dst_block = lb;
              amr_box_region(r, g.cst, slo, g.fst, dlo, ext, g.nvar);
              r.corner = behind_corner;
              p.fill.push_back(r);
              if (nface == 1) {  // flux correction across the shared face
                const int d = fdir;
                AmrRefOp op;
                op.kind = 5 + d;  // APK_RO_RESTRICT_FLUX1 + d
                op.level = l.level + 1;
                op.src_kind = RK_FLUX1 + d, op.src_block = fb, op.dst_kind = RK_COARSE, op.dst_block = fb, op.geom_block = fb;
                for (int q = 0; q < 3; ++q) op.lo[q] = g.cs[q], op.hi[q] = g.ce[q];
                op.lo[d] = op.hi[d] = (o[d] < 0) ? g.ce[d] + 1 : g.cs[d];  // the child's face towards me
                p.flux_restrict[d].push_back(op);
                for (int q = 0; q < 3; ++q) {
                  if (!g.act[q]) {
                    dlo[q] = slo[q] = 0, ext[q] = 1;
                  } else if (q == d) {
                    ext[q] = 1;
                    slo[q] = op.lo[d];
                    dlo[q] = (o[d] < 0) ? g.fs[d] : g.fe[d] + 1;
                  } else {
                    ext[q] = g.mb[q] / 2;
                    slo[q] = g.cs[q];
                    dlo[q] = g.fs[q] + c[q] * (g.mb[q] / 2);
                  }
                }
                BoxRegion fr;
                fr.src_kind = RK_COARSE, fr.src_block = fb, fr.dst_kind = RK_FLUX1 + d, fr.dst_block = lb;
                amr_box_region(fr, g.cst, slo, g.fst, dlo, ext, g.nvar);
                p.flux_copy[d].push_back(fr);
              }
            }
      } else {  // NB_COARSER
        // coarse-buffer ghost zone <- the coarser block's interior.  Global coarse cell index of my
        // coarse-buffer cell c along d: lx*mb/2 + (c - cs); the coarser block sits (unwrapped) at
        // floor(pos / 2) and starts at that times mb
        for (int d = 0; d < 3; ++d) {
          if (!g.act[d]) {
            dlo[d] = slo[d] = 0, ext[d] = 1;
            continue;
          }
          if (o[d] == 0) {
            dlo[d] = g.cs[d], ext[d] = g.mb[d] / 2;
          } else {
            ext[d] = g.cng;
            dlo[d] = (o[d] < 0) ? g.cs[d] - g.cng : g.ce[d] + 1;
          }
          const int64_t gcell = (int64_t)l.lx[d] * (g.mb[d] / 2) + (dlo[d] - g.cs[d]);
          const int ppos = (pos[d] >= 0) ? pos[d] / 2 : -((-pos[d] + 1) / 2);  // floor division
          slo[d] = (int)(gcell - (int64_t)ppos * g.mb[d]) + g.fs[d];
          if (slo[d] < g.fs[d] || slo[d] + ext[d] - 1 > g.fe[d]) throw std::runtime_error("coarse neighbour box outside its interior");
        }
        BoxRegion r;
        r.src_kind = RK_BLOCK, r.src_block = nbr, r.dst_kind = RK_COARSE, r.dst_block = lb;
        amr_box_region(r, g.fst, slo, g.cst, dlo, ext, g.nvar);
        p.fill.push_back(r);
        AmrRefOp op;
        op.kind = 0;  // APK_RO_PROLONGATE
        op.level = l.level;
        op.src_kind = RK_COARSE, op.src_block = lb, op.dst_kind = RK_BLOCK, op.dst_block = lb, op.geom_block = lb;
        for (int d = 0; d < 3; ++d) {
          if (!g.act[d] || o[d] == 0) {
            op.lo[d] = g.cs[d], op.hi[d] = g.ce[d];
          } else if (o[d] < 0) {
            op.lo[d] = g.cs[d] - gd / 2, op.hi[d] = g.cs[d] - 1;
          } else {
            op.lo[d] = g.ce[d] + 1, op.hi[d] = g.ce[d] + gd / 2;
          }
        }
        op.corner = behind_corner;
        p.prolongate.push_back(op);
      }
    });
    amr_bc_regions(t, lb, RK_BLOCK, g.fn, g.fs, g.fe, gd, g.fst, g.nvar, p.fine_bc);
    if (has_coarser[lb]) amr_bc_regions(t, lb, RK_COARSE, g.cn, g.cs, g.ce, g.cng, g.cst, g.nvar, p.coarse_bc);
  }
}


// ---- distribution over ranks ----------------------------------------------------------------------
// Contiguous ranges of the Z-ordered leaf list, equal block counts to within one (all blocks of a
// refined mesh have the same shape, hence the same cost): rank r owns [first[r], first[r + 1]).
struct AmrPartition {
  std::vector<int> first;
  void Build(int nblocks, int nranks) {
    first.resize(nranks + 1);
    for (int r = 0; r <= nranks; ++r) first[r] = (int)(((int64_t)r * nblocks) / nranks);
  }
  int Owner(int g) const { return (int)(std::upper_bound(first.begin(), first.end(), g) - first.begin()) - 1; }
  int Local(int g) const { return g - first[Owner(g)]; }
  int Count(int r) const { return first[r + 1] - first[r]; }
};

struct AmrMessages {  // one contiguous message per peer and direction
  std::vector<PeerPlan> peers;  // sorted by rank
  int PeerIndex(int rank) {
    for (int p = 0; p < (int)peers.size(); ++p)
      if (peers[p].rank == rank) return p;
    PeerPlan pp;
    pp.rank = rank;
    auto it = std::lower_bound(peers.begin(), peers.end(), rank, [](const PeerPlan &a, int r) { return a.rank < r; });
    return (int)(peers.insert(it, pp) - peers.begin());
  }
};

// Split a list of box copies between (globally numbered) blocks into this rank's share: copies
// between two of my blocks stay copies (with local block numbers), a copy out of one of my blocks into
// another rank's becomes a pack into that peer's send buffer, the reverse an unpack.  Every rank
// walks the same global list in the same order, so sender and receiver lay a message out alike.
// Pass 1 (register_peers) must run over ALL lists that share a message set before pass 2, because
// peer indices shift while peers are being added.
inline void AmrRegisterPeers(const std::vector<BoxRegion> &global, const AmrPartition &src_part, const AmrPartition &dst_part,
                             int rank, AmrMessages &msgs) {
  for (const BoxRegion &r : global) {
    const int so = src_part.Owner(r.src_block), to = dst_part.Owner(r.dst_block);
    if (so == to) continue;
    if (so == rank) msgs.PeerIndex(to);
    if (to == rank) msgs.PeerIndex(so);
  }
}
inline void AmrLocalize(const std::vector<BoxRegion> &global, const AmrPartition &src_part, const AmrPartition &dst_part, int rank,
                        AmrMessages &msgs, std::vector<BoxRegion> &local, std::vector<BoxRegion> &pack,
                        std::vector<BoxRegion> &unpack) {
  for (const BoxRegion &g : global) {
    const int so = src_part.Owner(g.src_block), to = dst_part.Owner(g.dst_block);
    if (so != rank && to != rank) continue;
    BoxRegion r = g;
    const int64_t count = (int64_t)g.ext[0] * g.ext[1] * g.ext[2] * g.nvar;
    if (so == rank) r.src_block = g.src_block - src_part.first[rank];
    if (to == rank) r.dst_block = g.dst_block - dst_part.first[rank];
    if (so == to) {
      local.push_back(r);
    } else if (so == rank) {
      const int p = msgs.PeerIndex(to);
      r.dst_kind = RK_SEND;
      r.dst_block = p;
      r.dst_off = msgs.peers[p].send_count;
      Mesh::Compact(g.ext, r.dst_stride);
      r.flip_var = -1;  // (boundary conditions are never remote)
      msgs.peers[p].send_count += count;
      pack.push_back(r);
    } else {
      const int p = msgs.PeerIndex(so);
      r.src_kind = RK_RECV;
      r.src_block = p;
      r.src_off = msgs.peers[p].recv_count;
      Mesh::Compact(g.ext, r.src_stride);
      msgs.peers[p].recv_count += count;
      unpack.push_back(r);
    }
  }
}

}  // namespace apk
