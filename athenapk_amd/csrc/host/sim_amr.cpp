// sim_amr.cpp -- refined meshes in the standalone host driver: forest set-up from the deck, this
// rank's share of the multilevel plans, their device forms, exchange, flux correction, regridding.
#include "sim_internal.hpp"

using namespace apk;

namespace apk {
namespace host {

// ---- mesh refinement: tree set-up from the deck (parthenon/mesh/refinement, numlevel,
// derefine_count, <parthenon/static_refinement#> blocks) ------------------------------------------
// cell width on a refinement level (collapsed dimensions are not refined)
double level_dx(const apk_sim *s, int level, int d) { return s->mesh.Active(d) ? s->dx[d] / (double)(1 << level) : s->dx[d]; }
// the leaf behind local block lb (this rank owns a contiguous range of the Z-ordered leaf list)
const AmrLeaf &amr_leaf(const apk_sim *s, int lb) { return s->amr->leaves[s->amr_part.first[s->rank] + lb]; }
int block_level(const apk_sim *s, int lb) { return s->amr ? amr_leaf(s, lb).level : 0; }

// refresh the uniform-mesh bookkeeping the rest of the driver reads (block counts, ids) from the tree
void amr_sync_mesh(apk_sim *s) {
  Mesh &m = s->mesh;
  const int n = (int)s->amr->leaves.size();
  if (n < s->nranks) throw std::runtime_error("fewer meshblocks than ranks");
  s->amr_part.Build(n, s->nranks);
  m.nblocks_total = n;
  m.local_gids.clear();
  m.gid_local.clear();
  m.gid_rank.assign(n, 0);
  for (int g = 0; g < n; ++g) {
    m.gid_rank[g] = s->amr_part.Owner(g);
    if (m.gid_rank[g] == s->rank) {
      m.gid_local[g] = (int)m.local_gids.size();
      m.local_gids.push_back(g);
    }
  }
  m.peers.clear();
  for (auto &p : m.plan) p.clear();
}

// the global plans of the current forest and this rank's share of them
void amr_localize(apk_sim *s) {
  BuildAmrPlans(*s->amr, s->amr_geom, s->amr_plans);
  const AmrPlans &g = s->amr_plans;
  const AmrPartition &part = s->amr_part;
  auto &l = s->amr_local;
  l = apk_sim::AmrLocalPlans();
  const int rank = s->rank;
  auto take_ops = [&](const std::vector<AmrRefOp> &in, std::vector<AmrRefOp> &out) {
    for (AmrRefOp o : in) {
      if (part.Owner(o.dst_block) != rank) continue;  // (these operators work inside one block)
      o.src_block -= part.first[rank];
      o.dst_block -= part.first[rank];
      out.push_back(o);
    }
  };
  auto take_bc = [&](const std::vector<BoxRegion> &in, std::vector<BoxRegion> &out) {
    for (BoxRegion r : in) {
      if (part.Owner(r.dst_block) != rank) continue;
      r.src_block -= part.first[rank];
      r.dst_block -= part.first[rank];
      out.push_back(r);
    }
  };
  take_ops(g.restrict_own, l.restrict_own);
  take_ops(g.prolongate, l.prolongate);
  for (int d = 0; d < 3; ++d) {
    take_ops(g.flux_restrict[d], l.flux_restrict[d]);
    take_bc(g.coarse_bc[d], l.coarse_bc[d]);
    take_bc(g.fine_bc[d], l.fine_bc[d]);
  }
  s->amr_halo.plan = AmrMessages();
  s->amr_fluxmsg.plan = AmrMessages();
  s->amr_halo_faces.plan = AmrMessages();
  AmrRegisterPeers(g.fill, part, part, rank, s->amr_halo.plan);
  AmrLocalize(g.fill, part, part, rank, s->amr_halo.plan, l.fill, l.fill_pack, l.fill_unpack);
  {  // the stage loop's exchange: the same lists without the boxes behind edges and corners
    std::vector<BoxRegion> faces;
    for (const BoxRegion &r : g.fill)
      if (!r.corner) faces.push_back(r);
    AmrRegisterPeers(faces, part, part, rank, s->amr_halo_faces.plan);
    AmrLocalize(faces, part, part, rank, s->amr_halo_faces.plan, l.fill_faces, l.fill_pack_faces, l.fill_unpack_faces);
    for (const AmrRefOp &o : l.prolongate)
      if (!o.corner) l.prolongate_faces.push_back(o);
    for (const BoxRegion &r : l.fill_faces)
      if (!r.same_face) l.fill_direct.push_back(r);
  }
  s->amr_halo_shell.plan = AmrMessages();
  if (s->amr_geom.ng > AMR_SHELL_DEPTH) {  // the shell exchange: the same walk with shallower boxes
    AmrPlans shell;
    BuildAmrPlans(*s->amr, s->amr_geom, shell, AMR_SHELL_DEPTH);
    AmrRegisterPeers(shell.fill, part, part, rank, s->amr_halo_shell.plan);
    AmrLocalize(shell.fill, part, part, rank, s->amr_halo_shell.plan, l.fill_shell, l.fill_pack_shell, l.fill_unpack_shell);
    for (const BoxRegion &r : l.fill_shell)
      if (!r.same_face) l.fill_shell_direct.push_back(r);
    take_ops(shell.prolongate, l.prolongate_shell);
    for (int d = 0; d < 3; ++d) take_bc(shell.fine_bc[d], l.fine_bc_shell[d]);
  }
  for (int d = 0; d < 3; ++d) {  // (BuildAmrPlans appends a face's restriction and its copy together: same index)
    for (size_t n = 0; n < g.flux_copy[d].size(); ++n) {
      const int fine_owner = part.Owner(g.flux_copy[d][n].src_block), coarse_owner = part.Owner(g.flux_copy[d][n].dst_block);
      if (fine_owner != rank) continue;
      AmrRefOp o = g.flux_restrict[d][n];
      o.src_block -= part.first[rank];
      o.dst_block -= part.first[rank];
      (coarse_owner == rank ? l.flux_fused_ops[d] : l.flux_restrict_remote[d]).push_back(o);
    }
  }
  for (int d = 0; d < 3; ++d) AmrRegisterPeers(g.flux_copy[d], part, part, rank, s->amr_fluxmsg.plan);
  for (int d = 0; d < 3; ++d)
    AmrLocalize(g.flux_copy[d], part, part, rank, s->amr_fluxmsg.plan, l.flux_copy[d], l.flux_pack[d], l.flux_unpack[d]);
}

void amr_initialize(apk_sim *s, bool adaptive) {
  ParameterInput &pin = s->pin;
  Mesh &m = s->mesh;
  if (s->problem_id == "turbulence") throw std::runtime_error("the turbulence driver needs a uniform mesh");
  s->amr.reset(new AmrTree());
  AmrTree &t = *s->amr;
  for (int d = 0; d < 3; ++d) {
    t.nrb[d] = m.nb[d];
    t.act[d] = m.Active(d);
    t.bc_in[d] = m.bc_in[d];
    t.bc_out[d] = m.bc_out[d];
  }
  t.ndim = m.ndim;
  s->amr_adaptive = adaptive;
  int max_level = adaptive ? pin.GetOrAddInteger("parthenon/mesh", "numlevel", 1) - 1 : 0;
  if (max_level < 0) throw std::runtime_error("parthenon/mesh/numlevel must be at least 1");
  s->amr_derefine_count = pin.GetOrAddInteger("parthenon/mesh", "derefine_count", 10);
  s->amr_check_interval = pin.GetOrAddInteger("parthenon/mesh", "check_refine_interval", 1);
  struct Region {
    double lo[3], hi[3];
    int level;
  };
  std::vector<Region> regions;
  const char *mink[3] = {"x1min", "x2min", "x3min"}, *maxk[3] = {"x1max", "x2max", "x3max"};
  for (const std::string &blk : pin.BlocksWithPrefix("parthenon/static_refinement")) {
    Region r;
    for (int d = 0; d < 3; ++d) {
      r.lo[d] = m.Active(d) ? pin.GetReal(blk, mink[d]) : s->xmin[d];
      r.hi[d] = m.Active(d) ? pin.GetReal(blk, maxk[d]) : s->xmax[d];
      if (r.lo[d] > r.hi[d]) throw std::runtime_error("static refinement region of <" + blk + "> is inverted");
      if (r.lo[d] < s->xmin[d] || r.hi[d] > s->xmax[d]) throw std::runtime_error("static refinement region of <" + blk + "> lies outside of the mesh");
    }
    r.level = pin.GetInteger(blk, "level");
    if (r.level < 1) throw std::runtime_error("static refinement level must be at least 1");
    max_level = std::max(max_level, r.level);
    regions.push_back(r);
  }
  if (max_level > 12) throw std::runtime_error("more than 12 refinement levels");
  t.max_level = max_level;
  AmrGeom &g = s->amr_geom;
  for (int d = 0; d < 3; ++d) {
    g.mb[d] = m.mb[d];
    g.act[d] = m.Active(d);
  }
  g.ng = m.ng;
  g.cng = (m.ng + 1) / 2 + 1;
  g.nvar = m.nvar;
  g.Build();
  t.InitRoot();
  // static regions: split every block that overlaps a region until it has the region's level
  for (const Region &r : regions) {
    for (int lev = 0; lev < r.level; ++lev) {
      std::vector<AmrLeaf> todo;
      for (const auto &kv : t.leafmap) {
        const AmrLeaf &l = kv.second;
        if (l.level != lev) continue;
        bool overlap = true;
        for (int d = 0; d < 3; ++d) {
          if (!m.Active(d)) continue;
          const double w = level_dx(s, l.level, d) * m.mb[d];
          const double lo = s->xmin[d] + l.lx[d] * w, hi = lo + w;
          if (hi <= r.lo[d] || lo >= r.hi[d]) {
            // a degenerate region (lo == hi) still selects the block that contains the point
            if (!(r.lo[d] == r.hi[d] && lo <= r.lo[d] && r.lo[d] < hi)) overlap = false;
          }
        }
        if (overlap) todo.push_back(l);
      }
      for (const AmrLeaf &l : todo) t.RefineBalanced(l.level, l.lx);
    }
  }
  t.Reindex();
  amr_sync_mesh(s);
  amr_localize(s);
}

// ---- mesh refinement on the device --------------------------------------------------------------
double *amr_base(apk_sim *s, int parity, int kind, int block, const apk_sim::MsgSet *msgs) {
  switch (kind) {
  case RK_BLOCK: return s->d_cons2[parity] + (int64_t)block * s->nper;
  case RK_COARSE: return s->d_coarse + (int64_t)block * s->amr_geom.coarse_doubles;
  case RK_FLUX1: case RK_FLUX2: case RK_FLUX3: return s->d_flux[kind - RK_FLUX1] + (int64_t)block * s->nper;
  case RK_SEND: return msgs ? msgs->send[block] : nullptr;
  case RK_RECV: return msgs ? msgs->recv[block] : nullptr;
  default: return nullptr;
  }
}

apk_copy_region to_copy_region(const BoxRegion &r, const double *src, double *dst) {
  apk_copy_region c{};
  c.src = src + r.src_off;
  c.dst = dst + r.dst_off;
  for (int q = 0; q < 3; ++q) c.ext[q] = r.ext[q];
  c.nvar = r.nvar;
  for (int q = 0; q < 4; ++q) {
    c.src_stride[q] = r.src_stride[q];
    c.dst_stride[q] = r.dst_stride[q];
  }
  c.flip_var = r.flip_var;
  return c;
}

int amr_make_copy_plan(apk_sim *s, int parity, const std::vector<BoxRegion> &regions, const apk_sim::MsgSet *msgs,
                       apk_copy_plan **out) {
  std::vector<apk_copy_region> regs;
  for (const BoxRegion &r : regions)
    regs.push_back(to_copy_region(r, amr_base(s, parity, r.src_kind, r.src_block, msgs), amr_base(s, parity, r.dst_kind, r.dst_block, msgs)));
  return apk_copy_plan_create(s->ctx, regs.data(), (int)regs.size(), out);
}

// ONE refine plan for the boxes of all levels: every box carries the cell widths of its level (the
// operators difference cell-centre coordinates).  Ops carry LOCAL block numbers for the arrays and the
// GLOBAL leaf number for the geometry.
int amr_make_refine_plans(apk_sim *s, int parity, const std::vector<AmrRefOp> &ops, std::vector<apk_refine_plan *> &out) {
  for (apk_refine_plan *p : out) apk_refine_plan_destroy(p);
  out.clear();
  const AmrGeom &g = s->amr_geom;
  std::vector<apk_refine_op> dev;
  for (const AmrRefOp &o : ops) {
    apk_refine_op d{};
    d.kind = o.kind;
    d.src = amr_base(s, parity, o.src_kind, o.src_block, nullptr);
    d.dst = amr_base(s, parity, o.dst_kind, o.dst_block, nullptr);
    for (int q = 0; q < 3; ++q) {
      d.lo[q] = o.lo[q];
      d.hi[q] = o.hi[q];
      d.dx[q] = level_dx(s, o.level, q);
      d.xmin[q] = s->xmin[q] + (double)s->amr->leaves[o.geom_block].lx[q] * g.mb[q] * d.dx[q];
    }
    dev.push_back(d);
  }
  if (dev.empty()) return APK_OK;
  apk_refine_geom rg{};
  for (int q = 0; q < 3; ++q) {
    rg.nx[q] = g.mb[q];
    rg.dx[q] = level_dx(s, 0, q);
  }
  rg.ng = g.ng;
  rg.cng = g.cng;
  apk_refine_plan *p = nullptr;
  SIM_TRY(s, apk_refine_plan_create(s->ctx, &rg, g.nvar, dev.data(), (int)dev.size(), &p));
  out.push_back(p);
  return APK_OK;
}

void amr_destroy_graphs(apk_sim *s);

void amr_destroy_device_plans(apk_sim *s) {
  auto &a = s->amr_dev;
  amr_destroy_graphs(s);
  dev_free(s, reinterpret_cast<double *>(a.d_cf_faces));
  a.d_cf_faces = nullptr;
  for (int par = 0; par < 2; ++par) {
    for (apk_refine_plan *p : a.restrict_own[par]) apk_refine_plan_destroy(p);
    for (apk_refine_plan *p : a.prolongate[par]) apk_refine_plan_destroy(p);
    a.restrict_own[par].clear();
    a.prolongate[par].clear();
    apk_copy_plan_destroy(a.fill[par]);
    apk_copy_plan_destroy(a.fill_pack[par]);
    apk_copy_plan_destroy(a.fill_unpack[par]);
    a.fill[par] = a.fill_pack[par] = a.fill_unpack[par] = nullptr;
    for (apk_refine_plan *p : a.prolongate_faces[par]) apk_refine_plan_destroy(p);
    a.prolongate_faces[par].clear();
    apk_copy_plan_destroy(a.fill_faces[par]);
    apk_copy_plan_destroy(a.fill_pack_faces[par]);
    apk_copy_plan_destroy(a.fill_unpack_faces[par]);
    a.fill_faces[par] = a.fill_pack_faces[par] = a.fill_unpack_faces[par] = nullptr;
    apk_copy_plan_destroy(a.fill_direct[par]);
    a.fill_direct[par] = nullptr;
    for (apk_refine_plan *p : a.prolongate_shell[par]) apk_refine_plan_destroy(p);
    a.prolongate_shell[par].clear();
    apk_copy_plan_destroy(a.fill_shell[par]);
    apk_copy_plan_destroy(a.fill_shell_direct[par]);
    a.fill_shell_direct[par] = nullptr;
    apk_copy_plan_destroy(a.fill_pack_shell[par]);
    apk_copy_plan_destroy(a.fill_unpack_shell[par]);
    a.fill_shell[par] = a.fill_pack_shell[par] = a.fill_unpack_shell[par] = nullptr;
    for (int d = 0; d < 3; ++d) {
      apk_copy_plan_destroy(a.fine_bc_shell[par][d]);
      a.fine_bc_shell[par][d] = nullptr;
    }
    for (int d = 0; d < 3; ++d) {
      apk_copy_plan_destroy(a.coarse_bc[par][d]);
      apk_copy_plan_destroy(a.fine_bc[par][d]);
      a.coarse_bc[par][d] = a.fine_bc[par][d] = nullptr;
    }
  }
  for (int d = 0; d < 3; ++d) {
    for (apk_refine_plan *p : a.flux_restrict[d]) apk_refine_plan_destroy(p);
    a.flux_restrict[d].clear();
    for (apk_refine_plan *p : a.flux_restrict_remote[d]) apk_refine_plan_destroy(p);
    a.flux_restrict_remote[d].clear();
    apk_copy_plan_destroy(a.flux_copy[d]);
    apk_copy_plan_destroy(a.flux_pack[d]);
    apk_copy_plan_destroy(a.flux_unpack[d]);
    a.flux_copy[d] = a.flux_pack[d] = a.flux_unpack[d] = nullptr;
    for (int par = 0; par < 2; ++par) {
      apk_flux_fix_plan_destroy(a.flux_fix[par][d]);
      apk_flux_fix_plan_destroy(a.flux_fix_unpack[par][d]);
      a.flux_fix[par][d] = a.flux_fix_unpack[par][d] = nullptr;
    }
  }
  for (int par = 0; par < 2; ++par) {
    apk_flux_fix_plan_destroy(a.flux_fix_all[par]);
    apk_flux_fix_plan_destroy(a.flux_fix_unpack_all[par]);
    a.flux_fix_all[par] = a.flux_fix_unpack_all[par] = nullptr;
  }
}

// the flux-correction copies of direction d as corrections of the cells next to the face (fused path)
// (ops: the restriction operator of every region -- same-rank faces: the kernel averages the fine block's fluxes
// itself, no restricted plane in between; null: the regions' sources hold the averages)
int amr_fix_regions(apk_sim *s, int parity, int d, const std::vector<BoxRegion> &regions, const apk_sim::MsgSet *msgs,
                    std::vector<apk_flux_fix_region> &regs, const std::vector<AmrRefOp> *ops = nullptr) {
  const AmrGeom &g = s->amr_geom;
  if (ops && ops->size() != regions.size()) return fail(s, APK_ERR_INVALID, "flux correction: operators and copies out of step");
  for (size_t n = 0; n < regions.size(); ++n) {
    const BoxRegion &r = regions[n];
    apk_flux_fix_region f{};
    f.fine_avg = amr_base(s, parity, r.src_kind, r.src_block, msgs) + r.src_off;
    if (ops) {
      const AmrRefOp &o = (*ops)[n];
      if (o.kind != APK_RO_RESTRICT_FLUX1 + d || o.dst_block != r.src_block) return fail(s, APK_ERR_INVALID, "flux correction: operator does not belong to the copy");
      int64_t off = 0;
      for (int q = 0; q < 3; ++q) off += (g.act[q] ? (int64_t)(o.lo[q] - g.cs[q]) * 2 + g.fs[q] : 0) * g.fst[q];
      f.fine_avg = s->d_flux[d] + (int64_t)o.src_block * s->nper + off;
      f.average = d + 1;
      f.ndim = s->mesh.ndim;
      double w = 1.0;  // (restrict_cell's order of the factors)
      for (int q = 0; q < 3; ++q)
        if (q != d) w *= level_dx(s, o.level, q);
      f.fine_area = w;
      for (int q = 0; q < 3; ++q) f.fine_stride[q] = g.act[q] ? g.fst[q] : 0;
    }
    f.coarse_flux = amr_base(s, parity, r.dst_kind, r.dst_block, msgs) + r.dst_off;
    const int idx = (int)((r.dst_off / g.fst[d]) % g.fn[d]);  // face index along d inside the block
    const bool lower = idx == g.fs[d];
    f.cons = s->d_cons2[parity] + (int64_t)r.dst_block * s->nper + r.dst_off - (lower ? 0 : g.fst[d]);
    for (int q = 0; q < 3; ++q) f.ext[q] = r.ext[q];
    f.nvar = r.nvar;
    for (int q = 0; q < 4; ++q) {
      f.src_stride[q] = r.src_stride[q];
      f.dst_stride[q] = r.dst_stride[q];
    }
    if (ops) {
      for (int q = 0; q < 3; ++q) f.src_stride[q] = g.act[q] ? 2 * g.fst[q] : 0;
      f.src_stride[3] = g.fst[3];
    }
    f.scale = (lower ? 1.0 : -1.0) / level_dx(s, block_level(s, r.dst_block), d);
    regs.push_back(f);
  }
  return APK_OK;
}

// The fix plans of one conserved buffer: ONE plan for the regions of all directions (one launch per stage instead of
// three: the launches are 7 - 8 us each for a few hundred small regions), or -- a forest whose regions the merged form
// refuses -- one per direction as before.
int amr_make_fix_plans(apk_sim *s, int parity, const std::vector<BoxRegion> (&regions)[3], const apk_sim::MsgSet *msgs,
                       const std::vector<AmrRefOp> (*ops)[3], apk_flux_fix_plan **merged, apk_flux_fix_plan *(&per_dir)[3]) {
  std::vector<apk_flux_fix_region> per[3], all;
  int nd[3] = {0, 0, 0};
  for (int d = 0; d < s->mesh.ndim; ++d) {
    SIM_TRY(s, amr_fix_regions(s, parity, d, regions[d], msgs, per[d], ops ? &(*ops)[d] : nullptr));
    nd[d] = (int)per[d].size();
    all.insert(all.end(), per[d].begin(), per[d].end());
  }
  *merged = nullptr;
  const int rc = apk_flux_fix_plan_create_merged(s->ctx, all.data(), nd, s->d_cons2[parity], s->nper, merged);
  if (rc == APK_OK) return APK_OK;
  if (rc != APK_ERR_UNSUPPORTED) return fail(s, rc, apk_last_error(s->ctx));
  *merged = nullptr;
  for (int d = 0; d < s->mesh.ndim; ++d) SIM_TRY(s, apk_flux_fix_plan_create(s->ctx, per[d].data(), nd[d], &per_dir[d]));
  return APK_OK;
}

// message buffers of a set: (re)allocated when a message outgrows its buffer, never shrunk
int amr_ensure_buffers(apk_sim *s, apk_sim::MsgSet &m, const char *name) {
  const size_t np = m.plan.peers.size();
  // (buffers belong to positions in the rank-sorted peer list, not to ranks: they are scratch)
  if (m.send.size() != np) {
    for (double *b : m.send) dev_free(s, b);
    for (double *b : m.recv) dev_free(s, b);
    m.send.assign(np, nullptr);
    m.recv.assign(np, nullptr);
    m.send_cap.assign(np, 0);
    m.recv_cap.assign(np, 0);
  }
  for (size_t p = 0; p < np; ++p) {
    const PeerPlan &pp = m.plan.peers[p];
    if (pp.send_count > m.send_cap[p]) {
      dev_free(s, m.send[p]);
      m.send_cap[p] = pp.send_count + pp.send_count / 4;
      SIM_TRY(s, dev_alloc(s, (std::string(name) + ":send:" + std::to_string(pp.rank)).c_str(), m.send_cap[p] * sizeof(double), &m.send[p]));
    }
    if (pp.recv_count > m.recv_cap[p]) {
      dev_free(s, m.recv[p]);
      m.recv_cap[p] = pp.recv_count + pp.recv_count / 4;
      SIM_TRY(s, dev_alloc(s, (std::string(name) + ":recv:" + std::to_string(pp.rank)).c_str(), m.recv_cap[p] * sizeof(double), &m.recv[p]));
    }
  }
  s->msg_generation += 1;
  return APK_OK;
}

void amr_free_buffers(apk_sim *s, apk_sim::MsgSet &m) {
  for (double *b : m.send) dev_free(s, b);
  for (double *b : m.recv) dev_free(s, b);
  m.send.clear();
  m.recv.clear();
  m.send_cap.clear();
  m.recv_cap.clear();
}

// one message per peer: hand the set to the comm ops (apk_sim_peer reports the active set)
int amr_exchange_messages(apk_sim *s, const apk_sim::MsgSet &m) {
  if (m.plan.peers.empty()) return APK_OK;
  if (!s->have_comm || !s->comm.exchange) return fail(s, APK_ERR_INVALID, "remote neighbours but no comm ops");
  if (s->active_msgs != &m) {
    s->active_msgs = &m;
    s->msg_generation += 1;
  }
  if (s->comm.exchange(s->comm.user) != 0) return fail(s, APK_ERR_DEVICE, "message exchange failed");
  return APK_OK;
}

// device arrays of a mesh of n local blocks (state, register, primitives, fluxes, coarse buffers)
int amr_allocate(apk_sim *s, size_t n, double *cons2[2], double **prim, double *flux[3], double **coarse) {
  const size_t bytes = (size_t)s->nper * n * sizeof(double);
  const size_t cbytes = (size_t)s->amr_geom.coarse_doubles * n * sizeof(double);
  SIM_TRY(s, dev_alloc(s, "cons", bytes, &cons2[0]));
  SIM_TRY(s, dev_alloc(s, "u1", bytes, &cons2[1]));
  SIM_TRY(s, dev_alloc(s, "prim", bytes, prim));
  SIM_TRY(s, dev_alloc(s, "coarse", cbytes, coarse));
  SIM_HIP(s, hipMemsetAsync(cons2[0], 0, bytes, hs(s)));
  SIM_HIP(s, hipMemsetAsync(cons2[1], 0, bytes, hs(s)));
  SIM_HIP(s, hipMemsetAsync(*prim, 0, bytes, hs(s)));
  SIM_HIP(s, hipMemsetAsync(*coarse, 0, cbytes, hs(s)));
  const char *tags[3] = {"flux1", "flux2", "flux3"};
  for (int d = 0; d < 3; ++d) {
    flux[d] = nullptr;
    if (d >= s->mesh.ndim) continue;
    SIM_TRY(s, dev_alloc(s, tags[d], bytes, &flux[d]));
    SIM_HIP(s, hipMemsetAsync(flux[d], 0, bytes, hs(s)));
  }
  return APK_OK;
}

// (re)build everything that depends on the block list: packs, message buffers, device plans
int amr_rebuild(apk_sim *s) {
  try {
    amr_sync_mesh(s);
    amr_localize(s);
  } catch (const std::exception &e) {
    return fail(s, APK_ERR_INVALID, e.what());
  }
  SIM_TRY(s, amr_ensure_buffers(s, s->amr_halo, "halo"));
  SIM_TRY(s, amr_ensure_buffers(s, s->amr_halo_faces, "halo_faces"));
  SIM_TRY(s, amr_ensure_buffers(s, s->amr_halo_shell, "halo_shell"));
  SIM_TRY(s, amr_ensure_buffers(s, s->amr_fluxmsg, "fluxcorr"));
  amr_destroy_device_plans(s);
  auto &a = s->amr_dev;
  const auto &p = s->amr_local;
  for (int par = 0; par < 2; ++par) {
    SIM_TRY(s, amr_make_refine_plans(s, par, p.restrict_own, a.restrict_own[par]));
    SIM_TRY(s, amr_make_refine_plans(s, par, p.prolongate, a.prolongate[par]));
    SIM_TRY(s, amr_make_copy_plan(s, par, p.fill, nullptr, &a.fill[par]));
    SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_pack, &s->amr_halo, &a.fill_pack[par]));
    SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_unpack, &s->amr_halo, &a.fill_unpack[par]));
    SIM_TRY(s, amr_make_refine_plans(s, par, p.prolongate_faces, a.prolongate_faces[par]));
    SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_faces, nullptr, &a.fill_faces[par]));
    SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_pack_faces, &s->amr_halo_faces, &a.fill_pack_faces[par]));
    SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_unpack_faces, &s->amr_halo_faces, &a.fill_unpack_faces[par]));
    SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_direct, nullptr, &a.fill_direct[par]));
    if (amr_has_shell(s)) {
      SIM_TRY(s, amr_make_refine_plans(s, par, p.prolongate_shell, a.prolongate_shell[par]));
      SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_shell, nullptr, &a.fill_shell[par]));
      SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_shell_direct, nullptr, &a.fill_shell_direct[par]));
      SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_pack_shell, &s->amr_halo_shell, &a.fill_pack_shell[par]));
      SIM_TRY(s, amr_make_copy_plan(s, par, p.fill_unpack_shell, &s->amr_halo_shell, &a.fill_unpack_shell[par]));
      for (int d = 0; d < 3; ++d) SIM_TRY(s, amr_make_copy_plan(s, par, p.fine_bc_shell[d], nullptr, &a.fine_bc_shell[par][d]));
    }
    for (int d = 0; d < 3; ++d) {
      SIM_TRY(s, amr_make_copy_plan(s, par, p.coarse_bc[d], nullptr, &a.coarse_bc[par][d]));
      SIM_TRY(s, amr_make_copy_plan(s, par, p.fine_bc[d], nullptr, &a.fine_bc[par][d]));
    }
  }
  for (int d = 0; d < s->mesh.ndim; ++d) {
    SIM_TRY(s, amr_make_refine_plans(s, 0, p.flux_restrict[d], a.flux_restrict[d]));
    SIM_TRY(s, amr_make_refine_plans(s, 0, p.flux_restrict_remote[d], a.flux_restrict_remote[d]));
    SIM_TRY(s, amr_make_copy_plan(s, 0, p.flux_copy[d], nullptr, &a.flux_copy[d]));
    SIM_TRY(s, amr_make_copy_plan(s, 0, p.flux_pack[d], &s->amr_fluxmsg, &a.flux_pack[d]));
    SIM_TRY(s, amr_make_copy_plan(s, 0, p.flux_unpack[d], &s->amr_fluxmsg, &a.flux_unpack[d]));
  }
  for (int par = 0; par < 2; ++par) {
    SIM_TRY(s, amr_make_fix_plans(s, par, p.flux_copy, nullptr, &p.flux_fused_ops, &a.flux_fix_all[par], a.flux_fix[par]));
    SIM_TRY(s, amr_make_fix_plans(s, par, p.flux_unpack, &s->amr_fluxmsg, nullptr, &a.flux_fix_unpack_all[par], a.flux_fix_unpack[par]));
  }
  {  // faces with a level change behind them
    const int nlb = (int)s->mesh.local_gids.size(), first = s->amr_part.first[s->rank];
    const AmrTree &t = *s->amr;
    std::vector<int> cf;
    for (int lb = 0; lb < nlb; ++lb) {
      const AmrLeaf &l = t.leaves[first + lb];
      for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2 && t.act[d]; ++side) {
          int pos[3] = {l.lx[0], l.lx[1], l.lx[2]}, leaf;
          pos[d] += side ? 1 : -1;
          const int kind = t.Classify(l.level, pos, &leaf);
          if (kind == NB_FINER || kind == NB_COARSER) cf.push_back(6 * lb + 2 * d + side);
        }
    }
    a.n_cf_faces = (int)cf.size();
    double *p8 = nullptr;
    SIM_TRY(s, dev_alloc(s, "cf_faces", sizeof(int) * (cf.size() + 2), &p8));
    a.d_cf_faces = reinterpret_cast<int *>(p8);
    if (!cf.empty()) SIM_HIP(s, hipMemcpy(a.d_cf_faces, cf.data(), sizeof(int) * cf.size(), hipMemcpyHostToDevice));
  }
  {  // apk_stage_args.face_neighbor of the refined mesh: the same-rank block of the SAME level behind each face, or -1
    const int nlb = (int)s->mesh.local_gids.size(), first = s->amr_part.first[s->rank];
    const AmrTree &t = *s->amr;
    std::vector<int> tab(6 * (size_t)nlb + 2, -1);
    for (int lb = 0; lb < nlb; ++lb) {
      const AmrLeaf &l = t.leaves[first + lb];
      for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2 && t.act[d]; ++side) {
          int pos[3] = {l.lx[0], l.lx[1], l.lx[2]}, leaf = -1;
          pos[d] += side ? 1 : -1;
          if (t.Classify(l.level, pos, &leaf) == NB_SAME && s->amr_part.Owner(leaf) == s->rank) tab[6 * (size_t)lb + 2 * d + side] = leaf - first;
        }
    }
    dev_free(s, reinterpret_cast<double *>(s->d_face_nbr));
    s->d_face_nbr = nullptr;
    double *p8 = nullptr;
    SIM_TRY(s, dev_alloc(s, "face_neighbors", sizeof(int) * tab.size(), &p8));
    s->d_face_nbr = reinterpret_cast<int *>(p8);
    SIM_HIP(s, hipMemcpy(s->d_face_nbr, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice));
  }
  // (a graph launch costs the host ~8 us and the stream a gap of that size: with no messages to wait for between the
  // halves -- one rank -- an exchange is ONE graph; APK_NO_GRAPH=1 launches everything one by one)
  a.xchg_whole = s->amr_halo.plan.peers.empty() && s->amr_halo_faces.plan.peers.empty() && s->amr_halo_shell.plan.peers.empty();
  const bool w = a.xchg_whole;
  for (int par = 0; par < 2; ++par) {
    amr_capture_half(s, par, true, 0, &a.xchg_pre[par], w);
    amr_capture_half(s, par, false, 0, &a.xchg_post[par]);
    amr_capture_half(s, par, true, 1, &a.xchg_pre_faces[par], w);
    amr_capture_half(s, par, false, 1, &a.xchg_post_faces[par]);
    amr_capture_half(s, par, true, 2, &a.xchg_pre_direct[par], w);
    if (amr_has_shell(s)) {
      amr_capture_half(s, par, true, AMR_XCHG_SHELL, &a.xchg_pre_shell[par], w);
      amr_capture_half(s, par, true, AMR_XCHG_SHELL_DIRECT, &a.xchg_pre_shell_direct[par], w);
      amr_capture_half(s, par, false, AMR_XCHG_SHELL, &a.xchg_post_shell[par]);
    }
  }
  s->amr_ghost_state = AMR_GHOSTS_COMPLETE;  // (whoever rebuilt the plans fills the new mesh completely next)
  return build_packs(s);
}

// the multilevel ghost exchange of the state in cons buffer `buf` (see amr.hpp), in the two halves
// either side of the message exchange
// is there a shell exchange (more ghost layers than it fills)?
bool amr_has_shell(const apk_sim *s) { return s->amr_geom.ng > AMR_SHELL_DEPTH; }

// (mode: AMR_XCHG_FULL, AMR_XCHG_FACES, AMR_XCHG_DIRECT or AMR_XCHG_SHELL)
int amr_exchange_pre(apk_sim *s, int buf, int mode) {
  auto &a = s->amr_dev;
  for (apk_refine_plan *p : a.restrict_own[buf]) SIM_TRY(s, apk_refine_plan_run(s->ctx, p, s->stream));
  if (mode == AMR_XCHG_SHELL || mode == AMR_XCHG_SHELL_DIRECT) {
    SIM_TRY(s, apk_copy_plan_run(s->ctx, a.fill_pack_shell[buf], s->stream));
    return apk_copy_plan_run(s->ctx, mode == AMR_XCHG_SHELL_DIRECT ? a.fill_shell_direct[buf] : a.fill_shell[buf], s->stream);
  }
  const bool faces = mode != AMR_XCHG_FULL;
  SIM_TRY(s, apk_copy_plan_run(s->ctx, faces ? a.fill_pack_faces[buf] : a.fill_pack[buf], s->stream));
  SIM_TRY(s, apk_copy_plan_run(s->ctx, mode == AMR_XCHG_DIRECT ? a.fill_direct[buf] : (faces ? a.fill_faces[buf] : a.fill[buf]), s->stream));
  return APK_OK;
}
int amr_exchange_post(apk_sim *s, int buf, int mode) {
  auto &a = s->amr_dev;
  if (mode == AMR_XCHG_SHELL || mode == AMR_XCHG_SHELL_DIRECT) {
    SIM_TRY(s, apk_copy_plan_run(s->ctx, a.fill_unpack_shell[buf], s->stream));
    for (int d = 0; d < 3; ++d) SIM_TRY(s, apk_copy_plan_run(s->ctx, a.coarse_bc[buf][d], s->stream));
    for (apk_refine_plan *p : a.prolongate_shell[buf]) SIM_TRY(s, apk_refine_plan_run(s->ctx, p, s->stream));
    for (int d = 0; d < 3; ++d) SIM_TRY(s, apk_copy_plan_run(s->ctx, a.fine_bc_shell[buf][d], s->stream));
    return APK_OK;
  }
  const bool faces = mode != AMR_XCHG_FULL;
  SIM_TRY(s, apk_copy_plan_run(s->ctx, faces ? a.fill_unpack_faces[buf] : a.fill_unpack[buf], s->stream));
  for (int d = 0; d < 3; ++d) SIM_TRY(s, apk_copy_plan_run(s->ctx, a.coarse_bc[buf][d], s->stream));
  for (apk_refine_plan *p : (faces ? a.prolongate_faces[buf] : a.prolongate[buf])) SIM_TRY(s, apk_refine_plan_run(s->ctx, p, s->stream));
  for (int d = 0; d < 3; ++d) SIM_TRY(s, apk_copy_plan_run(s->ctx, a.fine_bc[buf][d], s->stream));
  return APK_OK;
}

// Capture one half into an executable graph.  The sim's stream may be the legacy default stream,
// which cannot be captured: the launches are recorded on a private stream (nothing executes) and
// the graph is launched on the sim's stream later.  Any failure leaves *out null: the caller then
// launches the plans one by one as before.
// whole (with pre): both halves in one graph -- for meshes whose exchange has no messages between them
void amr_capture_half(apk_sim *s, int buf, bool pre, int mode, void **out, bool whole) {
  *out = nullptr;
  static const bool disabled = std::getenv("APK_NO_GRAPH") != nullptr;  // A/B switch
  if (disabled) return;
  hipStream_t cs = nullptr;
  if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  bool ok = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) == hipSuccess;
  if (ok) {
    const apk_stream_t saved = s->stream;
    const std::string saved_err = s->err;
    s->stream = reinterpret_cast<apk_stream_t>(cs);
    int rc = pre ? amr_exchange_pre(s, buf, mode) : amr_exchange_post(s, buf, mode);
    if (rc == APK_OK && pre && whole) rc = amr_exchange_post(s, buf, mode);
    s->stream = saved;
    ok = hipStreamEndCapture(cs, &graph) == hipSuccess && rc == APK_OK && graph != nullptr;
    if (rc != APK_OK) s->err = saved_err;
  }
  size_t nodes = 0;
  // (A graph launch leaves ~8 us of idle stream in front of its first node; launched one by one, kernels follow each other
  // within 1 - 2 us as long as the host is ahead -- it is, by half a cycle, on meshes of a few hundred blocks.  A graph pays
  // where it replaces many launches: physical-boundary phases, several prolongation plans.  Round 6, config 5's mesh, one
  // rank, periodic -- 4 - 5 nodes per exchange: 1.354e9 zone-cycles/s as graphs, 1.373e9 launched one by one, same box.)
  if (ok) ok = hipGraphGetNodes(graph, nullptr, &nodes) == hipSuccess && nodes > 6;
  if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipStreamDestroy(cs);
  (void)hipGetLastError();
  if (ok) *out = exec;
}

void amr_destroy_graphs(apk_sim *s) {
  auto &a = s->amr_dev;
  for (int buf = 0; buf < 2; ++buf) {
    if (a.xchg_pre[buf]) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(a.xchg_pre[buf]));
    if (a.xchg_post[buf]) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(a.xchg_post[buf]));
    if (a.xchg_pre_faces[buf]) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(a.xchg_pre_faces[buf]));
    if (a.xchg_post_faces[buf]) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(a.xchg_post_faces[buf]));
    if (a.xchg_pre_direct[buf]) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(a.xchg_pre_direct[buf]));
    if (a.xchg_pre_shell[buf]) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(a.xchg_pre_shell[buf]));
    if (a.xchg_pre_shell_direct[buf]) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(a.xchg_pre_shell_direct[buf]));
    a.xchg_pre_shell_direct[buf] = nullptr;
    if (a.xchg_post_shell[buf]) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(a.xchg_post_shell[buf]));
    a.xchg_pre[buf] = a.xchg_post[buf] = a.xchg_pre_faces[buf] = a.xchg_post_faces[buf] = a.xchg_pre_direct[buf] = nullptr;
    a.xchg_pre_shell[buf] = a.xchg_post_shell[buf] = nullptr;
  }
}

// AMR_XCHG_FACES: the stage loop's exchange -- everything but the ghost zones behind edges and corners,
// which no sweep or flux correction reads (sync_ghosts completes them for accessors, tagging and regridding;
// the last stage of a cycle that checks the refinement criteria exchanges in full).  AMR_XCHG_DIRECT: nor the
// ghost zones behind faces shared with a same-rank block of the same level, which the stages then read from
// that block's interior (the face table built in amr_rebuild; same-level faces of OTHER ranks still arrive
// in the messages).
int amr_exchange(apk_sim *s, int buf, int mode) {
  auto &a = s->amr_dev;
  if ((mode == AMR_XCHG_SHELL || mode == AMR_XCHG_SHELL_DIRECT) && !amr_has_shell(s)) mode = AMR_XCHG_FULL;
  const bool faces = mode == AMR_XCHG_FACES || mode == AMR_XCHG_DIRECT;
  const bool shell = mode == AMR_XCHG_SHELL || mode == AMR_XCHG_SHELL_DIRECT;
  void *pre = mode == AMR_XCHG_SHELL_DIRECT ? a.xchg_pre_shell_direct[buf]
              : mode == AMR_XCHG_SHELL ? a.xchg_pre_shell[buf]
              : mode == AMR_XCHG_DIRECT ? a.xchg_pre_direct[buf] : (faces ? a.xchg_pre_faces[buf] : a.xchg_pre[buf]);
  void *post = shell ? a.xchg_post_shell[buf] : (faces ? a.xchg_post_faces[buf] : a.xchg_post[buf]);
  if (pre) SIM_HIP(s, hipGraphLaunch(static_cast<hipGraphExec_t>(pre), hs(s)));
  else SIM_TRY(s, amr_exchange_pre(s, buf, mode));
  if (!(pre && a.xchg_whole)) {  // (else the graph held both halves)
    SIM_TRY(s, amr_exchange_messages(s, shell ? s->amr_halo_shell : (faces ? s->amr_halo_faces : s->amr_halo)));
    if (post) SIM_HIP(s, hipGraphLaunch(static_cast<hipGraphExec_t>(post), hs(s)));
    else SIM_TRY(s, amr_exchange_post(s, buf, mode));
  }
  // (of cons; the caller converts to primitives)
  s->amr_ghost_state = mode == AMR_XCHG_FULL ? AMR_GHOSTS_COMPLETE
                       : (mode == AMR_XCHG_SHELL ? AMR_GHOSTS_SHELL : (mode == AMR_XCHG_SHELL_DIRECT ? AMR_GHOSTS_SHELL_DIRECT : AMR_GHOSTS_FACES));
  return APK_OK;
}

// does the forest have a coarse-fine face at all?  (the global plan: the same answer on every rank)
bool amr_has_coarse_fine_faces(const apk_sim *s) {
  for (int d = 0; d < 3; ++d)
    if (!s->amr_plans.flux_copy[d].empty()) return true;
  return false;
}

// The flux correction for a stage that ran fused: the stage has applied every block's own face
// fluxes; recompute the fluxes on the block boundaries from the stage's input primitives, average
// the fine ones and correct the coarse cells next to each coarse-fine face by the difference.
// The boundary-plane fluxes of that correction only read the stage's input primitives and only write the flux arrays,
// which the fused stage does not touch: launched on a stream of their own in FRONT of the stage, they run in the wave
// slots the marches leave empty (a pack of a few hundred 16^3 blocks fills 80 % of them; by themselves the few hundred
// planes are one wave each and take 25 us of latency per stage).  Returns whether they were launched; amr_flux_fix then
// waits for them instead of computing them.  The caller has the flux arrays in place (ensure_flux_arrays).
bool amr_flux_planes_ahead(apk_sim *s, const apk_flux_cfg &cfg, int cons_input) {
  static const bool off = std::getenv("APK_AMR_PLANES_INLINE") != nullptr;  // A/B switch
  if (off || s->side_stream_failed || !amr_has_coarse_fine_faces(s)) return false;
  auto &a = s->amr_dev;
  if (!s->side_stream) {
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
      s->side_stream_failed = true;  // (not retried at every stage: the planes run on the sim's stream from now on)
      return false;
    }
    if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) {
      if (e0) (void)hipEventDestroy(e0);
      (void)hipStreamDestroy(st);
      s->side_stream_failed = true;
      return false;
    }
    s->side_stream = st, s->ev_fork = e0, s->ev_join = e1;
  }
  hipStream_t side = reinterpret_cast<hipStream_t>(s->side_stream);
  if (hipEventRecord(reinterpret_cast<hipEvent_t>(s->ev_fork), hs(s)) != hipSuccess) return false;
  if (hipStreamWaitEvent(side, reinterpret_cast<hipEvent_t>(s->ev_fork), 0) != hipSuccess) return false;
  // (cons_input >= 0: the stage's input is the conserved state in that buffer -- prim_from_cons)
  const int rc = cons_input >= 0
                     ? apk_calculate_fluxes_boundary_list_from_cons(s->ctx, s->mu0(), cfg, &s->pkg.eos, s->pkg.c_h, a.d_cf_faces, a.n_cf_faces,
                                                                    (long long)(s->d_cons2[cons_input] - s->d_cons2[s->cur]),
                                                                    reinterpret_cast<apk_stream_t>(side))
                     : apk_calculate_fluxes_boundary_list(s->ctx, s->mu0(), cfg, &s->pkg.eos, s->pkg.c_h, a.d_cf_faces, a.n_cf_faces,
                                                          reinterpret_cast<apk_stream_t>(side));
  // (whatever was enqueued is joined either way: through the event -- by amr_flux_fix, or by do_stage if the stage fails
  // in between -- or, if the event cannot be recorded, by waiting for the side stream here)
  if (hipEventRecord(reinterpret_cast<hipEvent_t>(s->ev_join), side) != hipSuccess) {
    (void)hipStreamSynchronize(side);
    return false;  // (the caller computes the planes on its own stream: the same values once more)
  }
  if (rc != APK_OK) {
    (void)hipStreamWaitEvent(hs(s), reinterpret_cast<hipEvent_t>(s->ev_join), 0);
    return false;
  }
  return true;
}

int amr_flux_fix(apk_sim *s, const apk_flux_cfg &cfg, double beta_dt, double psi_factor, bool planes_ahead, int cons_input) {
  if (!amr_has_coarse_fine_faces(s)) return APK_OK;
  auto &a = s->amr_dev;
  const int psi_var = (s->pkg.fluid == APK_FLUID_GLMMHD) ? 8 : -1;
  if (planes_ahead) {
    SIM_HIP(s, hipStreamWaitEvent(hs(s), reinterpret_cast<hipEvent_t>(s->ev_join), 0));
  } else if (cons_input >= 0) {
    // (the stage derived its input from the conserved state in buffer `cons_input`: so do the planes)
    SIM_TRY(s, apk_calculate_fluxes_boundary_list_from_cons(s->ctx, s->mu0(), cfg, &s->pkg.eos, s->pkg.c_h, a.d_cf_faces, a.n_cf_faces,
                                                            (long long)(s->d_cons2[cons_input] - s->d_cons2[s->cur]), s->stream));
  } else {
    SIM_TRY(s, apk_calculate_fluxes_boundary_list(s->ctx, s->mu0(), cfg, &s->pkg.eos, s->pkg.c_h, a.d_cf_faces, a.n_cf_faces, s->stream));
  }
  // (same-rank faces: the fix kernel averages the fine fluxes itself, reading the blocks' flux arrays; only faces whose
  // coarse side lives elsewhere are restricted into the coarse buffer, for the message -- direction by direction,
  // because the restricted planes of all three directions share the coarse buffers)
  if (a.flux_fix_all[s->cur]) SIM_TRY(s, apk_flux_fix_plan_run(s->ctx, a.flux_fix_all[s->cur], beta_dt, psi_var, psi_factor, s->stream));
  for (int d = 0; d < s->mesh.ndim; ++d) {
    for (apk_refine_plan *p : a.flux_restrict_remote[d]) SIM_TRY(s, apk_refine_plan_run(s->ctx, p, s->stream));
    if (!a.flux_fix_all[s->cur]) SIM_TRY(s, apk_flux_fix_plan_run(s->ctx, a.flux_fix[s->cur][d], beta_dt, psi_var, psi_factor, s->stream));
    SIM_TRY(s, apk_copy_plan_run(s->ctx, a.flux_pack[d], s->stream));
  }
  SIM_TRY(s, amr_exchange_messages(s, s->amr_fluxmsg));
  if (a.flux_fix_unpack_all[s->cur]) {
    SIM_TRY(s, apk_flux_fix_plan_run(s->ctx, a.flux_fix_unpack_all[s->cur], beta_dt, psi_var, psi_factor, s->stream));
  } else {
    for (int d = 0; d < s->mesh.ndim; ++d)
      SIM_TRY(s, apk_flux_fix_plan_run(s->ctx, a.flux_fix_unpack[s->cur][d], beta_dt, psi_var, psi_factor, s->stream));
  }
  return APK_OK;
}

// coarse-fine flux correction (hydro_driver.cpp:527-531): direction by direction, because the
// restricted fluxes of all three directions share the blocks' coarse buffers; faces whose coarse
// side lives on another rank travel in ONE message per peer after the three directions are packed
int amr_flux_correction(apk_sim *s) {
  auto &a = s->amr_dev;
  for (int d = 0; d < s->mesh.ndim; ++d) {
    for (apk_refine_plan *p : a.flux_restrict[d]) SIM_TRY(s, apk_refine_plan_run(s->ctx, p, s->stream));
    SIM_TRY(s, apk_copy_plan_run(s->ctx, a.flux_copy[d], s->stream));
    SIM_TRY(s, apk_copy_plan_run(s->ctx, a.flux_pack[d], s->stream));
  }
  SIM_TRY(s, amr_exchange_messages(s, s->amr_fluxmsg));
  for (int d = 0; d < s->mesh.ndim; ++d) SIM_TRY(s, apk_copy_plan_run(s->ctx, a.flux_unpack[d], s->stream));
  return APK_OK;
}

// ---- regridding -----------------------------------------------------------------------------
int refinement_criterion(apk_sim *s, int *criterion, double *p0, double *p1) {
  ParameterInput &pin = s->pin;
  *criterion = -1;
  *p0 = *p1 = 0.0;
  try {
    const std::string type = pin.GetOrAddString("refinement", "type", "unset");
    if (type == "pressure_gradient") {
      *criterion = APK_TAG_PRESSURE_GRADIENT;
      *p0 = pin.GetOrAddReal("refinement", "threshold_pressure_gradient", 0.0);
      if (!(*p0 > 0.)) throw std::runtime_error("Make sure to set refinement/threshold_pressure_gradient >0.");
    } else if (type == "xyvelocity_gradient") {
      *criterion = APK_TAG_VELOCITY_GRADIENT;
      *p0 = pin.GetOrAddReal("refinement", "threshold_xyvelocity_gradient", 0.0);
      if (!(*p0 > 0.)) throw std::runtime_error("Make sure to set refinement/threshold_xyvelocity_gradient >0.");
    } else if (type == "maxdensity") {
      *criterion = APK_TAG_MAX_DENSITY;
      *p1 = pin.GetOrAddReal("refinement", "maxdensity_deref_below", 0.0);
      *p0 = pin.GetOrAddReal("refinement", "maxdensity_refine_above", 0.0);
      if (!(*p1 > 0.)) throw std::runtime_error("Make sure to set refinement/maxdensity_deref_below > 0.");
      if (!(*p0 > 0.)) throw std::runtime_error("Make sure to set refinement/maxdensity_refine_above > 0.");
      if (!(*p1 < *p0)) throw std::runtime_error("Make sure to set refinement/maxdensity_deref_below < refinement/maxdensity_refine_above");
    } else {
      throw std::runtime_error("refinement/type is unset: no refinement criterion to evaluate");
    }
  } catch (const std::exception &e) {
    return fail(s, APK_ERR_INVALID, e.what());
  }
  return APK_OK;
}

// Apply per-block tags (+1 refine / -1 derefine / 0) to the tree -- Parthenon's
// MeshRefinement::CheckRefinementCondition + Mesh::UpdateMeshBlockTree: refinement keeps the 2:1
// balance by refining coarser neighbours first; a block asks for derefinement only after
// derefine_count consecutive -1 tags, and 2^ndim siblings merge only if all of them ask and the
// merged block would not touch a block two levels finer.  Returns whether the tree changed.
bool amr_update_tree(apk_sim *s, const std::vector<int> &tags, bool allow_derefine) {
  AmrTree &t = *s->amr;
  const std::vector<AmrLeaf> old = t.leaves;
  const long long mods_before = t.modifications;
  bool changed = false;
  for (int lb = 0; lb < (int)old.size(); ++lb) {
    if (tags[lb] < 0 && allow_derefine) t.SetDerefCount(lb, old[lb].deref_count + 1);
    else t.SetDerefCount(lb, 0);
  }
  for (int lb = 0; lb < (int)old.size(); ++lb)
    if (tags[lb] > 0 && old[lb].level < t.max_level) t.RefineBalanced(old[lb].level, old[lb].lx);
  if (allow_derefine) {
    std::unordered_set<uint64_t> seen;
    for (int lb = 0; lb < (int)old.size(); ++lb) {
      const AmrLeaf &l = old[lb];
      if (l.level == 0 || tags[lb] >= 0) continue;
      const int plx[3] = {l.lx[0] >> 1, l.lx[1] >> 1, l.lx[2] >> 1};
      const uint64_t pkey = AmrTree::Key(l.level - 1, plx);
      if (!seen.insert(pkey).second) continue;
      bool all_ready = true;
      t.ForEachChild(plx, [&](const int *, const int cl[3]) {
        auto it = t.leafmap.find(AmrTree::Key(l.level, cl));
        if (it == t.leafmap.end() || it->second.deref_count < s->amr_derefine_count) all_ready = false;
      });
      if (all_ready && t.CanMerge(l.level - 1, plx)) {
        t.Merge(l.level - 1, plx);
        s->amr_derefined += 1;
      }
    }
  }
  // (the common case, every cycle of a run: nothing split, nothing merged -- `leaves` and `index` are current)
  if (t.modifications == mods_before) return false;
  for (const AmrLeaf &l : old) {
    if (!t.leafmap.count(AmrTree::Key(l.level, l.lx))) changed = true;
    if (t.internal.count(AmrTree::Key(l.level, l.lx))) s->amr_refined += 1;
  }
  t.Reindex();
  return changed;
}

// Move the state from the old block list (and its distribution over ranks) to the new one:
// surviving blocks are copied, new fine blocks are prolongated from their parent (through their
// coarse buffer), merged blocks collect their children's restricted interiors; whatever changes
// rank travels in one message per peer.  Then everything that depends on the block list is rebuilt.
int amr_transfer(apk_sim *s, const std::vector<AmrLeaf> &old, const AmrPartition &old_part) {
  // (a new fine block is prolongated out of its parent's octant PLUS cng cells all round, edges and corners of
  // the parent's ghost zones included: complete them on the old mesh -- its plans and packs are still in place)
  SIM_TRY(s, sync_ghosts(s));
  const AmrGeom &g = s->amr_geom;
  AmrTree &t = *s->amr;
  const int rank = s->rank;
  std::unordered_map<uint64_t, int> old_index;
  for (int n = 0; n < (int)old.size(); ++n) old_index[AmrTree::Key(old[n].level, old[n].lx)] = n;
  AmrPartition new_part;
  new_part.Build((int)t.leaves.size(), s->nranks);
  // the children's restricted interiors of the old mesh (ConsToPrim floors may have touched cons since
  // the last exchange)
  for (apk_refine_plan *p : s->amr_dev.restrict_own[s->cur]) SIM_TRY(s, apk_refine_plan_run(s->ctx, p, s->stream));
  // global transfer list: sources are old blocks (RK_OLD_*), destinations new ones
  std::vector<BoxRegion> moves;
  std::vector<AmrRefOp> prol;
  const int zero[3] = {0, 0, 0};
  for (int nb = 0; nb < (int)t.leaves.size(); ++nb) {
    const AmrLeaf &l = t.leaves[nb];
    BoxRegion r;
    auto it = old_index.find(AmrTree::Key(l.level, l.lx));
    if (it != old_index.end()) {
      r.src_kind = RK_OLD_BLOCK, r.src_block = it->second, r.dst_kind = RK_BLOCK, r.dst_block = nb;
      amr_box_region(r, g.fst, zero, g.fst, zero, g.fn, g.nvar);
      moves.push_back(r);
      continue;
    }
    const int plx[3] = {l.lx[0] >> 1, l.lx[1] >> 1, l.lx[2] >> 1};
    it = (l.level > 0) ? old_index.find(AmrTree::Key(l.level - 1, plx)) : old_index.end();
    if (it != old_index.end()) {  // refined: parent octant (+ cng cells around it) -> my coarse buffer
      int slo[3];
      for (int d = 0; d < 3; ++d) slo[d] = g.act[d] ? g.fs[d] + (l.lx[d] & 1) * (g.mb[d] / 2) - g.cng : 0;
      r.src_kind = RK_OLD_BLOCK, r.src_block = it->second, r.dst_kind = RK_COARSE, r.dst_block = nb;
      amr_box_region(r, g.fst, slo, g.cst, zero, g.cn, g.nvar);
      moves.push_back(r);
      if (new_part.Owner(nb) == rank) {
        AmrRefOp op;
        op.kind = APK_RO_PROLONGATE;
        op.level = l.level;
        op.src_kind = RK_COARSE, op.dst_kind = RK_BLOCK;
        op.src_block = op.dst_block = nb - new_part.first[rank];
        op.geom_block = nb;
        for (int d = 0; d < 3; ++d) op.lo[d] = g.cs[d], op.hi[d] = g.ce[d];
        prol.push_back(op);
      }
      continue;
    }
    // merged: children's coarse buffers -> my octants
    bool ok = true;
    t.ForEachChild(l.lx, [&](const int c[3], const int cl[3]) {
      auto ci = old_index.find(AmrTree::Key(l.level + 1, cl));
      if (ci == old_index.end()) {
        ok = false;
        return;
      }
      int dlo[3], ext[3];
      for (int d = 0; d < 3; ++d) {
        ext[d] = g.act[d] ? g.mb[d] / 2 : 1;
        dlo[d] = g.act[d] ? g.fs[d] + c[d] * (g.mb[d] / 2) : 0;
      }
      BoxRegion m;
      m.src_kind = RK_OLD_COARSE, m.src_block = ci->second, m.dst_kind = RK_BLOCK, m.dst_block = nb;
      amr_box_region(m, g.cst, g.cs, g.fst, dlo, ext, g.nvar);
      moves.push_back(m);
    });
    if (!ok) return fail(s, APK_ERR_INVALID, "regridding: a new block has neither itself, its parent nor its children in the old mesh");
  }
  std::vector<BoxRegion> local, pack, unpack;
  s->amr_move.plan = AmrMessages();
  AmrRegisterPeers(moves, old_part, new_part, rank, s->amr_move.plan);
  AmrLocalize(moves, old_part, new_part, rank, s->amr_move.plan, local, pack, unpack);
  SIM_TRY(s, amr_ensure_buffers(s, s->amr_move, "regrid"));
  double *ncons2[2] = {nullptr, nullptr}, *nprim = nullptr, *nflux[3] = {nullptr, nullptr, nullptr}, *ncoarse = nullptr;
  SIM_TRY(s, amr_allocate(s, (size_t)new_part.Count(rank), ncons2, &nprim, nflux, &ncoarse));
  double *ocons = s->d_cons2[s->cur], *ocoarse = s->d_coarse;
  auto base = [&](int kind, int block) -> double * {
    switch (kind) {
    case RK_OLD_BLOCK: return ocons + (int64_t)block * s->nper;
    case RK_OLD_COARSE: return ocoarse + (int64_t)block * g.coarse_doubles;
    case RK_BLOCK: return ncons2[0] + (int64_t)block * s->nper;
    case RK_COARSE: return ncoarse + (int64_t)block * g.coarse_doubles;
    case RK_SEND: return s->amr_move.send[block];
    case RK_RECV: return s->amr_move.recv[block];
    default: return nullptr;
    }
  };
  auto run = [&](const std::vector<BoxRegion> &regions) -> int {
    std::vector<apk_copy_region> regs;
    for (const BoxRegion &r : regions) regs.push_back(to_copy_region(r, base(r.src_kind, r.src_block), base(r.dst_kind, r.dst_block)));
    apk_copy_plan *cp = nullptr;
    int rc = apk_copy_plan_create(s->ctx, regs.data(), (int)regs.size(), &cp);
    if (rc == APK_OK) rc = apk_copy_plan_run(s->ctx, cp, s->stream);
    if (hipStreamSynchronize(hs(s)) != hipSuccess && rc == APK_OK) rc = APK_ERR_DEVICE;
    apk_copy_plan_destroy(cp);
    return rc;
  };
  SIM_TRY(s, run(pack));
  SIM_TRY(s, run(local));
  SIM_TRY(s, amr_exchange_messages(s, s->amr_move));
  SIM_TRY(s, run(unpack));
  // swap in the new arrays
  for (int p = 0; p < 2; ++p) dev_free(s, s->d_cons2[p]);
  dev_free(s, s->d_prim2[0]);
  dev_free(s, s->d_prim2[1]);
  for (int d = 0; d < 3; ++d) dev_free(s, s->d_flux[d]);
  dev_free(s, s->d_coarse);
  s->d_cons2[0] = ncons2[0], s->d_cons2[1] = ncons2[1];
  s->cur = 0, s->u1buf = 1, s->pcur = 0;
  s->d_prim2[0] = nprim, s->d_prim2[1] = nullptr;
  for (int d = 0; d < 3; ++d) s->d_flux[d] = nflux[d];
  s->d_coarse = ncoarse;
  SIM_TRY(s, amr_rebuild(s));
  if (!prol.empty()) {
    std::vector<apk_refine_plan *> plans;
    SIM_TRY(s, amr_make_refine_plans(s, 0, prol, plans));
    int rc = APK_OK;
    for (apk_refine_plan *p : plans) rc = (rc == APK_OK) ? apk_refine_plan_run(s->ctx, p, s->stream) : rc;
    SIM_HIP(s, hipStreamSynchronize(hs(s)));
    for (apk_refine_plan *p : plans) apk_refine_plan_destroy(p);
    if (rc != APK_OK) return rc;
  }
  return APK_OK;
}

// the tags of every leaf of the forest: mine from the device, the others' through a sum reduction.
// In two halves: _begin launches the reduction and its read-back, _end waits and converts -- whatever the
// caller reads back in between (the time-step estimate at the end of a cycle) shares the host round trip.
int amr_tags_begin(apk_sim *s, AmrTagRequest *req) {
  if (s->amr_tags_posted) {  // (reduced with the time-step estimate at the end of the last stage: do_stage)
    s->amr_tags_posted = false;
    req->criterion = s->amr_posted_criterion, req->pending = s->amr_posted_pending, req->p0 = s->amr_posted_p0, req->p1 = s->amr_posted_p1;
    return APK_OK;
  }
  // the criteria difference every cell of the ring [s-1, e+1]^3 (refinement/gradient.cpp:33-36): ghost cells
  // behind edges and corners included, which the stage loop's faces-only exchange leaves stale.  The last
  // stage of a checking cycle exchanges in full or AMR_SHELL_DEPTH (= the criteria's reach) layers deep (do_stage),
  // so this is a no-op there; it is what covers apk_sim_regrid / apk_sim_check_refinement between cycles.
  // (AMR_GHOSTS_SHELL_DIRECT: the shell without the zones behind same-level same-rank faces -- the tag kernel reads those
  // cells from the neighbours' interiors through the face table, as the stages do)
  // (amr_prim_free_cycle: ... and stores of the primitives only what the criterion reads, amr_tag_vars_stored; between
  // cycles, stale primitives are regenerated first)
  if (s->amr_ghost_state == AMR_GHOSTS_FACES || (s->prim_stale && !s->amr_tag_vars_stored)) SIM_TRY(s, sync_ghosts(s));
  SIM_TRY(s, refinement_criterion(s, &req->criterion, &req->p0, &req->p1));
  const int *table = s->amr_ghost_state == AMR_GHOSTS_SHELL_DIRECT ? s->d_face_nbr : nullptr;
  SIM_TRY(s, apk_tag_blocks_begin_skip(s->ctx, s->mu0(), req->criterion, table, &req->pending, s->stream));
  return APK_OK;
}

int amr_tags_end(apk_sim *s, const AmrTagRequest &req, std::vector<int> &tags) {
  const int nlocal = (int)s->mesh.local_gids.size(), first = s->amr_part.first[s->rank];
  std::vector<int> mine(nlocal, 0);
  SIM_TRY(s, apk_tag_blocks_end(s->ctx, nlocal, req.criterion, req.pending, req.p0, req.p1, mine.data(), nullptr, s->stream));
  tags.assign(s->amr->leaves.size(), 0);
  for (int lb = 0; lb < nlocal; ++lb) tags[first + lb] = mine[lb];
  if (s->have_comm && s->nranks > 1) {
    std::vector<double> buf(tags.begin(), tags.end());
    if (s->comm.allreduce_sum(s->comm.user, buf.data(), (int)buf.size()) != 0) return fail(s, APK_ERR_DEVICE, "allreduce_sum failed");
    for (size_t n = 0; n < tags.size(); ++n) tags[n] = (int)std::lround(buf[n]);
  }
  return APK_OK;
}

int amr_global_tags(apk_sim *s, std::vector<int> &tags) {
  AmrTagRequest req;
  SIM_TRY(s, amr_tags_begin(s, &req));
  return amr_tags_end(s, req, tags);
}

// fresh (zeroed) arrays for the current block list; the state is NOT carried over
int amr_reallocate(apk_sim *s) {
  SIM_HIP(s, hipStreamSynchronize(hs(s)));
  for (int p = 0; p < 2; ++p) dev_free(s, s->d_cons2[p]);
  dev_free(s, s->d_prim2[0]);
  dev_free(s, s->d_prim2[1]);
  for (int d = 0; d < 3; ++d) dev_free(s, s->d_flux[d]);
  dev_free(s, s->d_coarse);
  s->d_prim2[1] = nullptr;
  s->cur = 0, s->u1buf = 1, s->pcur = 0;
  AmrPartition part;
  part.Build((int)s->amr->leaves.size(), s->nranks);
  SIM_TRY(s, amr_allocate(s, (size_t)part.Count(s->rank), s->d_cons2, &s->d_prim2[0], s->d_flux, &s->d_coarse));
  return amr_rebuild(s);
}

// Mesh::LoadBalancingAndAdaptiveMeshRefinement for one rank: tag, update the tree, move the data,
// refill ghost zones and primitives on the new mesh
int amr_regrid(apk_sim *s, bool *changed, const AmrTagRequest *posted) {
  *changed = false;
  std::vector<int> tags;
  if (posted) SIM_TRY(s, amr_tags_end(s, *posted, tags));
  else SIM_TRY(s, amr_global_tags(s, tags));
  const std::vector<AmrLeaf> old = s->amr->leaves;
  const AmrPartition old_part = s->amr_part;
  try {
    if (!amr_update_tree(s, tags, true)) return APK_OK;
  } catch (const std::exception &e) {
    return fail(s, APK_ERR_INVALID, e.what());
  }
  if ((int)s->amr->leaves.size() < s->nranks) return fail(s, APK_ERR_INVALID, "fewer meshblocks than ranks");
  SIM_TRY(s, amr_transfer(s, old, old_part));
  SIM_TRY(s, exchange_ghosts(s));
  SIM_TRY(s, fill_derived(s));
  s->prim_stale = false;  // (every cell of the new mesh)
  s->amr_tag_vars_stored = s->amr_tags_posted = false;
  *changed = true;
  return APK_OK;
}


}  // namespace host
}  // namespace apk
