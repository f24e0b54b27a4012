// mesh.hpp -- uniform mesh of equal meshblocks: Morton-ordered partition over ranks and the
// ghost-zone exchange plan (pure host logic; no HIP).  Stands in for the parts of Parthenon's
// Mesh / bvals-in-one that the hot path's standalone driver needs (un-vendored upstream;
// semantics per SURVEY.md App. A.5-A.6): each of the up-to-26 neighbour regions of a block
// is one strided box copy -- same-rank neighbours copy directly, other ranks go through one
// contiguous message buffer per peer; physical boundaries are applied afterwards in the
// order x1, x2, x3 over the entire transverse extent.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <vector>

namespace apk {

enum BcKind { BC_PERIODIC = 0, BC_OUTFLOW = 1, BC_REFLECT = 2 };
enum RegionKind { RK_BLOCK = 0, RK_SEND = 1, RK_RECV = 2 };
// PH_PACK_THIN / PH_UNPACK_THIN: the same messages ONE layer deep (kThinDepth) -- all that the exchange in front of a
// donor-cell stage has to deliver (VL2: the one at the end of a cycle); same buffers, a prefix of them.
// PH_*_NOX1: the same plans without the regions of x1 FACES (offset (+-1, 0, 0)) -- the segments keep their places in the
// messages -- for exchanges whose x1 strips never pass through a copy kernel: the finishing kernel of the stage stores
// its x1 boundary columns straight into the send buffers, the next stage's kernels read their x1 ghost columns straight
// from the receive buffers (apk_stage_args.x1_halo; Mesh::x1_send / x1_recv say where).  A row of an x1 strip is
// nghost doubles: 24 of every 128-byte line a copy kernel fetches from (or writes into) the block.
enum PlanPhase { PH_LOCAL = 0, PH_PACK = 1, PH_UNPACK = 2, PH_BC1 = 3, PH_BC2 = 4, PH_BC3 = 5, PH_PACK_THIN = 6, PH_UNPACK_THIN = 7,
                 PH_PACK_NOX1 = 8, PH_UNPACK_NOX1 = 9, PH_PACK_THIN_NOX1 = 10, PH_UNPACK_THIN_NOX1 = 11, PH_COUNT = 12 };
constexpr int kThinDepth = 1;

struct BoxRegion {
  int src_kind = 0, src_block = 0, dst_kind = 0, dst_block = 0;
  int64_t src_off = 0, dst_off = 0;
  int ext[3] = {1, 1, 1};
  int nvar = 0, flip_var = -1;
  int64_t src_stride[4] = {0, 0, 0, 0}, dst_stride[4] = {0, 0, 0, 0};
  int corner = 0;  // refined meshes: fills a block's ghost zone behind an EDGE or a CORNER (nothing in the stage loop reads it)
  int same_face = 0;  // refined meshes: fills the ghost zone behind a FACE from the interior of a block of the same level
                      // (not needed by a stage that reads that block directly, apk_stage_args.face_neighbor)
};

// where the x1 strip of a local block's lower / upper x1 face sits in a message (Mesh::x1_send / x1_recv)
struct X1Segment {
  int peer = -1;                 // index into Mesh::peers (-1: that face has no neighbour on another rank)
  int64_t off = 0, off_thin = 0;  // element offset of the segment in the peer's full / one-layer message
};

struct PeerPlan {
  int rank = -1;
  int64_t send_count = 0, recv_count = 0;
  int64_t send_count_thin = 0, recv_count_thin = 0;  // the one-layer form (PH_PACK_THIN / PH_UNPACK_THIN)
};

struct Mesh {
  // inputs
  int nx[3] = {1, 1, 1}, mb[3] = {1, 1, 1}, ng = 2, nvar = 5;
  int bc_in[3] = {0, 0, 0}, bc_out[3] = {0, 0, 0};
  int rank = 0, nranks = 1;
  // One-GPU rehearsal of a rank of the 2 x 2 x 2 run (apk_amd/rehearse_remote_faces): a neighbour reached across the
  // periodic boundary of the mesh counts as a block of ANOTHER rank -- pseudo rank nranks + (wrap code), one per
  // combination of wrapped directions: the 7 peers (3 faces, 3 edges, 1 corner) a brick has in the 2 x 2 x 2 rank
  // grid -- so that its ghost zones travel through pack -> message -> unpack like those between bricks.  The
  // messages are delivered by the loopback transport (comm_rccl.cpp): with the sender's and the receiver's
  // segment lists ordered by the same key, a rank's own send buffer IS the message its periodic image would send.
  int rehearse = 0;
  // Row pitch of the block arrays in doubles (0: the natural Ni) and the doubles in front of a block's first cell in its
  // slot (apk_pack_desc.stride; sim.cpp "apk_amd/row_pitch"): with pitch a multiple of 16 and lead = (-nghost) mod 16 the
  // first interior cell of every row sits on a 128-byte boundary.  Everything on the host addresses cells through sj / sk /
  // sn, never through ni / nj.
  int pitch = 0, lead = 0;
  // derived
  int nb[3] = {1, 1, 1}, nblocks_total = 1, ndim = 1;
  int ni = 1, nj = 1, nk = 1, is = 0, ie = 0, js = 0, je = 0, ks = 0, ke = 0;
  int64_t sj = 1, sk = 1, sn = 1;
  std::vector<int> gid_rank;    // owner of every global block id
  std::vector<int> local_gids;  // this rank's blocks, Morton order
  std::map<int, int> gid_local; // gid -> local index
  std::vector<PeerPlan> peers;
  std::vector<BoxRegion> plan[PH_COUNT];
  // per local block and x1 side (0 lower, 1 upper): the segment its interior strip next to that face is packed into /
  // the segment its ghost strip behind that face is unpacked from.  Segment layout (Compact): [var][k][j][depth], the
  // interior extent in x2 and x3, depth = nghost (full) or kThinDepth (one-layer) columns counted in x1 order.
  std::vector<std::array<X1Segment, 2>> x1_send, x1_recv;

  static uint64_t Morton(unsigned x, unsigned y, unsigned z) {
    uint64_t m = 0;
    for (int b = 0; b < 21; ++b) {
      m |= (uint64_t)((x >> b) & 1u) << (3 * b);
      m |= (uint64_t)((y >> b) & 1u) << (3 * b + 1);
      m |= (uint64_t)((z >> b) & 1u) << (3 * b + 2);
    }
    return m;
  }
  int Gid(const int bc[3]) const { return bc[0] + nb[0] * (bc[1] + nb[1] * bc[2]); }
  void Loc(int gid, int bc[3]) const {
    bc[0] = gid % nb[0];
    bc[1] = (gid / nb[0]) % nb[1];
    bc[2] = gid / (nb[0] * nb[1]);
  }
  bool Active(int d) const { return mb[d] > 1; }

  void Build() {
    for (int d = 0; d < 3; ++d) {
      if (mb[d] < 1 || nx[d] % mb[d] != 0) throw std::runtime_error("mesh size must be a multiple of the meshblock size");
      nb[d] = nx[d] / mb[d];
      if (mb[d] == 1 && nx[d] != 1) throw std::runtime_error("meshblock size 1 requires mesh size 1 in that direction");
      if (Active(d) && mb[d] < ng) throw std::runtime_error("meshblock smaller than nghost");
    }
    if (!Active(0)) throw std::runtime_error("x1 must be an active dimension");
    if (!Active(1) && Active(2)) throw std::runtime_error("nx2 == 1 requires nx3 == 1");
    ndim = Active(2) ? 3 : (Active(1) ? 2 : 1);
    nblocks_total = nb[0] * nb[1] * nb[2];
    if (nranks < 1 || rank < 0 || rank >= nranks) throw std::runtime_error("bad rank / nranks");
    if (rehearse && nranks != 1) throw std::runtime_error("apk_amd/rehearse_remote_faces is a one-rank rehearsal");
    if (nblocks_total < nranks) throw std::runtime_error("fewer meshblocks than ranks");
    ni = mb[0] + 2 * ng;
    nj = Active(1) ? mb[1] + 2 * ng : 1;
    nk = Active(2) ? mb[2] + 2 * ng : 1;
    is = ng;
    ie = ng + mb[0] - 1;
    js = Active(1) ? ng : 0;
    je = Active(1) ? ng + mb[1] - 1 : 0;
    ks = Active(2) ? ng : 0;
    ke = Active(2) ? ng + mb[2] - 1 : 0;
    if (pitch != 0 && pitch < ni) throw std::runtime_error("row pitch smaller than a row");
    sj = pitch > 0 ? pitch : ni;
    sk = sj * nj;
    sn = sk * nk;

    // Morton-ordered contiguous ranges, balanced to within one block
    std::vector<std::pair<uint64_t, int>> order;
    for (int g = 0; g < nblocks_total; ++g) {
      int bc[3];
      Loc(g, bc);
      order.push_back({Morton(bc[0], bc[1], bc[2]), g});
    }
    std::sort(order.begin(), order.end());
    gid_rank.assign(nblocks_total, 0);
    local_gids.clear();
    gid_local.clear();
    for (int p = 0; p < nblocks_total; ++p) {
      const int r = (int)(((int64_t)p * nranks) / nblocks_total);
      gid_rank[order[p].second] = r;
      if (r == rank) {
        gid_local[order[p].second] = (int)local_gids.size();
        local_gids.push_back(order[p].second);
      }
    }
    BuildPlans();
  }

  // neighbour of block bc at offset o; false if it lies beyond a non-periodic boundary
  bool Neighbor(const int bc[3], const int o[3], int out[3]) const {
    for (int d = 0; d < 3; ++d) {
      int c = bc[d] + o[d];
      if (c < 0) {
        if (bc_in[d] != BC_PERIODIC) return false;
        c += nb[d];
      } else if (c >= nb[d]) {
        if (bc_out[d] != BC_PERIODIC) return false;
        c -= nb[d];
      }
      out[d] = c;
    }
    return true;
  }

  // rank that owns the neighbour of block bc at offset o (Neighbor(bc, o, nbc) was true)
  int NeighborRank(const int bc[3], const int o[3], const int nbc[3]) const {
    if (rehearse) {
      int wrap = 0;
      for (int d = 0; d < 3; ++d)
        if (bc[d] + o[d] < 0 || bc[d] + o[d] >= nb[d]) wrap |= 1 << d;
      if (wrap) return nranks + wrap - 1;
    }
    return gid_rank[Gid(nbc)];
  }

  // are the ghost zones of local block lb on side (-1 / +1) of direction d filled only when the
  // exchange completes (neighbour on another rank, or a physical boundary condition) rather than
  // by a same-rank copy?
  bool LateFace(int lb, int d, int side) const {
    if (!Active(d)) return false;
    int bc[3], o[3] = {0, 0, 0}, nbc[3];
    Loc(local_gids[lb], bc);
    o[d] = side;
    if (!Neighbor(bc, o, nbc)) return true;  // physical boundary: applied after the unpack
    return NeighborRank(bc, o, nbc) != rank;
  }

  // index range [lo,hi] along dim d of the receiver's ghost region (dst) and of the
  // provider's interior strip (src) for a neighbour at offset o_d
  // (depth: how many layers, counted from the face; 0 = all nghost of them)
  void Range(int d, int o, bool src, int &lo, int &hi, int depth = 0) const {
    const int s = (d == 0) ? is : (d == 1 ? js : ks);
    const int e = (d == 0) ? ie : (d == 1 ? je : ke);
    const int n = (depth > 0 && depth < ng) ? depth : ng;
    if (!Active(d) || o == 0) {
      lo = s;
      hi = e;
    } else if (!src) {
      lo = (o < 0) ? s - n : e + 1;
      hi = (o < 0) ? s - 1 : e + n;
    } else {
      lo = (o < 0) ? e - n + 1 : s;
      hi = (o < 0) ? e : s + n - 1;
    }
  }

  struct Segment {
    int64_t key;   // (receiver gid, offset code)
    int block;     // local block (sender: provider, receiver: destination)
    int o[3];      // offset of the PROVIDER as seen from the receiver
  };

  void BlockStrides(int64_t st[4]) const {
    st[0] = 1;
    st[1] = sj;
    st[2] = sk;
    st[3] = sn;
  }

  void BuildPlans() {
    for (auto &p : plan) p.clear();
    peers.clear();
    x1_send.assign(local_gids.size(), {});
    x1_recv.assign(local_gids.size(), {});
    std::map<int, std::vector<Segment>> sends, recvs;  // by peer rank
    int lo[3], hi[3];
    for (int lb = 0; lb < (int)local_gids.size(); ++lb) {
      int bc[3];
      Loc(local_gids[lb], bc);
      for (int oz = -1; oz <= 1; ++oz)
        for (int oy = -1; oy <= 1; ++oy)
          for (int ox = -1; ox <= 1; ++ox) {
            const int o[3] = {ox, oy, oz};
            if (ox == 0 && oy == 0 && oz == 0) continue;
            if ((!Active(1) && oy != 0) || (!Active(2) && oz != 0)) continue;
            int nbc[3];
            if (!Neighbor(bc, o, nbc)) continue;
            const int ngid = Gid(nbc);
            const int code = (oz + 1) * 9 + (oy + 1) * 3 + (ox + 1);
            const int mo[3] = {-ox, -oy, -oz};
            const int mcode = (mo[2] + 1) * 9 + (mo[1] + 1) * 3 + (mo[0] + 1);
            const int owner = NeighborRank(bc, o, nbc);
            if (owner == rank) {
              // same-rank neighbour: direct box copy into my ghost region
              BoxRegion r;
              r.src_kind = RK_BLOCK;
              r.src_block = gid_local.at(ngid);
              r.dst_kind = RK_BLOCK;
              r.dst_block = lb;
              r.nvar = nvar;
              BlockStrides(r.src_stride);
              BlockStrides(r.dst_stride);
              r.src_off = r.dst_off = 0;
              for (int d = 0; d < 3; ++d) {
                Range(d, o[d], false, lo[d], hi[d]);
                r.ext[d] = hi[d] - lo[d] + 1;
                r.dst_off += lo[d] * r.dst_stride[d];
                int slo, shi;
                Range(d, o[d], true, slo, shi);
                r.src_off += slo * r.src_stride[d];
              }
              plan[PH_LOCAL].push_back(r);
            } else {
              const int peer = owner;
              // I receive my ghost region at offset o from block ngid ...
              recvs[peer].push_back({(int64_t)local_gids[lb] * 27 + code, lb, {ox, oy, oz}});
              // ... and block ngid receives, at its offset -o, a strip of my interior
              sends[peer].push_back({(int64_t)ngid * 27 + mcode, lb, {mo[0], mo[1], mo[2]}});
            }
          }
    }
    std::vector<int> peer_ranks;
    for (auto &kv : recvs) peer_ranks.push_back(kv.first);
    for (auto &kv : sends)
      if (!recvs.count(kv.first)) peer_ranks.push_back(kv.first);
    std::sort(peer_ranks.begin(), peer_ranks.end());
    for (int pr : peer_ranks) {
      PeerPlan pp;
      pp.rank = pr;
      const int pidx = (int)peers.size();
      auto bykey = [](const Segment &a, const Segment &b) { return a.key < b.key; };
      auto &sv = sends[pr];
      auto &rv = recvs[pr];
      std::sort(sv.begin(), sv.end(), bykey);
      std::sort(rv.begin(), rv.end(), bykey);
      for (const auto &sgm : sv) {  // pack: my interior strip facing the receiver
        BoxRegion r;
        r.src_kind = RK_BLOCK;
        r.src_block = sgm.block;
        r.dst_kind = RK_SEND;
        r.dst_block = pidx;
        r.nvar = nvar;
        BlockStrides(r.src_stride);
        // sgm.o = my offset as seen from the receiver => the strip is the "src" range for it
        for (int d = 0; d < 3; ++d) {
          Range(d, sgm.o[d], true, lo[d], hi[d]);
          r.ext[d] = hi[d] - lo[d] + 1;
          r.src_off += lo[d] * r.src_stride[d];
        }
        Compact(r.ext, r.dst_stride);
        r.dst_off = pp.send_count;
        pp.send_count += (int64_t)r.ext[0] * r.ext[1] * r.ext[2] * nvar;
        plan[PH_PACK].push_back(r);
        // (I am the receiver's neighbour at sgm.o: at its upper x1 side I send my LOWER strip)
        const bool x1face = sgm.o[0] != 0 && sgm.o[1] == 0 && sgm.o[2] == 0;
        if (!x1face) plan[PH_PACK_NOX1].push_back(r);
        BoxRegion t = r;  // the one-layer form of the same strip
        t.src_off = 0;
        for (int d = 0; d < 3; ++d) {
          Range(d, sgm.o[d], true, lo[d], hi[d], kThinDepth);
          t.ext[d] = hi[d] - lo[d] + 1;
          t.src_off += lo[d] * t.src_stride[d];
        }
        Compact(t.ext, t.dst_stride);
        t.dst_off = pp.send_count_thin;
        pp.send_count_thin += (int64_t)t.ext[0] * t.ext[1] * t.ext[2] * nvar;
        plan[PH_PACK_THIN].push_back(t);
        if (!x1face) plan[PH_PACK_THIN_NOX1].push_back(t);
        else x1_send[sgm.block][sgm.o[0] > 0 ? 0 : 1] = X1Segment{pidx, r.dst_off, t.dst_off};
      }
      for (const auto &sgm : rv) {  // unpack into my ghost region at offset sgm.o
        BoxRegion r;
        r.src_kind = RK_RECV;
        r.src_block = pidx;
        r.dst_kind = RK_BLOCK;
        r.dst_block = sgm.block;
        r.nvar = nvar;
        BlockStrides(r.dst_stride);
        for (int d = 0; d < 3; ++d) {
          Range(d, sgm.o[d], false, lo[d], hi[d]);
          r.ext[d] = hi[d] - lo[d] + 1;
          r.dst_off += lo[d] * r.dst_stride[d];
        }
        Compact(r.ext, r.src_stride);
        r.src_off = pp.recv_count;
        pp.recv_count += (int64_t)r.ext[0] * r.ext[1] * r.ext[2] * nvar;
        plan[PH_UNPACK].push_back(r);
        const bool x1face = sgm.o[0] != 0 && sgm.o[1] == 0 && sgm.o[2] == 0;
        if (!x1face) plan[PH_UNPACK_NOX1].push_back(r);
        BoxRegion t = r;
        t.dst_off = 0;
        for (int d = 0; d < 3; ++d) {
          Range(d, sgm.o[d], false, lo[d], hi[d], kThinDepth);
          t.ext[d] = hi[d] - lo[d] + 1;
          t.dst_off += lo[d] * t.dst_stride[d];
        }
        Compact(t.ext, t.src_stride);
        t.src_off = pp.recv_count_thin;
        pp.recv_count_thin += (int64_t)t.ext[0] * t.ext[1] * t.ext[2] * nvar;
        plan[PH_UNPACK_THIN].push_back(t);
        if (!x1face) plan[PH_UNPACK_THIN_NOX1].push_back(t);
        else x1_recv[sgm.block][sgm.o[0] < 0 ? 0 : 1] = X1Segment{pidx, r.src_off, t.src_off};
      }
      peers.push_back(pp);
    }
    // physical boundaries: inner then outer face of each active dimension, entire transverse
    for (int lb = 0; lb < (int)local_gids.size(); ++lb) {
      int bc[3];
      Loc(local_gids[lb], bc);
      const int ext_all[3] = {ni, nj, nk};
      const int s3[3] = {is, js, ks}, e3[3] = {ie, je, ke};
      for (int d = 0; d < 3; ++d) {
        if (!Active(d)) continue;
        for (int side = 0; side < 2; ++side) {
          const bool edge = side ? (bc[d] == nb[d] - 1) : (bc[d] == 0);
          const int kind = side ? bc_out[d] : bc_in[d];
          if (!edge || kind == BC_PERIODIC) continue;
          BoxRegion r;
          r.src_kind = r.dst_kind = RK_BLOCK;
          r.src_block = r.dst_block = lb;
          r.nvar = nvar;
          BlockStrides(r.src_stride);
          BlockStrides(r.dst_stride);
          for (int q = 0; q < 3; ++q) r.ext[q] = ext_all[q];
          r.ext[d] = ng;
          const int g0 = side ? e3[d] + 1 : 0;  // first ghost index
          r.dst_off = g0 * r.dst_stride[d];
          if (kind == BC_OUTFLOW) {  // copy the last active cell (docs/input.md:416-419)
            r.src_off = (side ? e3[d] : s3[d]) * r.src_stride[d];
            r.src_stride[d] = 0;
          } else {  // reflecting (src/bvals/boundary_conditions_apk.hpp:38-85)
            // ghost g0+m mirrors interior: inner: 2*s-1-(g0+m); outer: 2*e+1-(g0+m)
            const int first = side ? (2 * e3[d] + 1 - g0) : (2 * s3[d] - 1 - g0);
            r.src_off = first * r.src_stride[d];
            r.src_stride[d] = -r.src_stride[d];
            r.flip_var = 1 + d;  // normal momentum IM1+d
          }
          plan[PH_BC1 + d].push_back(r);
        }
      }
    }
  }

  static void Compact(const int ext[3], int64_t st[4]) {
    st[0] = 1;
    st[1] = ext[0];
    st[2] = (int64_t)ext[0] * ext[1];
    st[3] = (int64_t)ext[0] * ext[1] * ext[2];
  }
};

}  // namespace apk
