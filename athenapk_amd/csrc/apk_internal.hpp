// apk_internal.hpp -- internal types shared by the kernels and the C-ABI layer.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/apk_amd.h"

namespace apk {

// What a kernel sees of a MeshBlockPack: a device array of per-block descriptors plus the
// (uniform) block geometry.  Interior index bounds follow Parthenon's IndexDomain::interior
// (SURVEY.md App. A.5): [ng, ng+nx-1] in active dimensions, {0} in collapsed ones.
struct PackView {
  const apk_block_desc *blocks;  // device
  int nblocks, nvar, nhydro;
  int ng, ndim;
  int nx1, nx2, nx3;
  int ni, nj, nk;  // extents including ghosts
  int is, ie, js, je, ks, ke;
  int64_t sj, sk, sn;  // element strides of j, k and variable index
};

inline PackView make_view(const apk_pack_desc &d, const apk_block_desc *dev_blocks) {
  PackView v{};
  v.blocks = dev_blocks;
  v.nblocks = d.nblocks;
  v.nhydro = d.nhydro;
  v.nvar = d.nhydro + d.nscalars;
  v.ng = d.ng;
  v.nx1 = d.nx[0];
  v.nx2 = d.nx[1];
  v.nx3 = d.nx[2];
  v.ndim = (d.nx[2] > 1) ? 3 : ((d.nx[1] > 1) ? 2 : 1);
  v.ni = d.nx[0] + 2 * d.ng;
  v.nj = (d.nx[1] > 1) ? d.nx[1] + 2 * d.ng : 1;
  v.nk = (d.nx[2] > 1) ? d.nx[2] + 2 * d.ng : 1;
  v.is = d.ng;
  v.ie = d.ng + d.nx[0] - 1;
  v.js = (d.nx[1] > 1) ? d.ng : 0;
  v.je = (d.nx[1] > 1) ? d.ng + d.nx[1] - 1 : 0;
  v.ks = (d.nx[2] > 1) ? d.ng : 0;
  v.ke = (d.nx[2] > 1) ? d.ng + d.nx[2] - 1 : 0;
  v.sj = d.stride[0] > 0 ? d.stride[0] : v.ni;
  v.sk = d.stride[1] > 0 ? d.stride[1] : v.sj * v.nj;
  v.sn = d.stride[2] > 0 ? d.stride[2] : v.sk * v.nk;
  return v;
}

}  // namespace apk

struct apk_pack {
  apk::PackView view;
  apk_pack_desc desc;  // desc.blocks points into h_blocks
  std::vector<apk_block_desc> h_blocks;
  apk_block_desc *d_blocks = nullptr;
  bool have_flux[3] = {false, false, false};
};

struct apk_ctx {
  int device = -1;
  unsigned *d_flags = nullptr;       // latched APK_FLAG_* bits
  unsigned long long *d_u64 = nullptr;  // [8] scratch words (min-reduction, counters)
  double *d_partial = nullptr;       // reduction partials
  size_t partial_cap = 0;            // in doubles
  unsigned char *d_mark = nullptr;   // FOFC cell marks
  size_t mark_cap = 0;
  void *h_pinned = nullptr;          // 256 B pinned host staging
  // End-of-cycle gather (capi.hip: cycle_gather_kernel): ONE small kernel writes the stage's time-step word, the flag
  // words and the per-block tag criteria straight into pinned host memory and leaves the device words ready for the
  // next cycle (+max / 0) -- instead of two device-to-host copies, two fills and the next cycle's device-to-device reset
  // (5 - 7 us of stream time each; 0.9 ms cycles on refined meshes).  The *_clean flags say what the stream already
  // holds; everything that reduces into those words goes through prepare_dt_word / the tag launch, which clear them.
  unsigned long long *h_pinned_dev = nullptr;  // device address of h_pinned (null: no mapping, copies as before)
  double *h_partial_dev = nullptr;             // device address of h_partial
  unsigned long long *d_tagmax = nullptr;      // per-block criterion maxima of apk_tag_blocks (bit patterns), own buffer
  size_t tagmax_cap = 0;
  // the minimum apk_stage_dt_flags_read last took out of word 4 (the read resets the word): repeated reads, and a following
  // apk_stage_dt_read, return it until the next reduction into the word starts (prepare_dt_word)
  double last_stage_min = 0.0;
  bool last_stage_min_valid = false;
  bool dt_word_clean = false;                  // word 4 holds +max ...
  hipStream_t clean_stream = nullptr;          // ... as of the gather enqueued on this stream (another stream: reset again)
  int tag_words_clean = 0;                     // the first n words of d_tagmax hold 0
  int tags_pending = 0;                        // criteria of this many blocks reduced, not yet in h_partial
  double *h_partial = nullptr;       // pinned host mirror of d_partial (per-block reductions read back every cycle)
  size_t h_partial_cap = 0;
  double *d_du = nullptr;            // fused path: flux-difference accumulator
  size_t du_cap = 0;                 // in doubles
  double *d_mflux = nullptr;         // fused path with passive scalars: mass flux per face, [3][nblocks][sn]
  size_t mflux_cap = 0;
  // optional kernel timing (apk_kernel_timing_*)
  bool timing_on = false;
  struct TimedSpan {
    int slot;
    hipEvent_t start, stop;
  };
  std::vector<TimedSpan> spans;       // recorded, not yet read
  std::vector<hipEvent_t> free_events;
  double timing_ms[APK_T_COUNT] = {0};
  long long timing_n[APK_T_COUNT] = {0};
  char err[512] = {0};
};

// A copy plan's work list: the regions cut into chunks of at most kCopyChunkItems (cell, variable)
// items / kCopyChunkCells cells, one workgroup per chunk -- the boxes of a plan range from 64-cell
// corners to whole faces, and a (largest box) x (regions) grid leaves most workgroups empty.
constexpr int kCopyChunkItems = 1024, kCopyChunkCells = 256;
struct apk_copy_chunk {
  int region, first;  // first item (plain copy) / first cell (ConsToPrim variant) of the chunk
};
struct apk_copy_plan {
  apk_copy_region *d_regions = nullptr;
  int n = 0;
  int64_t max_cells = 0, max_items = 0;  // largest box in cells / in cells x variables
  apk_copy_chunk *d_chunks_items = nullptr, *d_chunks_cells = nullptr;
  int nchunks_items = 0, nchunks_cells = 0;
};

namespace apk {

inline int set_err(apk_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess) {
  if (ctx) {
    if (e != hipSuccess)
      std::snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, hipGetErrorString(e));
    else
      std::snprintf(ctx->err, sizeof(ctx->err), "%s", what);
  }
  return code;
}

#define APK_HIP_TRY(ctx, expr)                                              \
  do {                                                                      \
    hipError_t e__ = (expr);                                                \
    if (e__ != hipSuccess) return apk::set_err((ctx), APK_ERR_DEVICE, #expr, e__); \
  } while (0)

// ---- kernel launchers implemented in the .hip translation units ---------------------
// input of the boundary-plane fluxes taken from the conserved state (flux_kernel.hpp: face_states_from_cons)
struct FluxConsInput {
  int64_t delta;  // from a block's cons array to the array holding the input state (doubles)
  apk_eos eos;
  double eos_gm1, vceil_sq, pfloor_over_gm1;
};
// flux arrays path (one TU per (fluid, riemann) family to keep compile times parallel)
int launch_fluxes_euler_hlle(const PackView &pv, int recon, double gamma, double c_h,
                             hipStream_t s, int faces = 0, const int *face_list = nullptr, int nlist = 0,
                             const FluxConsInput *from_cons = nullptr);
int launch_fluxes_euler_hllc(const PackView &pv, int recon, double gamma, double c_h,
                             hipStream_t s, int faces = 0, const int *face_list = nullptr, int nlist = 0,
                             const FluxConsInput *from_cons = nullptr);
int launch_fluxes_mhd_hlle(const PackView &pv, int recon, double gamma, double c_h,
                           hipStream_t s, int faces = 0, const int *face_list = nullptr, int nlist = 0,
                             const FluxConsInput *from_cons = nullptr);
int launch_fluxes_mhd_hlld(const PackView &pv, int recon, double gamma, double c_h,
                           hipStream_t s, int faces = 0, const int *face_list = nullptr, int nlist = 0,
                             const FluxConsInput *from_cons = nullptr);
// (dc,none) and (dc,llf = CalculateFluxesTight) for both fluids
int launch_fluxes_misc(const PackView &pv, int fluid, int riemann, double gamma, double c_h,
                       hipStream_t s);

int launch_update_flux_div(const PackView &u0, const PackView &u1, double gam0, double gam1,
                           double beta_dt, hipStream_t s);
int launch_dedner(const PackView &pv, int extended, double coeff, double beta_dt,
                  hipStream_t s);
int launch_cons_to_prim(const PackView &pv, int fluid, const apk_eos &eos, unsigned *d_flags,
                        hipStream_t s, bool ghosts_only = false, const unsigned *late_regions = nullptr,
                        int part = 0, bool faces_only = false, const int *face_nbr = nullptr,
                        unsigned long long *dt_bits = nullptr, int depth = -1, unsigned store_vars = ~0u);
int tag_from_cons_kchunks(const PackView &pv);
int launch_tag_pgrad_from_cons(const PackView &pv, int fluid, const apk_eos &eos, unsigned *d_flags, unsigned long long *dt_bits,
                               unsigned long long *block_max, const int *face_nbr, hipStream_t s);
int launch_min_dt(const PackView &pv, int fluid, double gamma, unsigned long long *d_min_bits,
                  hipStream_t s);
int launch_history(const PackView &pv, int fluid, double *d_partial, int *nblocks_out,
                   double *d_out8, hipStream_t s);
int launch_fofc_mark(const PackView &u0, const PackView &u1, int fluid, double gam0,
                     double gam1, double beta_dt, int attempt, unsigned char *d_mark,
                     unsigned long long *d_count, hipStream_t s);
int launch_count_unphysical(const PackView &u0, int fluid, unsigned long long *d_count, hipStream_t s);
int launch_fofc_fix(const PackView &u0, int fluid, double gamma, double c_h,
                    const unsigned char *d_mark, hipStream_t s);
int launch_copy_regions(const apk_copy_plan &plan, hipStream_t s, int c2p_fluid = 0, const apk_eos *eos = nullptr,
                        unsigned *d_flags = nullptr, int64_t prim_delta = 0, bool prim_only = false);
// fused stage path (fused_dispatch.hip)
int launch_stage_fused(apk_ctx *ctx, const PackView &u0, const PackView &u1,
                       const apk_stage_args &a, double dedner_coeff, hipStream_t s);

// +max into the stage's time-step word (word 4) before a kernel reduces into it -- unless the stream already holds it
int prepare_dt_word(apk_ctx *ctx, hipStream_t s);
// the gather described at apk_ctx::h_pinned_dev: time-step word + flag words (+ pending tag criteria) to the host
int launch_cycle_gather(apk_ctx *ctx, hipStream_t s);

// RAII span: records start/stop events around the launches issued in its scope
struct ScopedTiming {
  apk_ctx *ctx;
  hipStream_t s;
  int idx = -1;
  ScopedTiming(apk_ctx *c, int slot, hipStream_t st) : ctx(c), s(st) {
    if (!ctx || !ctx->timing_on) return;
    auto grab = [&]() {
      hipEvent_t e = nullptr;
      if (!ctx->free_events.empty()) {
        e = ctx->free_events.back();
        ctx->free_events.pop_back();
      } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
      }
      return e;
    };
    apk_ctx::TimedSpan sp{slot, grab(), grab()};
    if (!sp.start || !sp.stop) return;
    (void)hipEventRecord(sp.start, s);
    ctx->spans.push_back(sp);
    idx = (int)ctx->spans.size() - 1;
  }
  ~ScopedTiming() {
    if (idx >= 0) (void)hipEventRecord(ctx->spans[idx].stop, s);
  }
};

// Thread -> (i, j) offset inside an ni x nj index rectangle for the (64, 4) workgroups of the
// per-cell kernels.  Rows of at least 48 cells: 64 lanes along i, 4 rows per workgroup (every load
// a full 512-B segment).  Shorter rows -- the 8^3 / 16^3 meshblocks of refined meshes -- are
// flattened row after row, so that the lanes stay busy; launch with rect_grid().
#ifdef __HIPCC__
__device__ __forceinline__ bool rect_ij(int ni, int nj, int &io, int &jo) {
  if (ni >= 48) {
    io = blockIdx.x * 64 + threadIdx.x;
    jo = blockIdx.y * 4 + threadIdx.y;
  } else {
    const int f = blockIdx.y * 256 + threadIdx.y * 64 + threadIdx.x;
    jo = f / ni;
    io = f - jo * ni;
  }
  return io < ni && jo < nj;
}
inline dim3 rect_grid(int ni, int nj, int nz) {
  return ni >= 48 ? dim3((ni + 63) / 64, (nj + 3) / 4, nz) : dim3(1, (ni * nj + 255) / 256, nz);
}
#endif

}  // namespace apk
