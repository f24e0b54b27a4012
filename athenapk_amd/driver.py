"""Python handle on the native host driver (include/apk_host.h) plus the two pieces of
plumbing torch provides: device memory for the fields / message buffers and
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests) for the
per-stage halo exchange and the tiny min/sum all-reduces (src/hydro/hydro.cpp:122-128).
"""
import ctypes as C

import os

import numpy as np
import torch

from . import lib as L


class HaloExchanger:
    """One grouped send/recv per peer rank per stage (ncclGroupStart .. ncclSend/ncclRecv ..
    ncclGroupEnd under the hood).  Works on whatever device the buffers live on, so the same
    code runs over RCCL on GPUs and over gloo on CPU in the tests."""

    def __init__(self, peers, group=None):
        # peers: list of (rank, send_tensor, recv_tensor)
        self.peers = list(peers)
        self.group = group
        self._ops = None  # the P2POp list of the direct (not host-staged) path, built once

    def begin(self):
        """post every send / receive of this stage's halo messages; returns without waiting"""
        import torch.distributed as dist
        self._reqs, self._back = [], []
        if not self.peers:
            return
        # gloo has no device send/recv: development runs that put several ranks on one GPU
        # (tests, `APK_DIST_BACKEND=gloo bench.py`) bounce the messages through host memory
        some = next((t for _, a, b in self.peers for t in (a, b) if t is not None), None)
        staged = some is not None and some.is_cuda and dist.get_backend(self.group) == "gloo"
        if not staged and self._ops is not None:  # same buffers, same peers every stage
            self._reqs = dist.batch_isend_irecv(self._ops) if self._ops else []
            return
        ops = []
        for rank, send_t, recv_t in self.peers:  # (a direction without data has no tensor: no op on either side)
            if staged and recv_t is not None:
                host_recv = torch.empty(recv_t.shape, dtype=recv_t.dtype)
                self._back.append((recv_t, host_recv))
                recv_t = host_recv
            if staged and send_t is not None:
                send_t = send_t.cpu()
            if recv_t is not None:
                ops.append(dist.P2POp(dist.irecv, recv_t, rank, group=self.group))
            if send_t is not None:
                ops.append(dist.P2POp(dist.isend, send_t, rank, group=self.group))
        if not staged:
            self._ops = ops
        self._reqs = dist.batch_isend_irecv(ops) if ops else []

    def end(self):
        """work enqueued on the current stream after this call sees the received data (over RCCL
        Work.wait() is a stream dependency, not a host wait)"""
        for req in self._reqs:
            req.wait()
        for dev_t, host_t in self._back:
            dev_t.copy_(host_t)
        self._reqs, self._back = [], []

    def exchange(self):
        self.begin()
        self.end()


def _allreduce(vals_ptr, n, op, device, group=None):
    import torch.distributed as dist
    a = np.ctypeslib.as_array(vals_ptr, shape=(n,))
    if dist.get_backend(group) == "gloo":
        device = "cpu"
    t = torch.from_numpy(a.copy()).to(device)
    dist.all_reduce(t, op=op, group=group)
    a[:] = t.cpu().numpy()


class _FmftHost:
    """Host spectral state of the few-modes turbulence driver (FewModesFT,
    /root/reference src/utils/few_modes_ft.cpp); works on host-only sims too."""

    def fmft_num_modes(self):
        return self.lib.apk_sim_fmft_num_modes(self.h)

    def fmft_var_hat(self):
        out = np.zeros((3, self.fmft_num_modes(), 2))
        rc = self.lib.apk_sim_fmft_var_hat(self.h, out.ctypes.data_as(L.c_dp))
        if rc != L.APK_OK:
            raise L.ApkError(rc, "no turbulence driver in this sim")
        return out

    def fmft_evolve(self, dt):
        rc = self.lib.apk_sim_fmft_evolve(self.h, float(dt))
        if rc != L.APK_OK:
            raise L.ApkError(rc, "no turbulence driver in this sim")

    def fmft_phases(self, axis, n, g0):
        out = np.zeros((2, self.fmft_num_modes(), n))
        rc = self.lib.apk_sim_fmft_phases(self.h, axis, n, g0, out.ctypes.data_as(L.c_dp))
        if rc != L.APK_OK:
            raise L.ApkError(rc, "no turbulence driver in this sim")
        return out


class _MeshView:
    """Block placement and ghost-exchange plans of a sim handle (valid without a GPU)."""

    PHASES = {"local": 0, "pack": 1, "unpack": 2, "bc1": 3, "bc2": 4, "bc3": 5, "pack_thin": 6, "unpack_thin": 7,
              # the pack / unpack plans without the x1 faces (apk_sim_set_x1_direct)
              "pack_nox1": 60, "unpack_nox1": 61, "pack_thin_nox1": 62, "unpack_thin_nox1": 63,
              # refined meshes (src/.. host/amr.hpp): all copies of the multilevel exchange, the
              # physical boundaries of coarse buffers / blocks, the flux-correction copies
              "amr_fill": 10, "amr_coarse_bc1": 11, "amr_coarse_bc2": 12, "amr_coarse_bc3": 13,
              "amr_bc1": 14, "amr_bc2": 15, "amr_bc3": 16, "amr_flux1": 17, "amr_flux2": 18, "amr_flux3": 19,
              # this rank's share of them (local block numbers, message buffers)
              "my_fill": 20, "my_fill_pack": 21, "my_fill_unpack": 22,
              "my_flux1": 25, "my_flux2": 26, "my_flux3": 27, "my_flux_pack1": 28, "my_flux_pack2": 29,
              "my_flux_pack3": 30, "my_flux_unpack1": 31, "my_flux_unpack2": 32, "my_flux_unpack3": 33,
              "my_coarse_bc1": 34, "my_coarse_bc2": 35, "my_coarse_bc3": 36, "my_bc1": 37, "my_bc2": 38, "my_bc3": 39,
              # the exchanges of the stage loop: faces only, faces without same-level same-rank copies, two-layer shell
              "my_fill_faces": 40, "my_fill_pack_faces": 41, "my_fill_unpack_faces": 42, "my_fill_direct": 43,
              "my_fill_shell": 44, "my_fill_pack_shell": 45, "my_fill_unpack_shell": 46,
              "my_bc_shell1": 47, "my_bc_shell2": 48, "my_bc_shell3": 49}
    AMR_OPS = {"restrict_own": 0, "prolongate": 1, "flux_restrict1": 2, "flux_restrict2": 3, "flux_restrict3": 4,
               "my_restrict_own": 10, "my_prolongate": 11, "my_flux_restrict1": 12, "my_flux_restrict2": 13,
               "my_flux_restrict3": 14, "my_prolongate_faces": 15, "my_prolongate_shell": 16}

    def messages(self, which):
        """[(peer rank, send doubles, recv doubles)] of a refined mesh's "halo" / "flux" message set"""
        self.lib.apk_sim_select_messages(self.h, {"uniform": 0, "halo": 1, "flux": 2, "halo_faces": 3, "halo_shell": 4, "uniform_thin": 5}[which])
        out = []
        for p in range(self.lib.apk_sim_num_peers(self.h)):
            pi = L.PeerInfo()
            self.lib.apk_sim_peer(self.h, p, C.byref(pi))
            out.append((pi.rank, pi.send_count, pi.recv_count))
        self.lib.apk_sim_select_messages(self.h, 0)
        return out

    def block_gid(self, lb):
        gid = C.c_int(0)
        loc = (C.c_int * 3)()
        self.lib.apk_sim_block_location(self.h, lb, C.byref(gid), C.byref(loc))
        return gid.value, tuple(loc)

    def block_level(self, lb):
        return self.lib.apk_sim_block_level(self.h, lb)

    def regions(self, phase):
        ph = self.PHASES[phase]
        n = self.lib.apk_sim_plan_size(self.h, ph)
        out = []
        for r in range(max(n, 0)):
            ri = L.RegionInfo()
            self.lib.apk_sim_plan_region(self.h, ph, r, C.byref(ri))
            out.append(ri)
        return out

    def amr_ops(self, which):
        w = self.AMR_OPS[which]
        out = []
        for n in range(max(self.lib.apk_sim_amr_ops_size(self.h, w), 0)):
            oi = L.AmrOpInfo()
            self.lib.apk_sim_amr_op(self.h, w, n, C.byref(oi))
            out.append(oi)
        return out

    def amr_stats(self):
        """(blocks refined, sibling groups merged, deepest level allowed, zone-cycles done)"""
        a, b, c, d = C.c_longlong(0), C.c_longlong(0), C.c_int(0), C.c_longlong(0)
        self.lib.apk_sim_amr_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return a.value, b.value, c.value, d.value

    def peers(self):
        """[(peer rank, doubles sent, doubles received)] of this rank's halo exchange"""
        out = []
        for p in range(self.info.npeers):
            pi = L.PeerInfo()
            self.lib.apk_sim_peer(self.h, p, C.byref(pi))
            out.append((pi.rank, pi.send_count, pi.recv_count))
        return out

    def refresh_info(self):
        self.lib.apk_sim_get_info(self.h, C.byref(self.info))
        return self.info


class Simulation(_FmftHost, _MeshView):
    """apk_sim: deck + overrides -> mesh partition, packs, ghost plans, stage loop (C++).

    COLLECTIVE accessors on N > 1 ranks: read_block / write_block / gather / history / turbulence_history / reldivb and
    the error norms bring the ghost zones of the current state up to date first (apk_host.h: sync_ghosts).  After a
    cycle that ended with a one-layer exchange (set_thin_exchange, the default of periodic uniform VL2 runs) or on a
    refined mesh that completion is a full halo exchange, i.e. a collective: call these on EVERY rank, in the same
    order, like the stage loop itself -- a call on one rank alone blocks in the exchange.  set_thin_exchange(False)
    restores rank-local accessors on uniform meshes.
    """

    def __init__(self, deck, overrides=(), rank=0, nranks=1, strict=False, use_torch_alloc=True,
                 group=None, comm=None):
        """comm (nranks > 1): "rccl" = the native transport of the C++ host (grouped ncclSend /
        ncclRecv on its own HIP stream, csrc/host/comm_rccl.cpp; one GPU per rank; torch.distributed
        only carries the 256-byte bootstrap ids), "torch" = callbacks into torch.distributed
        (any backend; the way several ranks can share one GPU through gloo for tests).  Default:
        APK_COMM, else "rccl" when the process group's backend is nccl, else "torch"."""
        self.lib = L.load(strict)
        self.rank, self.nranks = rank, nranks
        if nranks > 1 and comm is None:
            import torch.distributed as dist
            comm = os.environ.get("APK_COMM") or ("rccl" if dist.get_backend(group) == "nccl" else "torch")
        self.comm_kind = comm if nranks > 1 else None
        self._tensors = {}   # ptr -> tensor (keeps allocations alive)
        self._by_tag = {}
        self._group = group
        dev = torch.device("cuda", torch.cuda.current_device())
        self._dev = dev
        # The tiny dt / c_h / history reductions go through a host-side (gloo) group: on the RCCL
        # communicator they would queue behind halo messages that are still in flight and make the
        # host wait for them, which is exactly what the overlap avoids.
        red_group = group
        if nranks > 1 and self.comm_kind == "torch":
            import torch.distributed as dist
            if dist.get_backend(group) != "gloo":
                red_group = dist.new_group(backend="gloo")
        self._red_group = red_group

        def _alloc(user, tag, nbytes):
            t = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dev)
            self._tensors[t.data_ptr()] = t
            self._by_tag[tag.decode()] = t
            return t.data_ptr()

        def _release(user, ptr):
            self._tensors.pop(ptr, None)

        self._alloc_cb = L.ALLOC_FN(_alloc)
        self._release_cb = L.RELEASE_FN(_release)
        allocator = L.Allocator(None, self._alloc_cb, self._release_cb)

        self._halo = None
        self._halo_generation = None
        self._halo_cache = {}
        self._cb_counts = {"exchanges": 0, "reductions": 0}  # callback transport (the native one keeps its own)

        def _exchange(user):
            try:
                self._cb_counts["exchanges"] += 1
                self._current_halo().exchange()
                return 0
            except Exception as e:  # surfaced as APK_ERR_DEVICE by the driver
                self._cb_error = e
                return 1

        def _exchange_begin(user):
            try:
                self._cb_counts["exchanges"] += 1
                self._current_halo().begin()
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        def _exchange_end(user):
            try:
                self._halo.end()
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        def _amin(user, vals, n):
            try:
                import torch.distributed as dist
                self._cb_counts["reductions"] += 1
                _allreduce(vals, n, dist.ReduceOp.MIN, dev, red_group)
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        def _asum(user, vals, n):
            try:
                import torch.distributed as dist
                self._cb_counts["reductions"] += 1
                _allreduce(vals, n, dist.ReduceOp.SUM, dev, red_group)
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        self._cb_error = None
        self._ex_cb, self._amin_cb, self._asum_cb = L.EXCHANGE_FN(_exchange), L.ALLREDUCE_FN(_amin), L.ALLREDUCE_FN(_asum)
        self._exb_cb, self._exe_cb = L.EXCHANGE_FN(_exchange_begin), L.EXCHANGE_FN(_exchange_end)
        comm = L.CommOps(None, self._ex_cb, self._amin_cb, self._asum_cb, self._exb_cb, self._exe_cb)

        ov = (C.c_char_p * max(1, len(overrides)))(*[o.encode() for o in overrides])
        err = C.create_string_buffer(1024)
        h = C.c_void_p()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = self.lib.apk_sim_create(deck.encode(), ov, len(overrides), rank, nranks,
                                     C.byref(allocator) if use_torch_alloc else None,
                                     C.byref(comm) if (nranks > 1 and self.comm_kind == "torch") else None, stream,
                                     C.byref(h), err, len(err))
        if rc != L.APK_OK:
            raise L.ApkError(rc, err.value.decode())
        self.h = h
        if self.comm_kind == "rccl":
            # bootstrap: rank 0 makes the two ncclUniqueIds, torch.distributed hands them round
            import torch.distributed as dist
            ids = C.create_string_buffer(2 * L.APK_RCCL_ID_BYTES)
            rc0 = L.APK_OK
            if dist.get_rank(group) == 0:
                # (a failure here -- no librccl -- must not raise before the broadcast: the other ranks are
                # waiting in it; rank 0 hands its status round with the ids and everybody falls back together)
                rc0 = self.lib.apk_rccl_unique_ids(ids, len(ids))
            box = [(rc0, ids.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            rc0, raw = box[0]
            rc = self.lib.apk_sim_comm_rccl(self.h, raw, len(raw)) if rc0 == L.APK_OK else rc0
            # every rank must end up on the same transport: if the native one could not start anywhere
            # (no librccl, communicator creation failed), all ranks fall back to the callbacks -- loudly
            okt = torch.tensor([1 if rc == L.APK_OK else 0], dtype=torch.int32,
                               device=dev if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN, group=group)
            if int(okt.item()) == 0:
                import sys
                why = self.lib.apk_sim_last_error(self.h).decode()
                print("[athenapk_amd] rank %d: native RCCL transport unavailable (%s); using torch.distributed callbacks"
                      % (rank, why or "failed on another rank"), file=sys.stderr, flush=True)
                self.close()
                self.__init__(deck, overrides, rank, nranks, strict, use_torch_alloc, group, comm="torch")
                return
        self.info = L.SimInfo()
        self._check(self.lib.apk_sim_get_info(self.h, C.byref(self.info)))
        self._current_halo()

    def _current_halo(self):
        """the exchanger over the sim's current message set (apk_sim_peer): built once on uniform
        meshes; refined meshes switch between halo / flux-correction / regridding messages and
        change them when the mesh does, which the generation counter tells"""
        gen = self.lib.apk_sim_message_generation(self.h)
        if self._halo is None or gen != self._halo_generation:
            peers, key = [], []
            for p in range(self.lib.apk_sim_num_peers(self.h)):
                pi = L.PeerInfo()
                self._check(self.lib.apk_sim_peer(self.h, p, C.byref(pi)))
                st = self._tensors[pi.send_buf][:pi.send_count] if pi.send_count else None
                rt = self._tensors[pi.recv_buf][:pi.recv_count] if pi.recv_count else None
                peers.append((pi.rank, st, rt))
                key.append((pi.rank, pi.send_buf, pi.send_count, pi.recv_buf, pi.recv_count))
            # (a uniform mesh alternates between its full and its one-layer message set: keep both exchangers)
            key = tuple(key)
            if key not in self._halo_cache:
                if len(self._halo_cache) >= 4:  # (refined meshes change their sets with every regridding)
                    self._halo_cache.clear()
                self._halo_cache[key] = HaloExchanger(peers, self._group)
            self._halo = self._halo_cache[key]
            self._halo_generation = gen
        return self._halo

    def _check(self, rc):
        if rc != L.APK_OK:
            msg = self.lib.apk_sim_last_error(self.h).decode() if self.h else ""
            if self._cb_error is not None:
                msg += " [callback: %r]" % (self._cb_error,)
            raise L.ApkError(rc, msg)

    def close(self):
        if getattr(self, "h", None):
            self.lib.apk_sim_destroy(self.h)
            self.h = None
            self._tensors.clear()
            self._by_tag.clear()
            self._halo = None         # (the exchangers hold slices of the message buffers)
            self._halo_cache.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- driver -----------------------------------------------------------------------
    def set_overlap(self, overlap):
        """overlap the halo exchange between stages with the next stage's x1 sweep (default on)"""
        self._check(self.lib.apk_sim_set_overlap(self.h, int(overlap)))
        return self

    @property
    def loop_seconds(self):
        return self.lib.apk_sim_loop_seconds(self.h)

    @property
    def loop_zone_cycles(self):
        return self.lib.apk_sim_loop_zone_cycles(self.h)

    @property
    def loop_cycles(self):
        return self.lib.apk_sim_loop_cycles(self.h)

    @property
    def overlapped_exchanges(self):
        return self.lib.apk_sim_overlapped_exchanges(self.h)

    def comm_stats(self):
        """{"exchanges", "reductions"}: halo exchanges posted and collectives done so far by this rank's transport"""
        if self.comm_kind == "rccl":
            ex, red = C.c_longlong(0), C.c_longlong(0)
            self._check(self.lib.apk_sim_comm_stats(self.h, C.byref(ex), C.byref(red)))
            return {"exchanges": int(ex.value), "reductions": int(red.value)}
        return dict(self._cb_counts)

    def skipped_local_exchanges(self):
        return self.lib.apk_sim_skipped_local_exchanges(self.h)

    def amr_c2p_passes_skipped(self):
        """ConsToPrim passes between the stages a refined mesh did without (apk_sim_amr_c2p_passes_skipped)"""
        return self.lib.apk_sim_amr_c2p_passes_skipped(self.h)

    def thin_exchanges(self):
        """one-layer exchanges so far (apk_sim_set_thin_exchange)"""
        return self.lib.apk_sim_thin_exchanges(self.h)

    def set_thin_exchange(self, on):
        self._check(self.lib.apk_sim_set_thin_exchange(self.h, int(on)))

    def x1_direct_exchanges(self):
        """exchanges so far whose x1 strips bypassed the pack / unpack kernels (apk_sim_set_x1_direct)"""
        return self.lib.apk_sim_x1_direct_exchanges(self.h)

    def set_x1_direct(self, on):
        self._check(self.lib.apk_sim_set_x1_direct(self.h, int(on)))
        return self

    def set_direct_neighbors(self, on):
        self._check(self.lib.apk_sim_set_direct_neighbors(self.h, int(on)))
        return self

    def set_prim_free(self, on):
        """full-step primitives kept out of memory where the cycle allows it (apk_sim_set_prim_free)"""
        self._check(self.lib.apk_sim_set_prim_free(self.h, int(on)))
        return self

    @property
    def prim_is_stale(self):
        return bool(self.lib.apk_sim_prim_is_stale(self.h))

    def turb_dt_kicks(self):
        """forcing kicks so far that estimated the time step without storing primitives (apk_sim_turb_dt_kicks)"""
        return self.lib.apk_sim_turb_dt_kicks(self.h)

    def set_amr_full_exchange(self, on):
        """refined meshes: 1 = the stage loop exchanges every ghost zone, not only those behind block faces"""
        self._check(self.lib.apk_sim_set_amr_full_exchange(self.h, int(on)))

    def set_fused(self, fused):
        self._check(self.lib.apk_sim_set_fused(self.h, int(fused)))
        self._check(self.lib.apk_sim_get_info(self.h, C.byref(self.info)))

    def initialize(self):
        self._check(self.lib.apk_sim_initialize(self.h))
        return self

    def step(self):
        self._check(self.lib.apk_sim_step(self.h))

    def run(self, nlim=-1):
        n = C.c_int(0)
        self._check(self.lib.apk_sim_run(self.h, nlim, C.byref(n)))
        return n.value

    time = property(lambda s: s.lib.apk_sim_time(s.h))
    dt = property(lambda s: s.lib.apk_sim_dt(s.h))
    tlim = property(lambda s: s.lib.apk_sim_tlim(s.h))
    c_h = property(lambda s: s.lib.apk_sim_c_h(s.h))
    ncycle = property(lambda s: s.lib.apk_sim_ncycle(s.h))
    fofc_count = property(lambda s: s.lib.apk_sim_fofc_count(s.h))

    @property
    def block_shape(self):
        i = self.info
        nk = i.mb[2] + 2 * i.ng if i.mb[2] > 1 else 1
        nj = i.mb[1] + 2 * i.ng if i.mb[1] > 1 else 1
        return (i.nhydro + i.nscalars, nk, nj, i.mb[0] + 2 * i.ng)

    def read_block(self, lb, field="cons"):
        """one local block incl. ghost zones (collective on N > 1 ranks: see the class docstring)"""
        out = np.empty(self.block_shape)
        self._check(self.lib.apk_sim_read_block(self.h, lb, {"cons": 0, "prim": 1, "u1": 2}[field],
                                                out.ctypes.data_as(L.c_dp)))
        return out

    def write_block(self, lb, arr, field="cons"):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self.block_shape
        self._check(self.lib.apk_sim_write_block(self.h, lb, {"cons": 0, "prim": 1, "u1": 2}[field],
                                                 a.ctypes.data_as(L.c_dp)))

    def gather(self, field="cons"):
        i = self.info
        out = np.zeros((i.nhydro + i.nscalars, i.nx[2], i.nx[1], i.nx[0]))
        self._check(self.lib.apk_sim_gather(self.h, {"cons": 0, "prim": 1, "u1": 2}[field],
                                            out.ctypes.data_as(L.c_dp)))
        return out

    @property
    def fofc_fallback_stages(self):
        return self.lib.apk_sim_fofc_fallback_stages(self.h)

    def apply_tags(self, tags):
        """one regridding pass for given per-block tags (+1 / 0 / -1 for every block of the forest)"""
        arr = (C.c_int * len(tags))(*[int(t) for t in tags])
        ch = C.c_int(0)
        self._check(self.lib.apk_sim_amr_apply_tags(self.h, arr, len(tags), C.byref(ch)))
        self.refresh_info()
        return bool(ch.value)

    def regrid(self):
        """one tag -> refine / derefine -> transfer pass; True if the mesh changed"""
        ch = C.c_int(0)
        self._check(self.lib.apk_sim_regrid(self.h, C.byref(ch)))
        self.refresh_info()
        return bool(ch.value)

    def history(self):
        out = (C.c_double * 8)()
        self._check(self.lib.apk_sim_history(self.h, out))
        return np.array(out[:])

    def check_refinement(self):
        """the deck's <refinement> criterion on every local block: (tags, criterion values)"""
        n = self.info.nblocks_local
        tags, crit = (C.c_int * n)(), (C.c_double * n)()
        self._check(self.lib.apk_sim_check_refinement(self.h, tags, crit))
        return np.array(tags[:]), np.array(crit[:])

    def history_labels(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.apk_sim_history_labels(self.h, buf, len(buf)))
        return buf.value.decode().split()

    def write_history(self, path):
        """append one row in Parthenon's .hst layout (header when the file is new)"""
        self._check(self.lib.apk_sim_write_history(self.h, str(path).encode()))

    def write_linear_wave_errors(self, path):
        """append one row of linearwave-errors.dat (src/pgen/linear_wave.cpp:296-334)"""
        self._check(self.lib.apk_sim_write_linear_wave_errors(self.h, str(path).encode()))

    def execute(self, outdir):
        """initialize + main loop + the deck's hst outputs + the linear-wave error file"""
        n = C.c_int(0)
        self._check(self.lib.apk_sim_execute(self.h, str(outdir).encode(), C.byref(n)))
        return n.value

    def turbulence_history(self):
        """volume sums of sonic Mach number, Alfvenic Mach number, plasma beta (TurbulenceHst)"""
        out = (C.c_double * 3)()
        self._check(self.lib.apk_sim_turbulence_history(self.h, out))
        return np.array(out[:])

    def user_reldivb(self):
        """field_loop's "UserRelDivB" history column: fixed-B0 relative div(B) (RelDivBHst)"""
        out = C.c_double(0.0)
        self._check(self.lib.apk_sim_user_reldivb(self.h, C.byref(out)))
        return out.value

    def read_acc(self, lb):
        out = np.empty((3,) + self.block_shape[1:])
        self._check(self.lib.apk_sim_read_acc(self.h, lb, out.ctypes.data_as(L.c_dp)))
        return out

    def cpaw_errors(self):
        rms = C.c_double(0.0)
        err = (C.c_double * 8)()
        self._check(self.lib.apk_sim_cpaw_errors(self.h, C.byref(rms), err))
        return rms.value, np.array(err[:])

    def linear_wave_errors(self):
        rms = C.c_double(0.0)
        l1, mx = (C.c_double * 5)(), (C.c_double * 5)()
        self._check(self.lib.apk_sim_linear_wave_errors(self.h, C.byref(rms), l1, mx))
        return rms.value, np.array(l1[:]), np.array(mx[:])

    def linear_wave_mhd_errors(self):
        """(RMS, L1[8], max[8]) of d, M1, M2, M3, E, B1, B2, B3 against the analytic MHD wave (linear_wave_mhd.cpp:177-276)"""
        rms = C.c_double(0.0)
        l1, mx = (C.c_double * 8)(), (C.c_double * 8)()
        self._check(self.lib.apk_sim_linear_wave_mhd_errors(self.h, C.byref(rms), l1, mx))
        return rms.value, np.array(l1[:]), np.array(mx[:])

    def kernel_timing(self, on):
        self._check(self.lib.apk_sim_kernel_timing_enable(self.h, int(on)))

    def read_kernel_timing(self):
        """{slot name: (total_ms, launches)} accumulated since the last read (HIP events)."""
        out = {}
        for slot, name in enumerate(L.TIMING_SLOTS):
            ms, n = C.c_double(0.0), C.c_longlong(0)
            self._check(self.lib.apk_sim_kernel_timing_read(self.h, slot, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def exchange_ghosts(self):
        self._check(self.lib.apk_sim_exchange_ghosts(self.h))

    def fill_derived(self):
        self._check(self.lib.apk_sim_fill_derived(self.h))

    def estimate_timestep(self):
        dt = C.c_double(0.0)
        self._check(self.lib.apk_sim_estimate_timestep(self.h, C.byref(dt)))
        return dt.value

    def reset_time_step(self):
        """after write_block + exchange_ghosts + fill_derived: the time step as initialize() derives it"""
        self._check(self.lib.apk_sim_reset_time_step(self.h))
        return self.dt


class HostPlan(_FmftHost, _MeshView):
    """Host-only view of a rank's mesh partition and ghost-exchange plan (no GPU needed)."""

    def __init__(self, deck, overrides=(), rank=0, nranks=1, strict=False):
        self.lib = L.load(strict)
        ov = (C.c_char_p * max(1, len(overrides)))(*[o.encode() for o in overrides])
        err = C.create_string_buffer(1024)
        h = C.c_void_p()
        rc = self.lib.apk_sim_create_host_only(deck.encode(), ov, len(overrides), rank, nranks,
                                               C.byref(h), err, len(err))
        if rc != L.APK_OK:
            raise L.ApkError(rc, err.value.decode())
        self.h = h
        self.info = L.SimInfo()
        self.lib.apk_sim_get_info(self.h, C.byref(self.info))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.apk_sim_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def apply_tags(self, tags):
        """host logic of one regridding pass (forest, distribution, plans) for per-block tags"""
        arr = (C.c_int * len(tags))(*[int(t) for t in tags])
        ch = C.c_int(0)
        rc = self.lib.apk_sim_amr_apply_tags(self.h, arr, len(tags), C.byref(ch))
        if rc != L.APK_OK:
            raise L.ApkError(rc, self.lib.apk_sim_last_error(self.h).decode())
        self.refresh_info()
        return bool(ch.value)

    @property
    def tlim(self):
        return self.lib.apk_sim_tlim(self.h)
