"""Host-side mirror of the reference's task functions for the flux-divergence path.

Names follow the reference (Hydro::CalculateFluxes, Update::UpdateWithFluxDivergence,
GLMMHD::DednerSource, ConservedToPrimitive, EstimateTimestep, FirstOrderFluxCorrect) so the
parity tests read like the reference's call sites (src/hydro/hydro_driver.cpp:499-577).
torch is plumbing only: it owns device memory (the role Parthenon plays for AthenaPK) and
the HIP stream; every operation is one call through the C-ABI of libapk_amd.so.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import lib as L


def _check(rc, ctx_lib=None, ctx=None):
    if rc != L.APK_OK:
        msg = ""
        if ctx_lib is not None and ctx:
            msg = ctx_lib.apk_last_error(ctx).decode()
        raise L.ApkError(rc, msg)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Context:
    """apk_ctx: workspace on the current HIP device.  No device -> ApkError (no fallback)."""

    def __init__(self, strict=False):
        self.lib = L.load(strict)
        self.strict = bool(strict)
        h = C.c_void_p()
        rc = self.lib.apk_create(C.byref(h))
        if rc != L.APK_OK:
            raise L.ApkError(rc, "apk_create: no usable gfx950 device; the HIP path has no CPU fallback")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.apk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def poll_flags(self):
        f = C.c_uint(0)
        _check(self.lib.apk_poll_device_flags(self.h, C.byref(f), _stream()), self.lib, self.h)
        return f.value


class MeshData:
    """A MeshBlockPack: per-block cons / prim / flux tensors [nvar][Nk][Nj][Ni] on the GPU.

    Built from host arrays (numpy) or existing CUDA tensors; owns the tensors it creates.
    """

    def __init__(self, ctx, nx, ng, nhydro, nscalars=0, dx=(1.0, 1.0, 1.0), nblocks=1,
                 cons=None, prim=None, with_flux=True, row_pitch=None):
        """row_pitch: None / "natural" = rows of Ni doubles (LayoutRight, apk_pack_desc.stride = 0); "aligned" = rows at a
        pitch that is a multiple of 16 doubles from a base that puts the first interior cell of a row on a 128-byte
        boundary; an int = that pitch.  (APK_TEST_ROW_PITCH in the environment sets the default: the parity tests run
        unchanged on either layout.)  self.cons / prim / flux[d] are VIEWS [nblocks][nvar][Nk][Nj][Ni] of the padded
        storage; the padding holds NaNs."""
        self.ctx = ctx
        self.nx, self.ng, self.nhydro, self.nscalars = tuple(nx), ng, nhydro, nscalars
        self.nvar = nhydro + nscalars
        ni = nx[0] + 2 * ng
        nj = nx[1] + 2 * ng if nx[1] > 1 else 1
        nk = nx[2] + 2 * ng if nx[2] > 1 else 1
        self.shape = (self.nvar, nk, nj, ni)
        self.ndim = 3 if nx[2] > 1 else (2 if nx[1] > 1 else 1)
        self.nblocks = nblocks
        self.dx = tuple(dx)
        dev = torch.device("cuda")
        if row_pitch is None:
            row_pitch = os.environ.get("APK_TEST_ROW_PITCH") or None
        if row_pitch in (None, "natural"):
            pitch, lead = ni, 0
        elif row_pitch == "aligned":
            pitch, lead = (ni + 15) // 16 * 16, (-ng) % 16
        else:
            pitch, lead = int(row_pitch), 0
        assert pitch >= ni
        self.pitch, self.lead = pitch, lead
        sk, sn = pitch * nj, pitch * nj * nk
        natural = pitch == ni and lead == 0
        slot = self.nvar * sn if natural else (lead + self.nvar * sn + 15) // 16 * 16
        self._storage = []

        def field(src):
            if natural:
                if src is None:
                    return torch.zeros((nblocks,) + self.shape, dtype=torch.float64, device=dev)
                if isinstance(src, torch.Tensor):
                    t = src.to(device=dev, dtype=torch.float64).contiguous()
                else:
                    t = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float64)).to(dev)
                return t.reshape((nblocks,) + self.shape).contiguous()
            store = torch.full((nblocks * slot,), float("nan"), dtype=torch.float64, device=dev)
            self._storage.append(store)
            view = store.as_strided((nblocks,) + self.shape, (slot, sn, sk, pitch, 1), lead)
            if src is None:
                view.zero_()
            elif isinstance(src, torch.Tensor):
                view.copy_(src.to(device=dev, dtype=torch.float64).reshape((nblocks,) + self.shape))
            else:
                view.copy_(torch.from_numpy(np.ascontiguousarray(src, dtype=np.float64)).to(dev).reshape((nblocks,) + self.shape))
            return view

        self.cons = field(cons)
        self.prim = field(prim)
        self.flux = [field(None) if (with_flux and d < self.ndim) else None for d in range(3)]
        blocks = (L.BlockDesc * nblocks)()
        per = slot * 8
        for b in range(nblocks):
            blocks[b].cons = self.cons.data_ptr() + b * per
            blocks[b].prim = self.prim.data_ptr() + b * per
            for d in range(3):
                blocks[b].flux[d] = (self.flux[d].data_ptr() + b * per) if self.flux[d] is not None else None
            blocks[b].dx[:] = list(dx)
        desc = L.PackDesc()
        desc.nblocks, desc.nhydro, desc.nscalars, desc.ng = nblocks, nhydro, nscalars, ng
        desc.nx[:] = list(nx)
        desc.blocks = blocks
        if not natural:
            desc.stride[:] = [pitch, sk, sn]
        h = C.c_void_p()
        _check(ctx.lib.apk_pack_create(ctx.h, C.byref(desc), C.byref(h)), ctx.lib, ctx.h)
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.apk_pack_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def cons_host(self):
        torch.cuda.synchronize()
        return self.cons.cpu().numpy()

    def prim_host(self):
        torch.cuda.synchronize()
        return self.prim.cpu().numpy()

    def flux_host(self, d):
        torch.cuda.synchronize()
        return self.flux[d].cpu().numpy()


def _cfg(fluid, recon, riemann):
    return L.FluxCfg(L.FLUID[fluid], L.RECON[recon], L.RIEMANN[riemann])


class CopyPlan:
    """apk_copy_plan: a list of strided box copies (ghost-zone fills, message packing / unpacking, physical boundaries)
    executed in one launch.  regions: dicts with src / dst (device addresses), ext (ni, nj, nk), nvar, src_stride /
    dst_stride (elements, for i, j, k, variable) and optionally flip_var."""

    def __init__(self, ctx, regions):
        self.ctx = ctx
        regs = (L.CopyRegion * len(regions))()
        for r, q in zip(regs, regions):
            r.src, r.dst, r.nvar = q["src"], q["dst"], q["nvar"]
            r.ext[:] = list(q["ext"])
            r.src_stride[:] = list(q["src_stride"])
            r.dst_stride[:] = list(q["dst_stride"])
            r.flip_var = q.get("flip_var", -1)
        h = C.c_void_p()
        _check(ctx.lib.apk_copy_plan_create(ctx.h, regs, len(regions), C.byref(h)), ctx.lib, ctx.h)
        self.h = h

    def run(self):
        _check(self.ctx.lib.apk_copy_plan_run(self.ctx.h, self.h, _stream()), self.ctx.lib, self.ctx.h)

    def run_c2p(self, fluid, eos, prim_delta, latch_flags=True, prim_only=False):
        """the copy with ConservedToPrimitive of every destination cell (primitives at dst + prim_delta elements);
        prim_only: nothing is stored at dst itself"""
        fn = self.ctx.lib.apk_copy_plan_run_c2p_prim_only if prim_only else self.ctx.lib.apk_copy_plan_run_c2p
        _check(fn(self.ctx.h, self.h, L.FLUID[fluid], C.byref(eos), int(prim_delta), 1 if latch_flags else 0, _stream()),
               self.ctx.lib, self.ctx.h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.apk_copy_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass


# ---- the task functions ---------------------------------------------------------------------
def CalculateFluxes(md, fluid, recon, riemann, eos, c_h=0.0, tight=False, boundary=False, face_list=None, from_cons=None):
    """Hydro::CalculateFluxes<fluid,recon,rsolver>(md)  -- src/hydro/hydro.cpp:1025; tight: only the
    faces of interior cells (the loop limits of CalculateFluxesTight, hydro.cpp:1006-1009); boundary:
    only the 2 ndim block-boundary planes (for the flux correction after a fused stage), with face_list
    (int32 CUDA tensor of 6 * block + face codes) only the listed planes, in one launch; from_cons (with
    face_list): a MeshData of the same shape whose CONSERVED state is the input, converted in registers
    (apk_calculate_fluxes_boundary_list_from_cons) -- md itself for its own state"""
    ctx = md.ctx
    if from_cons is not None:
        assert boundary and face_list is not None and face_list.dtype == torch.int32 and face_list.is_cuda and face_list.dim() == 1
        delta = (from_cons.cons.data_ptr() - md.cons.data_ptr()) // 8
        _check(ctx.lib.apk_calculate_fluxes_boundary_list_from_cons(ctx.h, md.h, _cfg(fluid, recon, riemann), C.byref(eos), float(c_h),
                                                                    face_list.data_ptr(), int(face_list.numel()), int(delta), _stream()),
               ctx.lib, ctx.h)
        return
    if face_list is not None:
        assert boundary and face_list.dtype == torch.int32 and face_list.is_cuda and face_list.dim() == 1
        _check(ctx.lib.apk_calculate_fluxes_boundary_list(ctx.h, md.h, _cfg(fluid, recon, riemann), C.byref(eos), float(c_h),
                                                          face_list.data_ptr(), int(face_list.numel()), _stream()), ctx.lib, ctx.h)
        return
    fn = (ctx.lib.apk_calculate_fluxes_boundary if boundary else
          (ctx.lib.apk_calculate_fluxes_tight if tight else ctx.lib.apk_calculate_fluxes))
    _check(fn(ctx.h, md.h, _cfg(fluid, recon, riemann), C.byref(eos), float(c_h), _stream()), ctx.lib, ctx.h)


def UpdateWithFluxDivergence(u0, u1, gam0, gam1, beta_dt):
    """parthenon::Update::UpdateWithFluxDivergence -- call site hydro_driver.cpp:534"""
    ctx = u0.ctx
    _check(ctx.lib.apk_update_with_flux_divergence(ctx.h, u0.h, u1.h, gam0, gam1, beta_dt,
                                                   _stream()), ctx.lib, ctx.h)


def DednerSource(md, extended, alpha, c_h, mindx, beta_dt):
    """GLMMHD::DednerSource<extended>(md, beta_dt) -- src/hydro/glmmhd/dedner_source.cpp:17"""
    ctx = md.ctx
    _check(ctx.lib.apk_dedner_source(ctx.h, md.h, int(extended), alpha, c_h, mindx, beta_dt,
                                     _stream()), ctx.lib, ctx.h)


def StageFused(u0, u1, fluid, recon, riemann, eos, c_h, gam0, gam1, beta_dt, dedner=0,
               glmmhd_alpha=0.1, mindx=1.0, fill_derived=False, estimate_dt=False, phase=0, window=None,
               face_neighbor=None, cons_store=0, prim_from_cons=False, cons_out=None, x1_halo=None):
    """Fused CalculateFluxes -> UpdateWithFluxDivergence -> DednerSource for one RK stage;
    optionally also FillDerived / the dt estimate on the updated cells.  phase / x1_window split
    the stage around a halo exchange (window: int32 CUDA tensor [nblocks][8] =
    i0, rl, ilo, ihi, jlo, jhi, klo, khi)."""
    ctx = u0.ctx
    a = L.StageArgs()
    a.cfg = _cfg(fluid, recon, riemann)
    a.eos = eos
    a.c_h, a.gam0, a.gam1, a.beta_dt = c_h, gam0, gam1, beta_dt
    a.dedner, a.glmmhd_alpha, a.mindx = dedner, glmmhd_alpha, mindx
    a.fill_derived, a.estimate_dt = int(fill_derived), int(estimate_dt)
    a.phase = phase
    if window is not None:
        assert window.dtype == torch.int32 and window.is_cuda and window.shape == (u0.nblocks, 8)
        a.window = window.data_ptr()
        a.window_rl = int(window[:, 1].max().item())
        a.window_rows = max(1, int((window[:, 5] - window[:, 4] + 1).max().item()))
    if face_neighbor is not None:  # direct neighbour addressing: int32 CUDA tensor [nblocks][6]
        assert face_neighbor.dtype == torch.int32 and face_neighbor.is_cuda and face_neighbor.shape == (u0.nblocks, 6)
        a.face_neighbor = face_neighbor.data_ptr()
    a.prim_from_cons = int(prim_from_cons)  # the stage derives its input primitives from u1.cons (apk_stage_args.prim_from_cons)
    if cons_out is not None:  # the updated conserved state goes to this MeshData of identical layout (apk_stage_args.cons_out_delta)
        assert cons_out.cons.shape == u0.cons.shape
        a.cons_out_delta = (cons_out.cons.data_ptr() - u0.cons.data_ptr()) // 8
    a.cons_store = int(cons_store)  # 0 all cells, 1 the nghost-deep shell of every block, 2 none (apk_stage_args.cons_store)
    keep = None
    if x1_halo is not None:
        # apk_stage_args.x1_halo: dict(recv=[per block (lo, hi) float64 CUDA tensors or None], send=[...], recv_depth=,
        # send_depth=, send_field=0 cons / 1 prim) -- x1 strips read from / stored into exchange-buffer segments
        # [nvar][nx3][nx2][depth]
        nb = u0.nblocks
        tab = (L.X1HaloBlock * nb)()
        for b in range(nb):
            for side in range(2):
                r = x1_halo.get("recv", [(None, None)] * nb)[b][side]
                w = x1_halo.get("send", [(None, None)] * nb)[b][side]
                for t in (r, w):
                    assert t is None or (t.dtype == torch.float64 and t.is_cuda and t.is_contiguous())
                tab[b].recv[side] = r.data_ptr() if r is not None else None
                tab[b].send[side] = w.data_ptr() if w is not None else None
        dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda()
        h = L.X1Halo()
        h.blocks = dev.data_ptr()
        h.recv_depth, h.send_depth = int(x1_halo.get("recv_depth", 0)), int(x1_halo.get("send_depth", 0))
        h.send_field = int(x1_halo.get("send_field", 0))
        a.x1_halo = C.addressof(h)
        keep = (tab, dev, h)
    _check(ctx.lib.apk_stage_fused(ctx.h, u0.h, u1.h, C.byref(a), _stream()), ctx.lib, ctx.h)
    if keep is not None:
        torch.cuda.current_stream().synchronize()  # (the table is a temporary of this call)


def StageFollowsX1Halo(u0, fluid, recon, riemann, eos, fill_derived=2, dedner=0, prim_from_cons=0):
    """apk_stage_x1_halo: does a whole-block stage of this scheme follow apk_stage_args.x1_halo?"""
    return bool(u0.ctx.lib.apk_stage_x1_halo(u0.h, C.byref(_cfg(fluid, recon, riemann)), C.byref(eos), int(fill_derived), int(dedner),
                                             int(prim_from_cons)))


def ConservedToPrimitive(md, fluid, eos):
    """EquationOfState::ConservedToPrimitive(md) -- src/eos/adiabatic_hydro.cpp:33"""
    ctx = md.ctx
    _check(ctx.lib.apk_cons_to_prim(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), _stream()), ctx.lib, ctx.h)


def ConservedToPrimitiveDt(md, fluid, eos, cfl, ghost_depth=-1, face_neighbor=None, store_vars=None):
    """ConsToPrim of every cell (ghost_depth >= 0: of the cells at most that many layers outside the interior) and the
    hyperbolic time-step estimate of the interior in one pass (apk_cons_to_prim_dt); face_neighbor (int32 device tensor
    [nblocks, 6]): not the ghost cells straight behind the faces whose entry is >= 0 (apk_cons_to_prim_dt_skip);
    store_vars (a bit mask of primitives, with ghost_depth >= 0): only those are stored (apk_cons_to_prim_dt_select)."""
    ctx = md.ctx
    if store_vars is not None:
        fn = C.c_void_p(face_neighbor.data_ptr()) if face_neighbor is not None else None
        _check(ctx.lib.apk_cons_to_prim_dt_select(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), ghost_depth, fn, int(store_vars), _stream()),
               ctx.lib, ctx.h)
        return StageDt(ctx, cfl)
    if face_neighbor is None:
        _check(ctx.lib.apk_cons_to_prim_dt(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), ghost_depth, _stream()), ctx.lib, ctx.h)
    else:
        _check(ctx.lib.apk_cons_to_prim_dt_skip(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), ghost_depth,
                                                C.c_void_p(face_neighbor.data_ptr()), _stream()), ctx.lib, ctx.h)
    return StageDt(ctx, cfl)


def ConservedToPrimitiveFacesDt(md, fluid, eos, cfl, face_neighbor=None):
    """ConservedToPrimitiveFaces with the hyperbolic time-step estimate of the interior (apk_cons_to_prim_faces_dt)."""
    ctx = md.ctx
    fn = C.c_void_p(face_neighbor.data_ptr()) if face_neighbor is not None else None
    _check(ctx.lib.apk_cons_to_prim_faces_dt(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), fn, _stream()), ctx.lib, ctx.h)
    return StageDt(ctx, cfl)


def ConservedToPrimitiveFaces(md, fluid, eos, face_neighbor=None):
    """ConsToPrim of the interior and of the ghost cells straight behind a block face (at most one ghost coordinate);
    face_neighbor (int32 device tensor [nblocks, 6]): not behind the faces whose entry is >= 0."""
    ctx = md.ctx
    if face_neighbor is None:
        _check(ctx.lib.apk_cons_to_prim_faces(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), _stream()), ctx.lib, ctx.h)
    else:
        _check(ctx.lib.apk_cons_to_prim_faces_skip(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), C.c_void_p(face_neighbor.data_ptr()),
                                                   _stream()), ctx.lib, ctx.h)


def ConservedToPrimitiveGhosts(md, fluid, eos):
    """ConsToPrim on the ghost zones only (companion of StageFused(fill_derived=True))."""
    ctx = md.ctx
    _check(ctx.lib.apk_cons_to_prim_ghosts(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), _stream()), ctx.lib, ctx.h)


def StageDt(ctx, cfl):
    """cfl * min dx/(|v|+c) reduced by the last StageFused(estimate_dt=True)."""
    dt = C.c_double(0.0)
    _check(ctx.lib.apk_stage_dt_read(ctx.h, cfl, C.byref(dt), _stream()), ctx.lib, ctx.h)
    return dt.value


def EstimateTimestep(md, fluid, eos, cfl):
    """Hydro::EstimateHyperbolicTimestep<fluid>(md) -- src/hydro/hydro.cpp:828"""
    ctx = md.ctx
    dt = C.c_double(0.0)
    _check(ctx.lib.apk_estimate_timestep(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), cfl, C.byref(dt),
                                         _stream()), ctx.lib, ctx.h)
    return dt.value


def FirstOrderFluxCorrect(u0, u1, fluid, eos, c_h, gam0, gam1, beta_dt):
    """Hydro::FirstOrderFluxCorrect<fluid>(u0,u1,gam0,gam1,beta_dt) -- hydro.cpp:1223"""
    ctx = u0.ctx
    n = C.c_longlong(0)
    _check(ctx.lib.apk_first_order_flux_correct(ctx.h, u0.h, u1.h, L.FLUID[fluid], C.byref(eos), c_h,
                                                gam0, gam1, beta_dt, C.byref(n), _stream()), ctx.lib, ctx.h)
    return n.value


def CountUnphysical(md, fluid):
    """cells whose conserved state fails FirstOrderFluxCorrect's test (hydro.cpp:1297-1306)"""
    ctx = md.ctx
    n = C.c_longlong(0)
    _check(ctx.lib.apk_count_unphysical(ctx.h, md.h, L.FLUID[fluid], C.byref(n), _stream()), ctx.lib, ctx.h)
    return n.value


class FluxFixPlan:
    """Coarse-fine flux correction applied to the cells next to the faces after a fused stage:
    regions = list of (fine_avg tensor view, coarse_flux tensor view, cons tensor view, scale) where
    the three views have the same shape [nvar][nk][nj][ni] (strides are taken from the views)."""

    def __init__(self, ctx, regions, merged=None):
        """merged = (n_by_dir, field tensor [nblocks][nvar][nk][nj][ni]): the regions of all directions (direction 0's
        first) as ONE plan that runs in one launch (apk_flux_fix_plan_create_merged)."""
        self.ctx = ctx
        arr = (L.FluxFixRegion * max(1, len(regions)))()
        self._keep = []
        for n, reg in enumerate(regions):
            fa, cf, cons, scale = reg[:4]
            assert fa.shape == cf.shape == cons.shape and cf.stride() == cons.stride()
            if len(reg) > 4:  # (direction 1..3, ndim, fine_area, fine array): fa = its every-second-face view; average in the kernel
                arr[n].average, arr[n].ndim, arr[n].fine_area = int(reg[4]), int(reg[5]), float(reg[6])
                arr[n].fine_stride[:] = [reg[7].stride(3), reg[7].stride(2), reg[7].stride(1)]
                self._keep.append(reg[7])
            arr[n].fine_avg, arr[n].coarse_flux, arr[n].cons = fa.data_ptr(), cf.data_ptr(), cons.data_ptr()
            arr[n].nvar = fa.shape[0]
            arr[n].ext[:] = [fa.shape[3], fa.shape[2], fa.shape[1]]
            arr[n].src_stride[:] = [fa.stride(3), fa.stride(2), fa.stride(1), fa.stride(0)]
            arr[n].dst_stride[:] = [cons.stride(3), cons.stride(2), cons.stride(1), cons.stride(0)]
            arr[n].scale = scale
            self._keep += [fa, cf, cons]
        h = C.c_void_p()
        if merged is None:
            _check(ctx.lib.apk_flux_fix_plan_create(ctx.h, arr, len(regions), C.byref(h)), ctx.lib, ctx.h)
        else:
            n_by_dir, field = merged
            assert sum(n_by_dir) == len(regions) and field.is_contiguous()
            nd = (C.c_int * 3)(*[int(x) for x in n_by_dir])
            _check(ctx.lib.apk_flux_fix_plan_create_merged(ctx.h, arr, nd, field.data_ptr(), int(field[0].numel()), C.byref(h)),
                   ctx.lib, ctx.h)
            self._keep.append(field)
        self.h = h

    def run(self, beta_dt, psi_var=-1, psi_factor=1.0):
        _check(self.ctx.lib.apk_flux_fix_plan_run(self.ctx.h, self.h, float(beta_dt), int(psi_var), float(psi_factor), _stream()),
               self.ctx.lib, self.ctx.h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.apk_flux_fix_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass


def HydroHst(md, fluid):
    """HydroHst<...> reductions -- src/hydro/hydro.cpp:145-208"""
    ctx = md.ctx
    out = (C.c_double * 8)()
    _check(ctx.lib.apk_history(ctx.h, md.h, L.FLUID[fluid], out, _stream()), ctx.lib, ctx.h)
    return np.array(out[:])


# ---- few-modes turbulence driver -------------------------------------------------------------
class FewModesFT:
    """Device side of FewModesFT / turbulence::Perturb for one MeshData: the acceleration field
    "acc" and the per-block phase tables (src/utils/few_modes_ft.cpp:142-195,
    src/pgen/turbulence.cpp:119-127).  `phases[b]` is (phases_i, phases_j, phases_k), each
    [2][num_modes][n] (re|im, mode, cell)."""

    def __init__(self, md, phases):
        self.md = md
        ctx = md.ctx
        dev = torch.device("cuda")
        self.num_modes = int(np.asarray(phases[0][0]).shape[1])
        assert np.asarray(phases[0][0]).shape[0] == 2
        self.acc = torch.zeros((md.nblocks, 3) + md.shape[1:], dtype=torch.float64, device=dev)
        self._ph = [[torch.from_numpy(np.ascontiguousarray(p, dtype=np.float64)).to(dev) for p in blk]
                    for blk in phases]
        blocks = (L.FmftBlock * md.nblocks)()
        per = 3 * int(np.prod(md.shape[1:])) * 8
        for b in range(md.nblocks):
            blocks[b].acc = self.acc.data_ptr() + b * per
            blocks[b].phases_i, blocks[b].phases_j, blocks[b].phases_k = [t.data_ptr() for t in self._ph[b]]
        h = C.c_void_p()
        _check(ctx.lib.apk_fmft_create(ctx.h, blocks, md.nblocks, self.num_modes, C.byref(h)), ctx.lib, ctx.h)
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.md.ctx.lib.apk_fmft_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def Inverse(self, var_hat):
        """FewModesFT::Generate's inverse transform -- src/utils/few_modes_ft.cpp:322-347"""
        ctx = self.md.ctx
        vh = np.ascontiguousarray(var_hat, dtype=np.float64)
        assert vh.shape == (3, self.num_modes, 2)
        _check(ctx.lib.apk_fmft_inverse(ctx.h, self.md.h, self.h, vh.ctypes.data_as(L.c_dp), _stream()), ctx.lib, ctx.h)
        torch.cuda.current_stream().synchronize()  # vh must outlive the copy

    def Perturb(self, dt, accel_rms, box_volume, allreduce_sum=None, fill=None):
        """turbulence::Perturb -- src/pgen/turbulence.cpp:384-470.  `allreduce_sum(np.ndarray)`
        stands where the reference calls MPI_Allreduce.  fill = (fluid, eos, estimate_dt): the kick also does
        FillDerived (prim in place) and the time-step estimate of its cells (apk_turb_apply_fill)."""
        ctx = self.md.ctx
        sums = np.zeros(4)
        _check(ctx.lib.apk_turb_mean_momentum(ctx.h, self.md.h, self.h, sums.ctypes.data_as(L.c_dp), _stream()),
               ctx.lib, ctx.h)
        if allreduce_sum is not None:
            allreduce_sum(sums)
        ampl = np.zeros(1)
        _check(ctx.lib.apk_turb_remove_mean(ctx.h, self.md.h, self.h, sums.ctypes.data_as(L.c_dp),
                                            ampl.ctypes.data_as(L.c_dp), _stream()), ctx.lib, ctx.h)
        if allreduce_sum is not None:
            allreduce_sum(ampl)
        norm = accel_rms / np.sqrt(ampl[0] / box_volume)
        if fill is not None and len(fill) == 4 and not fill[3]:  # (fluid, eos, True, store_prim = False): apk_turb_apply_dt
            fluid, eos = fill[0], fill[1]
            _check(ctx.lib.apk_turb_apply_dt(ctx.h, self.md.h, self.h, float(norm), float(dt), L.FLUID[fluid], C.byref(eos), _stream()),
                   ctx.lib, ctx.h)
        elif fill is not None:
            fluid, eos, estimate_dt = fill[:3]
            _check(ctx.lib.apk_turb_apply_fill(ctx.h, self.md.h, self.h, float(norm), float(dt), L.FLUID[fluid], C.byref(eos),
                                               int(estimate_dt), _stream()), ctx.lib, ctx.h)
        else:
            _check(ctx.lib.apk_turb_apply(ctx.h, self.md.h, self.h, float(norm), float(dt), _stream()), ctx.lib, ctx.h)
        return norm

    def acc_host(self):
        torch.cuda.synchronize()
        return self.acc.cpu().numpy()


def TurbulenceHst(md, fluid, gamma):
    """TurbulenceHst<Ms|Ma|pb> -- src/pgen/turbulence.cpp:47-101"""
    ctx = md.ctx
    out = (C.c_double * 3)()
    _check(ctx.lib.apk_turbulence_history(ctx.h, md.h, L.FLUID[fluid], float(gamma), out, _stream()), ctx.lib, ctx.h)
    return np.array(out[:])


# ---- mesh-refinement operators and tagging -----------------------------------------------------
class RefinePlan:
    """A batch of prolongation / restriction index boxes over meshblocks of one shape, run in one
    launch.  ops: list of (kind, src_tensor, dst_tensor, lo, hi, xmin) with kind in
    lib.REFINE_OPS; tensors are CUDA float64 (fine arrays [nvar][Nk][Nj][Ni], coarse buffers
    [nvar][cNk][cNj][cNi], face arrays one longer along their direction).
    ProlongateCellMinModMultiD -- src/hydro/prolongation/custom_ops.hpp:49-186; RestrictAverage
    (Parthenon), registered at src/hydro/hydro.cpp:780-781."""

    def __init__(self, ctx, nx, ng, cng, dx, nvar, ops):
        self.ctx = ctx
        g = L.RefineGeom()
        g.nx[:] = list(nx)
        g.ng, g.cng = ng, cng
        g.dx[:] = list(dx)
        arr = (L.RefineOp * len(ops))()
        self._keep = []
        for n, (kind, src, dst, lo, hi, xmin) in enumerate(ops):
            arr[n].kind = L.REFINE_OPS[kind]
            arr[n].src, arr[n].dst = src.data_ptr(), dst.data_ptr()
            arr[n].lo[:], arr[n].hi[:], arr[n].xmin[:] = list(lo), list(hi), list(xmin)
            self._keep += [src, dst]
        h = C.c_void_p()
        _check(ctx.lib.apk_refine_plan_create(ctx.h, C.byref(g), nvar, arr, len(ops), C.byref(h)), ctx.lib, ctx.h)
        self.h = h

    def run(self):
        _check(self.ctx.lib.apk_refine_plan_run(self.ctx.h, self.h, _stream()), self.ctx.lib, self.ctx.h)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.apk_refine_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass


def TagBlocksDtFromCons(md, fluid, eos, cfl, p0, face_neighbor=None):
    """The pressure-gradient criterion and the hyperbolic time-step estimate in one pass over the conserved state
    (apk_tag_blocks_dt_from_cons + apk_tag_blocks_end).  Returns (tags, criterion values, dt)."""
    ctx = md.ctx
    tags = (C.c_int * md.nblocks)()
    crit = (C.c_double * md.nblocks)()
    pending = C.c_int(0)
    fn = C.c_void_p(face_neighbor.data_ptr()) if face_neighbor is not None else None
    _check(ctx.lib.apk_tag_blocks_dt_from_cons(ctx.h, md.h, L.FLUID[fluid], C.byref(eos), fn, C.byref(pending), _stream()), ctx.lib, ctx.h)
    dt = StageDt(ctx, cfl)
    code = L.TAG_CRITERIA["pressure_gradient"]
    _check(ctx.lib.apk_tag_blocks_end(ctx.h, md.nblocks, code, pending.value, float(p0), 0.0, tags, crit, _stream()), ctx.lib, ctx.h)
    return np.array(tags[:]), np.array(crit[:]), dt


def TagBlocks(md, criterion, p0, p1=0.0, face_neighbor=None):
    """refinement::gradient::PressureGradient / VelocityGradient (src/refinement/gradient.cpp:18-99),
    refinement::other::MaxDensity (src/refinement/other.cpp:18-44) for every block of the pack.
    Returns (tags, criterion values); tags: +1 refine, 0 same, -1 derefine.  face_neighbor (int32 device tensor
    [nblocks, 6]): the ghost cells straight behind the faces whose entry is >= 0 are read from that block's interior
    (apk_tag_blocks_begin_skip + apk_tag_blocks_end)."""
    ctx = md.ctx
    tags = (C.c_int * md.nblocks)()
    crit = (C.c_double * md.nblocks)()
    if face_neighbor is None:
        _check(ctx.lib.apk_tag_blocks(ctx.h, md.h, L.TAG_CRITERIA[criterion], float(p0), float(p1), tags, crit, _stream()),
               ctx.lib, ctx.h)
    else:
        pending = C.c_int(0)
        code = L.TAG_CRITERIA[criterion]
        _check(ctx.lib.apk_tag_blocks_begin_skip(ctx.h, md.h, code, C.c_void_p(face_neighbor.data_ptr()), C.byref(pending), _stream()),
               ctx.lib, ctx.h)
        _check(ctx.lib.apk_tag_blocks_end(ctx.h, md.nblocks, code, pending.value, float(p0), float(p1), tags, crit, _stream()),
               ctx.lib, ctx.h)
    return np.array(tags[:]), np.array(crit[:])
