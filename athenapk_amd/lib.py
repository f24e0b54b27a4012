"""ctypes binding of libapk_amd.so (include/apk_amd.h + include/apk_host.h).

The shared library is the product; this module only loads it and declares the C
signatures.  There is no Python/CPU fallback: if the library is missing, or no gfx950
device is visible when a context is created, the call fails loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")

# enums of include/apk_amd.h (positions in the reference's enum classes, src/main.hpp:35-38)
FLUID = {"euler": 1, "glmmhd": 2}
RECON = {"dc": 1, "plm": 2, "ppm": 3, "wenoz": 4, "weno3": 5, "limo3": 6}
RIEMANN = {"none": 1, "hlle": 2, "llf": 3, "hllc": 4, "hlld": 5}
INTEGRATOR = {"rk1": 1, "rk2": 2, "vl2": 3, "rk3": 4}

TIMING_SLOTS = ("fused_x1", "fused_x2", "fused_x3", "fluxes", "update", "dedner", "cons_to_prim",
                "min_dt", "copy_regions", "fused_dc_x1", "fused_dc_x2", "fused_dc_x3")

APK_OK = 0
APK_RCCL_ID_BYTES = 128
APK_ERR_INVALID, APK_ERR_UNSUPPORTED, APK_ERR_NGHOST, APK_ERR_DEVICE, APK_ERR_NO_DEVICE = -1, -2, -3, -4, -5
FLAG_NEG_DENSITY, FLAG_NEG_PRESSURE = 1, 2

c_dp = C.POINTER(C.c_double)


class Eos(C.Structure):
    _fields_ = [("gamma", C.c_double), ("pfloor", C.c_double), ("dfloor", C.c_double),
                ("efloor", C.c_double), ("vceil", C.c_double), ("eceil", C.c_double)]


def make_eos(gamma, pfloor=-1.0, dfloor=-1.0, efloor=-1.0, vceil=float("inf"),
             eceil=float("inf")):
    return Eos(gamma, pfloor, dfloor, efloor, vceil, eceil)


class FluxCfg(C.Structure):
    _fields_ = [("fluid", C.c_int), ("recon", C.c_int), ("riemann", C.c_int)]


class BlockDesc(C.Structure):
    _fields_ = [("cons", C.c_void_p), ("prim", C.c_void_p), ("flux", C.c_void_p * 3),
                ("dx", C.c_double * 3)]


class PackDesc(C.Structure):
    _fields_ = [("nblocks", C.c_int), ("nhydro", C.c_int), ("nscalars", C.c_int),
                ("nx", C.c_int * 3), ("ng", C.c_int), ("blocks", C.POINTER(BlockDesc)), ("stride", C.c_int64 * 3)]


class StageArgs(C.Structure):
    _fields_ = [("cfg", FluxCfg), ("eos", Eos), ("c_h", C.c_double), ("gam0", C.c_double),
                ("gam1", C.c_double), ("beta_dt", C.c_double), ("dedner", C.c_int),
                ("glmmhd_alpha", C.c_double), ("mindx", C.c_double), ("fill_derived", C.c_int),
                ("estimate_dt", C.c_int), ("phase", C.c_int), ("window", C.c_void_p),
                ("window_rl", C.c_int), ("window_rows", C.c_int), ("trial", C.c_int), ("count_unphysical", C.c_int), ("cons_out_delta", C.c_int64),
                ("face_neighbor", C.c_void_p), ("cons_store", C.c_int), ("prim_from_cons", C.c_int), ("x1_halo", C.c_void_p)]


class X1HaloBlock(C.Structure):
    _fields_ = [("recv", C.c_void_p * 2), ("send", C.c_void_p * 2)]


class X1Halo(C.Structure):
    _fields_ = [("blocks", C.c_void_p), ("recv_depth", C.c_int), ("send_depth", C.c_int), ("send_field", C.c_int)]


class FmftBlock(C.Structure):
    _fields_ = [("acc", C.c_void_p), ("phases_i", C.c_void_p), ("phases_j", C.c_void_p),
                ("phases_k", C.c_void_p)]


class RefineGeom(C.Structure):
    _fields_ = [("nx", C.c_int * 3), ("ng", C.c_int), ("cng", C.c_int), ("dx", C.c_double * 3)]


class RefineOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("src", C.c_void_p), ("dst", C.c_void_p), ("lo", C.c_int * 3),
                ("hi", C.c_int * 3), ("xmin", C.c_double * 3), ("dx", C.c_double * 3)]


REFINE_OPS = {"prolongate": 0, "restrict_cell": 1, "restrict_face1": 2, "restrict_face2": 3, "restrict_face3": 4,
              "restrict_flux1": 5, "restrict_flux2": 6, "restrict_flux3": 7}
TAG_CRITERIA = {"pressure_gradient": 0, "xyvelocity_gradient": 1, "maxdensity": 2}


class FluxFixRegion(C.Structure):
    _fields_ = [("fine_avg", C.c_void_p), ("coarse_flux", C.c_void_p), ("cons", C.c_void_p), ("ext", C.c_int * 3),
                ("nvar", C.c_int), ("src_stride", C.c_int64 * 4), ("dst_stride", C.c_int64 * 4), ("scale", C.c_double),
                ("average", C.c_int), ("ndim", C.c_int), ("fine_stride", C.c_int64 * 3), ("fine_area", C.c_double)]


class CopyRegion(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("ext", C.c_int * 3),
                ("nvar", C.c_int), ("src_stride", C.c_int64 * 4),
                ("dst_stride", C.c_int64 * 4), ("flip_var", C.c_int)]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t)
RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_dp, C.c_int)


class Allocator(C.Structure):
    _fields_ = [("user", C.c_void_p), ("alloc", ALLOC_FN), ("release", RELEASE_FN)]


class CommOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("exchange", EXCHANGE_FN), ("allreduce_min", ALLREDUCE_FN),
                ("allreduce_sum", ALLREDUCE_FN), ("exchange_begin", EXCHANGE_FN), ("exchange_end", EXCHANGE_FN)]


class SimInfo(C.Structure):
    _fields_ = [("fluid", C.c_int), ("recon", C.c_int), ("riemann", C.c_int), ("integrator", C.c_int),
                ("nx", C.c_int * 3), ("mb", C.c_int * 3), ("ng", C.c_int), ("nhydro", C.c_int),
                ("nscalars", C.c_int), ("ndim", C.c_int), ("nblocks_total", C.c_int),
                ("nblocks_local", C.c_int), ("first_gid", C.c_int), ("rank", C.c_int),
                ("nranks", C.c_int), ("npeers", C.c_int), ("fofc", C.c_int),
                ("dedner_extended", C.c_int), ("fused", C.c_int), ("cfl", C.c_double),
                ("gamma", C.c_double), ("glmmhd_alpha", C.c_double), ("xmin", C.c_double * 3),
                ("xmax", C.c_double * 3), ("dx", C.c_double * 3), ("cells_per_block", C.c_int64),
                ("zones_local", C.c_int64), ("zones_total", C.c_int64)]


class PeerInfo(C.Structure):
    _fields_ = [("rank", C.c_int), ("send_count", C.c_int64), ("recv_count", C.c_int64),
                ("send_buf", C.c_void_p), ("recv_buf", C.c_void_p)]


class RegionInfo(C.Structure):
    _fields_ = [("src_kind", C.c_int), ("src_block", C.c_int), ("dst_kind", C.c_int),
                ("dst_block", C.c_int), ("src_off", C.c_int64), ("dst_off", C.c_int64),
                ("ext", C.c_int * 3), ("nvar", C.c_int), ("flip_var", C.c_int),
                ("src_stride", C.c_int64 * 4), ("dst_stride", C.c_int64 * 4)]


class AmrOpInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("level", C.c_int), ("src_kind", C.c_int), ("src_block", C.c_int),
                ("dst_kind", C.c_int), ("dst_block", C.c_int), ("lo", C.c_int * 3), ("hi", C.c_int * 3),
                ("xmin", C.c_double * 3), ("dx", C.c_double * 3), ("cng", C.c_int), ("coarse_doubles", C.c_int64)]


# every symbol include/apk_amd.h and include/apk_host.h declare: name -> (restype, argtypes)
def _signatures():
    i, d, vp, ll = C.c_int, C.c_double, C.c_void_p, C.c_longlong
    E = C.POINTER(Eos)
    pp = C.POINTER(C.c_void_p)
    strs = C.POINTER(C.c_char_p)
    return {
        # apk_amd.h
        "apk_version": (i, []),
        "apk_fp_strict": (i, []),
        "apk_create": (i, [pp]),
        "apk_destroy": (None, [vp]),
        "apk_last_error": (C.c_char_p, [vp]),
        "apk_pack_create": (i, [vp, C.POINTER(PackDesc), pp]),
        "apk_pack_destroy": (None, [vp]),
        "apk_calculate_fluxes": (i, [vp, vp, FluxCfg, E, d, vp]),
        "apk_calculate_fluxes_tight": (i, [vp, vp, FluxCfg, E, d, vp]),
        "apk_calculate_fluxes_boundary": (i, [vp, vp, FluxCfg, E, d, vp]),
        "apk_calculate_fluxes_boundary_list": (i, [vp, vp, FluxCfg, E, d, vp, C.c_int, vp]),
        "apk_calculate_fluxes_boundary_list_from_cons": (i, [vp, vp, FluxCfg, E, d, vp, C.c_int, C.c_longlong, vp]),
        "apk_flux_fix_plan_create": (i, [vp, C.POINTER(FluxFixRegion), i, pp]),
        "apk_flux_fix_plan_create_merged": (i, [vp, C.POINTER(FluxFixRegion), C.POINTER(C.c_int), vp, C.c_int64, pp]),
        "apk_flux_fix_plan_destroy": (None, [vp]),
        "apk_flux_fix_plan_run": (i, [vp, vp, d, i, d, vp]),
        "apk_update_with_flux_divergence": (i, [vp, vp, vp, d, d, d, vp]),
        "apk_dedner_source": (i, [vp, vp, i, d, d, d, d, vp]),
        "apk_stage_fused": (i, [vp, vp, vp, C.POINTER(StageArgs), vp]),
        "apk_cons_to_prim": (i, [vp, vp, i, E, vp]),
        "apk_cons_to_prim_ghosts": (i, [vp, vp, i, E, vp]),
        "apk_cons_to_prim_faces": (i, [vp, vp, i, E, vp]),
        "apk_cons_to_prim_dt": (i, [vp, vp, i, E, i, vp]),
        "apk_cons_to_prim_dt_skip": (i, [vp, vp, i, E, i, vp, vp]),
        "apk_cons_to_prim_dt_select": (i, [vp, vp, i, E, i, vp, C.c_uint, vp]),
        "apk_cons_to_prim_faces_skip": (i, [vp, vp, i, E, vp, vp]),
        "apk_cons_to_prim_faces_dt": (i, [vp, vp, i, E, vp, vp]),
        "apk_cons_to_prim_ghosts_split": (i, [vp, vp, i, E, vp, i, vp]),
        "apk_stage_dt_read": (i, [vp, d, c_dp, vp]),
        "apk_stage_dt_flags_read": (i, [vp, d, c_dp, C.POINTER(C.c_uint), vp]),
        "apk_estimate_timestep": (i, [vp, vp, i, E, d, c_dp, vp]),
        "apk_first_order_flux_correct": (i, [vp, vp, vp, i, E, d, d, d, d, C.POINTER(ll), vp]),
        "apk_count_unphysical": (i, [vp, vp, i, C.POINTER(ll), vp]),
        "apk_history": (i, [vp, vp, i, c_dp, vp]),
        "apk_fmft_create": (i, [vp, C.POINTER(FmftBlock), i, i, pp]),
        "apk_fmft_destroy": (None, [vp]),
        "apk_fmft_inverse": (i, [vp, vp, vp, c_dp, vp]),
        "apk_turb_mean_momentum": (i, [vp, vp, vp, c_dp, vp]),
        "apk_turb_remove_mean": (i, [vp, vp, vp, c_dp, c_dp, vp]),
        "apk_turb_apply": (i, [vp, vp, vp, d, d, vp]),
        "apk_turb_apply_fill": (i, [vp, vp, vp, d, d, i, E, i, vp]),
        "apk_turb_apply_dt": (i, [vp, vp, vp, d, d, i, E, vp]),
        "apk_turbulence_history": (i, [vp, vp, i, d, c_dp, vp]),
        "apk_history_user_reldivb": (i, [vp, vp, d, c_dp, vp]),
        "apk_refine_plan_create": (i, [vp, C.POINTER(RefineGeom), i, C.POINTER(RefineOp), i, pp]),
        "apk_refine_plan_destroy": (None, [vp]),
        "apk_refine_plan_run": (i, [vp, vp, vp]),
        "apk_tag_blocks": (i, [vp, vp, i, d, d, C.POINTER(C.c_int), c_dp, vp]),
        "apk_tag_blocks_begin": (i, [vp, vp, i, C.POINTER(C.c_int), vp]),
        "apk_tag_blocks_begin_skip": (i, [vp, vp, i, vp, C.POINTER(C.c_int), vp]),
        "apk_tag_blocks_dt_from_cons": (i, [vp, vp, i, E, vp, C.POINTER(C.c_int), vp]),
        "apk_tag_blocks_end": (i, [vp, i, i, i, d, d, C.POINTER(C.c_int), c_dp, vp]),
        "apk_poll_device_flags": (i, [vp, C.POINTER(C.c_uint), vp]),
        "apk_trial_flags": (i, [vp, i, vp]),
        "apk_stage_split_axis": (i, [vp, vp, i]),
        "apk_stage_single_march": (i, [vp, vp]),
        "apk_bench_scheme_floor": (i, [i, i, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
        "apk_stage_x1_halo": (i, [vp, vp, E, i, i, i]),
        "apk_stage_unphysical_read": (i, [vp, C.POINTER(C.c_longlong), vp]),
        "apk_copy_plan_create": (i, [vp, C.POINTER(CopyRegion), i, pp]),
        "apk_copy_plan_destroy": (None, [vp]),
        "apk_copy_plan_run": (i, [vp, vp, vp]),
        "apk_copy_plan_run_c2p": (i, [vp, vp, i, E, C.c_int64, i, vp]),
        "apk_copy_plan_run_c2p_prim_only": (i, [vp, vp, i, E, C.c_int64, i, vp]),
        "apk_kernel_timing_enable": (i, [vp, i]),
        "apk_kernel_timing_read": (i, [vp, i, c_dp, C.POINTER(ll)]),
        # apk_host.h
        "apk_sim_create": (i, [C.c_char_p, strs, i, i, i, C.POINTER(Allocator), C.POINTER(CommOps),
                               vp, pp, C.c_char_p, C.c_size_t]),
        "apk_sim_create_host_only": (i, [C.c_char_p, strs, i, i, i, pp, C.c_char_p, C.c_size_t]),
        "apk_sim_destroy": (None, [vp]),
        "apk_sim_last_error": (C.c_char_p, [vp]),
        "apk_sim_initialize": (i, [vp]),
        "apk_sim_step": (i, [vp]),
        "apk_sim_run": (i, [vp, i, C.POINTER(C.c_int)]),
        "apk_sim_time": (d, [vp]),
        "apk_sim_dt": (d, [vp]),
        "apk_sim_tlim": (d, [vp]),
        "apk_sim_c_h": (d, [vp]),
        "apk_sim_ncycle": (i, [vp]),
        "apk_sim_fofc_count": (ll, [vp]),
        "apk_sim_set_fused": (i, [vp, i]),
        "apk_sim_set_overlap": (i, [vp, i]),
        "apk_sim_overlapped_exchanges": (ll, [vp]),
        "apk_sim_skipped_local_exchanges": (ll, [vp]),
        "apk_sim_amr_c2p_passes_skipped": (ll, [vp]),
        "apk_sim_set_thin_exchange": (i, [vp, i]),
        "apk_sim_thin_exchanges": (ll, [vp]),
        "apk_sim_set_x1_direct": (i, [vp, i]),
        "apk_sim_x1_direct_exchanges": (ll, [vp]),
        "apk_sim_set_direct_neighbors": (C.c_int, [vp, C.c_int]),
        "apk_sim_set_amr_full_exchange": (C.c_int, [vp, C.c_int]),
        "apk_sim_set_prim_free": (C.c_int, [vp, C.c_int]),
        "apk_sim_prim_is_stale": (C.c_int, [vp]),
        "apk_sim_turb_dt_kicks": (ll, [vp]),
        "apk_sim_loop_seconds": (d, [vp]),
        "apk_sim_loop_cycles": (i, [vp]),
        "apk_sim_get_info": (i, [vp, C.POINTER(SimInfo)]),
        "apk_sim_block_location": (i, [vp, i, C.POINTER(C.c_int), C.POINTER(C.c_int * 3)]),
        "apk_sim_block_ptr": (vp, [vp, i, i]),
        "apk_sim_gather": (i, [vp, i, c_dp]),
        "apk_sim_read_block": (i, [vp, i, i, c_dp]),
        "apk_sim_write_block": (i, [vp, i, i, c_dp]),
        "apk_sim_history": (i, [vp, c_dp]),
        "apk_sim_linear_wave_errors": (i, [vp, c_dp, c_dp, c_dp]),
        "apk_sim_linear_wave_mhd_errors": (i, [vp, c_dp, c_dp, c_dp]),
        "apk_sim_cpaw_errors": (i, [vp, c_dp, c_dp]),
        "apk_sim_write_cpaw_errors": (i, [vp, C.c_char_p]),
        "apk_sim_check_refinement": (i, [vp, C.POINTER(C.c_int), c_dp]),
        "apk_sim_history_labels": (i, [vp, C.c_char_p, C.c_size_t]),
        "apk_sim_write_history": (i, [vp, C.c_char_p]),
        "apk_sim_write_linear_wave_errors": (i, [vp, C.c_char_p]),
        "apk_sim_execute": (i, [vp, C.c_char_p, C.POINTER(C.c_int)]),
        "apk_sim_turbulence_history": (i, [vp, c_dp]),
        "apk_sim_user_reldivb": (i, [vp, c_dp]),
        "apk_sim_fmft_num_modes": (i, [vp]),
        "apk_sim_fmft_var_hat": (i, [vp, c_dp]),
        "apk_sim_fmft_evolve": (i, [vp, d]),
        "apk_sim_fmft_phases": (i, [vp, i, i, i, c_dp]),
        "apk_sim_read_acc": (i, [vp, i, c_dp]),
        "apk_sim_exchange_ghosts": (i, [vp]),
        "apk_sim_fill_derived": (i, [vp]),
        "apk_sim_estimate_timestep": (i, [vp, c_dp]),
        "apk_sim_reset_time_step": (C.c_int, [vp]),
        "apk_sim_kernel_timing_enable": (i, [vp, i]),
        "apk_sim_kernel_timing_read": (i, [vp, i, c_dp, C.POINTER(ll)]),
        "apk_sim_peer": (i, [vp, i, C.POINTER(PeerInfo)]),
        "apk_sim_plan_size": (i, [vp, i]),
        "apk_sim_plan_region": (i, [vp, i, i, C.POINTER(RegionInfo)]),
        "apk_sim_num_peers": (i, [vp]),
        "apk_sim_select_messages": (i, [vp, i]),
        "apk_sim_message_generation": (ll, [vp]),
        "apk_sim_amr_ops_size": (i, [vp, i]),
        "apk_sim_amr_op": (i, [vp, i, i, C.POINTER(AmrOpInfo)]),
        "apk_sim_loop_zone_cycles": (ll, [vp]),
        "apk_sim_fofc_fallback_stages": (ll, [vp]),
        "apk_sim_block_level": (i, [vp, i]),
        "apk_sim_amr_stats": (i, [vp, C.POINTER(ll), C.POINTER(ll), C.POINTER(i), C.POINTER(ll)]),
        "apk_sim_regrid": (i, [vp, C.POINTER(i)]),
        "apk_rccl_unique_ids": (i, [C.c_char_p, C.c_size_t]),
        "apk_sim_comm_rccl": (i, [vp, C.c_char_p, C.c_size_t]),
        "apk_sim_comm_stats": (i, [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
        "apk_sim_comm_error": (C.c_char_p, [vp]),
        "apk_rccl_selftest": (i, [i, C.c_char_p, C.c_size_t]),
        "apk_sim_amr_apply_tags": (i, [vp, C.POINTER(i), i, C.POINTER(i)]),
    }


SYMBOLS = tuple(_signatures().keys())


def lib_path(strict=False):
    # APK_LIB_PATH: profiling aid, an A/B variant of the product build (csrc/Makefile `variant`)
    if not strict and os.environ.get("APK_LIB_PATH"):
        return os.path.abspath(os.environ["APK_LIB_PATH"])
    return os.path.join(_HERE, "libapk_amd_strict.so" if strict else "libapk_amd.so")


def build(strict=None, jobs=8):
    """Compile the gfx950 libraries in-tree with hipcc (strict=None builds both)."""
    targets = []
    if strict in (None, False):
        targets.append("../libapk_amd.so")
    if strict in (None, True):
        targets.append("../libapk_amd_strict.so")
    subprocess.run(["make", "-C", _CSRC, "-j%d" % jobs] + targets, check=True,
                   stdout=subprocess.DEVNULL)
    return [os.path.normpath(os.path.join(_CSRC, t)) for t in targets]


_LIBS = {}


def load(strict=False):
    """Load libapk_amd[_strict].so.  Raises if it has not been built."""
    key = bool(strict)
    if key not in _LIBS:
        path = lib_path(strict)
        if not os.path.exists(path):
            raise RuntimeError(
                "%s not found: run athenapk_amd.lib.build() / __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
        lib = C.CDLL(path)
        # (APK_LIB_PATH: a profiling variant, possibly built from an older commit for a same-box comparison -- entry
        # points it lacks are left unbound; the product library must export every declared symbol)
        variant = bool(os.environ.get("APK_LIB_PATH")) and not strict
        skipped = []
        for name, (res, args) in _signatures().items():
            if variant and not hasattr(lib, name):
                skipped.append(name)
                continue
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if skipped:  # (said aloud: a call through an unbound entry point would go out with default int argtypes)
            import warnings
            warnings.warn("APK_LIB_PATH=%s lacks %d declared entry point(s), left unbound: %s"
                          % (path, len(skipped), ", ".join(skipped)))
        _LIBS[key] = lib
    return _LIBS[key]


class ApkError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("apk error %d: %s" % (code, msg))
        self.code = code
