"""ctypes binding of libenslam_hip.so (include/enslam_hip.h).  No torch types cross this boundary:
device pointers as integers, sizes, a hipStream_t.  Fails loudly when the library is missing."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int32, c_int64, c_size_t, c_void_p

LIB_PATH = os.environ.get("ENSLAM_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libenslam_hip.so")

STAGE = {'coarse': 0, 'middle': 1, 'fine': 2, 'color': 3}
MLP_COARSE, MLP_MIDDLE, MLP_FINE, MLP_COLOR = 0, 1, 2, 3
MLP_NAMES = ('coarse_decoder', 'middle_decoder', 'fine_decoder', 'color_decoder')
MAX_SMALL_TENSORS = 72      # ENS_ADAM_MAX_TENSORS: dense tensors per adam_tensors / bucket launch
GRID_NAMES = ('grid_coarse', 'grid_middle', 'grid_fine', 'grid_color')
# grids / decoders read by each stage (NICE.forward, decoder.py:312-342); fine reads grid_middle twice
STAGE_KINDS = {'coarse': (0,), 'middle': (1,), 'fine': (1, 2), 'color': (1, 2, 3)}


class MlpParams(ctypes.Structure):
    _fields_ = [("W", c_void_p * 5), ("b", c_void_p * 5), ("Wc", c_void_p * 5), ("bc", c_void_p * 5),
                ("Wo", c_void_p), ("bo", c_void_p), ("B", c_void_p)]


class Grid(ctypes.Structure):
    _fields_ = [("data", c_void_p), ("D", c_int32), ("H", c_int32), ("W", c_int32)]


class Scene(ctypes.Structure):
    _fields_ = [("bound", c_double * 6), ("coarse_bound", c_double * 6), ("grids", Grid * 4),
                ("packed", c_void_p * 4)]


class StepPlan(ctypes.Structure):             # enslam_step_plan
    _fields_ = [("stage", c_int32), ("n_rays", c_int32), ("n_lin", c_int32), ("n_surf", c_int32), ("lindisp", c_int32),
                ("act_light", c_int32), ("need_rays", c_int32), ("use_work_list", c_int32), ("loss_kind", c_int32), ("use_color", c_int32),
                ("w_color", ctypes.c_float), ("grid_mode", c_int32 * 4), ("par_grad", c_int32 * 4),
                ("grid_D", c_int32 * 4), ("grid_H", c_int32 * 4), ("grid_W", c_int32 * 4),
                ("bound", c_double * 6), ("coarse_bound", c_double * 6), ("params", MlpParams * 4),
                ("pgrad_off", (c_int64 * 23) * 4), ("pgrad_floats", c_int64), ("t_lin", c_void_p), ("t_surf", c_void_p)]


class StepLayout(ctypes.Structure):           # enslam_step_layout
    _fields_ = [("scratch_bytes", c_int64), ("grad_bytes", c_int64), ("out_bytes", c_int64), ("s_zero_bytes", c_int64),
                ("s_flags", c_int64 * 4), ("s_packed", c_int64 * 4), ("s_z", c_int64), ("s_dmax", c_int64), ("s_raw", c_int64),
                ("s_act", c_int64), ("s_work", c_int64), ("s_draw", c_int64), ("s_dgw", c_int64), ("s_vm", c_int64 * 4),
                ("s_gacc", c_int64 * 4), ("g_flat", c_int64), ("g_flat_floats", c_int64), ("g_packed", c_int64 * 4),
                ("g_ro", c_int64), ("g_rd", c_int64), ("g_counter", c_int64), ("g_nat", c_int64 * 4), ("g_params", c_int64),
                ("g_dense", c_int64 * 4), ("o_depth", c_int64), ("o_var", c_int64), ("o_rgb", c_int64), ("o_loss", c_int64),
                ("n_samples", c_int32), ("finish_needed", c_int32), ("inline_rays", c_int32), ("merged", c_int32)]


class EnslamError(RuntimeError):
    pass


_lib = None

_SIGS = {
    "enslam_abi_version": (ctypes.c_int, []),
    "enslam_arch": (c_char_p, []),
    "enslam_packed_floats": (c_size_t, [ctypes.c_int]),
    "enslam_packed_grad_floats": (c_size_t, [ctypes.c_int]),
    "enslam_pack_mlp": (ctypes.c_int, [ctypes.c_int, POINTER(MlpParams), c_void_p, c_void_p]),
    "enslam_pack_mlp_multi": (ctypes.c_int, [c_int32, POINTER(c_int32), POINTER(MlpParams), POINTER(c_void_p), c_void_p]),
    "enslam_unpack_mlp_grads": (ctypes.c_int, [ctypes.c_int, c_void_p, POINTER(MlpParams), c_void_p]),
    "enslam_unpack_mlp_grads_multi": (ctypes.c_int, [c_int32, POINTER(c_int32), POINTER(c_void_p), POINTER(MlpParams),
                                                     c_void_p]),
    "enslam_grids_convert": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_int32,
                                            c_void_p]),
    "enslam_mark_blocks": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene),
                                          POINTER(c_void_p), c_void_p]),
    "enslam_grids_convert_sparse": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64),
                                                   POINTER(c_void_p), POINTER(c_void_p), c_int32, c_void_p]),
    "enslam_pose_rays_fwd": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_float, c_void_p, c_void_p, c_void_p]),
    "enslam_pose_rays_bwd": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_tracker_loss_fwd": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float,
                                               c_void_p, c_void_p]),
    "enslam_tracker_loss_bwd": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_adam_masked": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                          POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p), POINTER(c_void_p),
                                          ctypes.c_double, ctypes.c_double, ctypes.c_double, c_void_p]),
    "enslam_adam_tensors": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                           POINTER(c_int64), c_void_p, c_void_p, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, c_void_p]),
    "enslam_adam_tensors_step": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                           POINTER(c_int64), c_void_p, c_void_p, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_double, c_void_p]),
    "enslam_bucket_pack": (ctypes.c_int, [c_int32, POINTER(c_void_p), c_int32, POINTER(c_int64), POINTER(c_int32), c_void_p,
                                          c_void_p, c_int32, POINTER(c_void_p), POINTER(c_int64), c_int64, c_void_p, c_void_p]),
    "enslam_bucket_unpack": (ctypes.c_int, [c_int32, POINTER(c_void_p), c_int32, POINTER(c_int64), POINTER(c_int32), c_void_p,
                                            c_void_p, c_int32, POINTER(c_void_p), POINTER(c_int64), c_int64, c_void_p, c_void_p]),
    "enslam_mark_blocks_g": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene),
                                            POINTER(c_void_p), c_int32, c_void_p]),
    "enslam_bucket_pack_g": (ctypes.c_int, [c_int32, POINTER(c_void_p), c_int32, POINTER(c_int64), POINTER(c_int32), c_void_p,
                                            c_void_p, c_int32, POINTER(c_void_p), POINTER(c_int64), c_int64, c_void_p, c_int32, c_void_p]),
    "enslam_bucket_unpack_g": (ctypes.c_int, [c_int32, POINTER(c_void_p), c_int32, POINTER(c_int64), POINTER(c_int32), c_void_p,
                                              c_void_p, c_int32, POINTER(c_void_p), POINTER(c_int64), c_int64, c_void_p, c_int32, c_void_p]),
    "enslam_step_prepare": (ctypes.c_int, [c_int32, POINTER(c_int32), POINTER(MlpParams), POINTER(c_void_p),
                                           c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                           POINTER(c_void_p), c_int32, POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                           c_void_p, c_int64, c_void_p]),
    "enslam_step_finish": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                          c_int32, POINTER(c_int32), POINTER(c_void_p), POINTER(MlpParams), c_void_p]),
    "enslam_step_finish_rays": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                               c_int32, POINTER(c_int32), POINTER(c_void_p), POINTER(MlpParams), c_int32, c_int32,
                                               c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene), c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p]),
    "enslam_step_finish_rays_prev": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                                    POINTER(c_void_p), c_int32, POINTER(c_int32), POINTER(c_void_p), POINTER(MlpParams),
                                                    c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene), c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_gather_pixels": (ctypes.c_int, [c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_ray_grad_bwd": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene), c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "enslam_zero_blocks": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p), c_void_p,
                                          c_int64, c_void_p]),
    "enslam_grid_to_voxel_major": (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "enslam_grid_from_voxel_major": (ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "enslam_sample_rays": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                          POINTER(c_double), c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                          c_int32, c_void_p, c_int32, POINTER(Scene), POINTER(c_void_p), c_void_p]),
    "enslam_sample_rays_g": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                            POINTER(c_double), c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                            c_int32, c_void_p, c_int32, POINTER(Scene), POINTER(c_void_p), c_int32, POINTER(c_void_p),
                                            c_void_p]),
    "enslam_render_fwd": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene),
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "enslam_activation_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "enslam_grid_handoff_floats": (c_size_t, [c_int32, c_int32, c_int32]),
    "enslam_eval_points": (ctypes.c_int, [c_int32, c_int64, c_void_p, POINTER(Scene), c_int32, c_void_p, c_void_p]),
    "enslam_render_bwd": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene),
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(Grid),
                                         POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                         c_void_p]),
    "enslam_render_loss_fwd": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene), c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, ctypes.c_float,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_composite_loss_bwd": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_composite_bwd_list": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_composite_fwd": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p]),
    "enslam_composite_bwd": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p]),
    "enslam_decoder_bwd": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene),
                                          c_void_p, c_void_p, c_int32, c_void_p, POINTER(Grid), POINTER(c_void_p), c_void_p,
                                          c_void_p, c_void_p]),
    "enslam_decoder_bwd_scaled": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene),
                                                 c_void_p, c_void_p, c_void_p, c_int32, c_void_p, POINTER(Grid), POINTER(c_void_p),
                                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_bwd_partial_floats": (c_size_t, [ctypes.c_int]),
    "enslam_decoder_bwd_partials": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene),
                                                   c_void_p, c_void_p, c_void_p, c_int32, c_void_p, POINTER(Grid), POINTER(c_void_p),
                                                   POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_step_finish_partials": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                                   POINTER(c_void_p), c_int32, POINTER(c_int32), POINTER(c_void_p), POINTER(c_void_p),
                                                   POINTER(MlpParams), c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                                   POINTER(Scene), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_step_finish_native": (ctypes.c_int, [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                                 POINTER(c_void_p), c_int32, POINTER(c_int32), POINTER(c_void_p), POINTER(c_void_p),
                                                 POINTER(MlpParams), c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                                 POINTER(Scene), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_int64, c_void_p]),
    "enslam_sample_prepare": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                             POINTER(c_double), c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                             c_int32, c_void_p, c_int32, POINTER(Scene), POINTER(c_void_p), c_int32, POINTER(c_void_p),
                                             c_int32, POINTER(c_int32), POINTER(MlpParams), POINTER(c_void_p),
                                             c_int32, POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p), c_void_p, c_int64,
                                             c_void_p]),
    "enslam_tracker_rays": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                           ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, POINTER(c_double), c_int32,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int32, c_void_p]),
    "enslam_tracker_tail_max_rays": (ctypes.c_int, []),
    "enslam_render_tracker_loss_fwd": (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(Scene),
                                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                                      ctypes.c_float, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                      c_void_p]),
    "enslam_plan_struct_bytes": (c_int64, [c_int32]),
    "enslam_plan_layout": (ctypes.c_int, [POINTER(StepPlan), POINTER(StepLayout)]),
    "enslam_plan_forward": (ctypes.c_int, [POINTER(StepPlan), POINTER(StepLayout), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, POINTER(c_void_p), c_void_p]),
    "enslam_plan_backward": (ctypes.c_int, [POINTER(StepPlan), POINTER(StepLayout), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_rgbd_loss_fwd": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p,
                                            c_void_p]),
    "enslam_rgbd_loss_bwd": (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p,
                                            c_void_p, c_void_p, c_void_p]),
    "enslam_voxel_index": (ctypes.c_int, [c_int64, c_void_p, POINTER(c_double), c_int32, c_int32, c_int32,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "enslam_ray_points": (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, POINTER(c_double),
                                         c_void_p, c_void_p, c_void_p]),
    "enslam_fourier_sincos": (ctypes.c_int, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
}
EXPORTS = tuple(_SIGS)


def lib():
    """The loaded library; raises EnslamError if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EnslamError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                              f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        # diagnostic A/B runs against an OLDER build (ENSLAM_LIB=<path> ENSLAM_LIB_ALLOW_MISSING=1) may lack newer entry
        # points; the shipped library must export every one of them (tests/test_host_cpu.py checks the header against it)
        allow_missing = bool(os.environ.get("ENSLAM_LIB")) and os.environ.get("ENSLAM_LIB_ALLOW_MISSING") == "1"
        for name, (res, args) in _SIGS.items():
            if allow_missing and not hasattr(handle, name):
                continue
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.enslam_abi_version() != 1:
            raise EnslamError("libenslam_hip.so ABI version mismatch")
        if hasattr(handle, "enslam_plan_struct_bytes") and (handle.enslam_plan_struct_bytes(0) != ctypes.sizeof(StepPlan) or
                                                           handle.enslam_plan_struct_bytes(1) != ctypes.sizeof(StepLayout)):
            raise EnslamError("libenslam_hip.so: enslam_step_plan / enslam_step_layout do not have the sizes this binding declares")
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        raise EnslamError(f"{what} failed with code {code} "
                          f"({ {-1: 'EINVAL', -2: 'ELAUNCH', -3: 'EUNSUPPORTED'}.get(code, '?')})")
