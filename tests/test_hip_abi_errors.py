"""-m gpu: the C ABI's error conventions (include/enslam_hip.h: 0 on success, negative ENSLAM_E* otherwise; no
exceptions, no aborts) exercised with bad arguments, and the Python binding turning them into EnslamError."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

EINVAL, ELAUNCH, EUNSUPPORTED = -1, -2, -3


def test_bad_arguments_return_error_codes():
    import evennicer_slam_amd as E
    L = E._lib
    lib = L.lib()
    dev = torch.device('cuda', 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    sc = L.Scene()
    ro = torch.zeros(4, 3, device=dev)
    z = torch.zeros(4, 48, dtype=torch.float64, device=dev)
    out_d = torch.zeros(4, dtype=torch.float64, device=dev)
    out_c = torch.zeros(4, 3, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    # empty scene: the stage's grids / decoders are missing
    assert lib.enslam_render_fwd(3, 4, 48, P(ro), P(ro), P(z), ctypes.byref(sc), P(out_d), P(out_d), P(out_c), None, None, 0, st) == EINVAL
    # unsupported sample count
    assert lib.enslam_render_fwd(3, 4, 40, P(ro), P(ro), P(z), ctypes.byref(sc), P(out_d), P(out_d), P(out_c), None, None, 0, st) == EUNSUPPORTED
    # negative ray count / zero rays
    assert lib.enslam_render_fwd(3, -1, 48, P(ro), P(ro), P(z), ctypes.byref(sc), P(out_d), P(out_d), P(out_c), None, None, 0, st) == EINVAL
    assert lib.enslam_render_fwd(3, 0, 48, None, None, None, ctypes.byref(sc), None, None, None, None, None, 0, st) == 0
    # NULL where a pointer is required
    assert lib.enslam_grid_to_voxel_major(None, P(ro), 4, st) == EINVAL
    assert lib.enslam_composite_fwd(4, 65, P(ro), P(z), P(out_d), P(out_d), P(out_c), None, st) != 0           # S > 64
    assert lib.enslam_adam_tensors(73, None, None, None, None, None, None, None, 0.9, 0.999, 1e-8, st) == EUNSUPPORTED
    assert lib.enslam_pose_rays_fwd(4, None, P(ro), P(ro), 1.0, 1.0, 0.0, 0.0, P(ro), P(ro), st) == EINVAL
    assert lib.enslam_tracker_loss_fwd(4, P(out_d), P(out_d), P(out_c), P(ro), None, 0.5, P(out_d), st) == EINVAL   # colour without gt
    assert lib.enslam_step_prepare(4, None, None, None, 0, None, None, None, None, None, 0, None, None, None, None, 0, st) == EINVAL
    assert lib.enslam_step_prepare(0, None, None, None, 0, None, None, None, None, None, 0, None, None, None, None, 0, st) == 0
    # entry points of the fused loss, the merged finish launch and the gradient bucket
    assert lib.enslam_render_loss_fwd(3, 4, 48, P(ro), P(ro), P(z), ctypes.byref(sc), P(out_d), P(out_d), P(out_c), P(out_c), None, 0,
                                      P(ro), None, 0.2, P(out_d), None, None, None, st) == EINVAL       # empty scene
    assert lib.enslam_render_loss_fwd(3, 0, 48, None, None, None, ctypes.byref(sc), None, None, None, None, None, 0, None, None, 0.2,
                                      None, None, None, None, st) == 0
    assert lib.enslam_decoder_bwd_scaled(3, 4, 48, P(ro), P(ro), P(z), ctypes.byref(sc), None, None, None, 0, None, None, None, None,
                                         None, None, None, st) == EINVAL
    assert lib.enslam_composite_bwd_list(4, 48, P(ro), P(z), P(out_d), None, None, None, P(ro), P(ro), None, st) == EINVAL   # list without count
    assert lib.enslam_composite_loss_bwd(4, 48, None, P(z), P(out_d), P(out_c), P(ro), None, 0.2, P(out_d), P(ro), None, None, st) == EINVAL
    assert lib.enslam_composite_loss_bwd(4, 48, P(ro), P(z), P(out_d), None, P(ro), P(out_c), 0.2, P(out_d), P(ro), None, None, st) == EINVAL  # colour without rgb
    assert lib.enslam_step_finish_rays(0, None, None, None, None, 0, None, None, None, 3, 4, 40, P(ro), P(ro), P(z), ctypes.byref(sc),
                                       P(ro), P(ro), P(ro), None, None, st) == EUNSUPPORTED           # sample count
    assert lib.enslam_step_finish_rays(0, None, None, None, None, 0, None, None, None, 3, 4, 48, P(ro), P(ro), P(z), ctypes.byref(sc),
                                       None, P(ro), P(ro), None, None, st) == EINVAL                  # (scene / hand-off missing)
    assert lib.enslam_step_finish_rays(0, None, None, None, None, 0, None, None, None, 0, 4, 32, None, None, None, None, None, None,
                                       None, None, None, st) == 0                                     # coarse: plain finish, nothing to do
    assert lib.enslam_bucket_unpack(0, None, 32, None, None, None, None, 73, None, None, 0, P(ro), st) == EUNSUPPORTED
    # round 2: 64 samples per ray are a supported tile count (the padded second pass of hierarchical sampling), 80 are not;
    # the Fourier parity entry
    z64 = torch.zeros(4, 64, dtype=torch.float64, device=dev)
    assert lib.enslam_render_fwd(3, 4, 64, P(ro), P(ro), P(z64), ctypes.byref(sc), P(out_d), P(out_d), P(out_c), None, None, 0, st) == EINVAL   # (empty scene, not EUNSUPPORTED)
    assert lib.enslam_render_fwd(3, 4, 80, P(ro), P(ro), P(z64), ctypes.byref(sc), P(out_d), P(out_d), P(out_c), None, None, 0, st) == EUNSUPPORTED
    x = torch.linspace(-3, 3, 7, device=dev)
    so, co = torch.empty_like(x), torch.empty_like(x)
    assert lib.enslam_fourier_sincos(-1, P(x), P(so), P(co), st) == EINVAL
    assert lib.enslam_fourier_sincos(7, None, P(so), P(co), st) == EINVAL
    assert lib.enslam_fourier_sincos(7, P(x), None, None, st) == EINVAL
    assert lib.enslam_fourier_sincos(0, None, None, None, st) == 0
    assert lib.enslam_fourier_sincos(7, P(x), P(so), None, st) == 0 and lib.enslam_fourier_sincos(7, P(x), None, P(co), st) == 0
    torch.cuda.synchronize()
    assert torch.allclose(so, torch.sin(x), atol=2e-7) and torch.allclose(co, torch.cos(x), atol=2e-7)
    torch.cuda.synchronize()
    assert lib.enslam_abi_version() >= 1
    assert lib.enslam_activation_floats(0, 100, 48, 0) == 0                     # the coarse stage keeps no workspace
    assert lib.enslam_activation_floats(3, 100, 48, 1) < lib.enslam_activation_floats(3, 100, 48, 0)


def test_binding_raises_enslam_error_with_the_entry_name():
    import evennicer_slam_amd as E
    with pytest.raises(E.EnslamError, match="some_entry"):
        E._lib.check(-1, "some_entry")
    assert E._lib.check(0, "ok") is None
    with pytest.raises(E.EnslamError):                         # CPU tensors never reach the kernels
        E.functional._require_hip(torch.zeros(3), "x")


def test_step_plan_entries_refuse_inconsistent_plans():
    """enslam_plan_layout / _forward / _backward: malformed plans and missing blobs return ENSLAM_EINVAL (nothing is launched)"""
    import evennicer_slam_amd as E
    L = E._lib
    lib = L.lib()
    dev = torch.device('cuda', 0)
    t_lin = torch.linspace(0, 1, 32, device=dev)
    t_surf = torch.linspace(0, 1, 16, dtype=torch.float64, device=dev)

    def plan(**kw):
        P = L.StepPlan()
        P.stage, P.n_rays, P.n_lin, P.n_surf = 3, 64, 32, 16
        P.act_light = 1
        for k in (1, 2, 3):
            P.grid_mode[k] = 1
            P.grid_D[k], P.grid_H[k], P.grid_W[k] = 4, 5, 6
        P.t_lin, P.t_surf = t_lin.data_ptr(), t_surf.data_ptr()
        for name, v in kw.items():
            if isinstance(v, dict):
                for i, x in v.items():
                    getattr(P, name)[i] = x
            else:
                setattr(P, name, v)
        return P

    Lay = L.StepLayout()
    assert lib.enslam_plan_layout(ctypes.byref(plan()), ctypes.byref(Lay)) == 0
    assert Lay.n_samples == 48 and Lay.inline_rays == 1 and Lay.merged == 1 and Lay.scratch_bytes > 0 and Lay.out_bytes > 0
    assert lib.enslam_plan_layout(ctypes.byref(plan(grid_mode={2: 3}, need_rays=1)), ctypes.byref(Lay)) == 0
    assert Lay.merged == 0 and Lay.finish_needed == 1 and Lay.s_vm[2] >= 0 and Lay.g_dense[2] >= 0 and Lay.s_dgw >= 0
    for bad in (plan(stage=0), plan(n_rays=0), plan(n_surf=8), plan(grid_mode={3: 0}), plan(grid_mode={0: 1}), plan(par_grad={2: 1}),
                plan(act_light=0), plan(loss_kind=2), plan(grid_D={1: 0}), plan(t_lin=None), plan(stage=2)):
        assert lib.enslam_plan_layout(ctypes.byref(bad), ctypes.byref(Lay)) == EINVAL
    assert lib.enslam_plan_layout(ctypes.byref(plan(stage=2, grid_mode={3: 0})), ctypes.byref(Lay)) == 0        # fine stage: two grids
    P = plan()
    assert lib.enslam_plan_layout(ctypes.byref(P), ctypes.byref(Lay)) == 0
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gv = (ctypes.c_void_p * 4)()
    ro = torch.zeros(64, 3, device=dev)
    gd = torch.ones(64, device=dev)
    blob = torch.zeros(max(Lay.scratch_bytes, Lay.grad_bytes, Lay.out_bytes), dtype=torch.uint8, device=dev)
    b = blob.data_ptr()
    # a used grid without values; missing blobs; missing rays
    assert lib.enslam_plan_forward(ctypes.byref(P), ctypes.byref(Lay), b, b, b, ro.data_ptr(), ro.data_ptr(), gd.data_ptr(), None, None, gv, st) == EINVAL
    assert lib.enslam_plan_forward(ctypes.byref(P), ctypes.byref(Lay), None, b, b, ro.data_ptr(), ro.data_ptr(), gd.data_ptr(), None, None, gv, st) == EINVAL
    assert lib.enslam_plan_forward(ctypes.byref(P), ctypes.byref(Lay), b, b, b, None, ro.data_ptr(), gd.data_ptr(), None, None, gv, st) == EINVAL
    assert lib.enslam_plan_backward(ctypes.byref(P), ctypes.byref(Lay), b, None, b, ro.data_ptr(), ro.data_ptr(), gd.data_ptr(), None, gv, None,
                                    None, None, None, st) == EINVAL
    torch.cuda.synchronize()
