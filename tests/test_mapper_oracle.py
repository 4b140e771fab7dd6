"""The mapper-glue oracle (oracle/mapper_oracle.py) against the fixture produced by the reference's own statements
(tests/golden/tiny_mapper_iters.npz).  CPU only."""
import numpy as np
import torch

from tests.util import load, rel_err, tiny_scene


def test_mapper_iterations_match_reference_statements():
    from oracle import mapper_oracle as M
    g = load("tiny_mapper_iters")
    params, grids, bound, s = tiny_scene()
    masks = {k: torch.from_numpy(g['mask_' + k]) for k in M.KEYS}
    ro, rd = torch.from_numpy(s['rays_o']), torch.from_numpy(s['rays_d'])
    gd, gc = torch.from_numpy(s['gt_depth']), torch.from_numpy(s['gt_color'])
    n = int(g['num_joint_iters'])
    assert [M.stage_of(i, n) for i in range(n)] == list(g['stages'])
    torch.set_num_threads(4)
    losses = M.optimize_map_iters(params, grids, masks, ro, rd, gd, gc, bound, n, w_color=float(g['w_color_loss']),
                                  lr_factor=float(g['lr_factor']))
    assert rel_err(losses, g['losses']) <= 1e-6
    for k in M.KEYS:
        ref = g['final_' + k]
        got = grids[k].numpy()
        # the update is lr * m/(sqrt(v)+eps): bounded by lr per step whatever the gradient, and smooth in it
        assert np.abs(got - ref).max() <= 2e-5, k
        unmasked = ~g['mask_' + k]
        assert np.array_equal(got[0, :, unmasked], s[k][0, :, unmasked]), k       # unmasked voxels never move
    for name, ref in g.items():
        if name.startswith('final_cd_'):
            got = params['color_decoder.' + name[len('final_cd_'):]].detach().numpy()
            assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), name
