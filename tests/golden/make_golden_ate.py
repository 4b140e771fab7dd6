#!/usr/bin/env python3
"""Golden fixture for the ATE evaluation: tests/golden/ate_cases.npz.  Runs only in the build
container (needs /root/reference): seeded trajectories -> the reference's own src/tools/eval_ate.py (`associate`,
`align`, `evaluate_ate`) -> expected matches, alignment and statistics."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402,F401  (sys.path / cwd / stubs for the reference)

import numpy  # noqa: E402
import numpy as np  # noqa: E402

from src.tools import eval_ate as EA  # noqa: E402


def rot(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def main():
    rng = np.random.default_rng(7)
    out = {}
    for case, (n, noise, reflect) in enumerate([(50, 0.01, False), (200, 0.05, False), (12, 0.2, True), (3, 0.0, False)]):
        t = np.linspace(0, 6, n)
        gt = np.stack([np.cos(t) * 2, np.sin(0.7 * t) * 1.5, 0.3 * t], 1) + 0.02 * rng.standard_normal((n, 3))
        R, tr = rot(rng), rng.standard_normal(3)
        est = (gt - tr) @ R + noise * rng.standard_normal((n, 3))          # = R^T (gt - tr): a rigidly moved, noisy copy
        if reflect:
            est[:, 2] *= -1                                                # forces the determinant branch of align()
        gt7 = np.concatenate([gt, rng.standard_normal((n, 4))], 1)
        est7 = np.concatenate([est, rng.standard_normal((n, 4))], 1)
        first = {i: gt7[i] for i in range(n)}
        second = {i: est7[i] for i in range(n)}
        res = EA.evaluate_ate(first, second, "")
        r, tv, err = EA.align(numpy.matrix(est.T), numpy.matrix(gt.T))
        out[f'c{case}_gt'] = gt7
        out[f'c{case}_est'] = est7
        out[f'c{case}_rot'] = np.asarray(r)
        out[f'c{case}_trans'] = np.asarray(tv)
        out[f'c{case}_err'] = np.asarray(err)
        out[f'c{case}_stats'] = np.array([res[k] for k in sorted(res.keys())], dtype=np.float64)
        out[f'c{case}_stat_keys'] = np.array('\n'.join(sorted(res.keys())))
    # association with non-integer, jittered stamps and an offset
    sa = {float(s): None for s in np.round(np.arange(0, 3, 0.1) + 0.004 * rng.standard_normal(30), 4)}
    sb = {float(s): None for s in np.round(np.arange(0.05, 3, 0.1)[:25] + 0.004 * rng.standard_normal(25), 4)}
    m = EA.associate(sa, sb, offset=-0.05, max_difference=0.02)
    out['assoc_a'] = np.array(sorted(sa.keys()))
    out['assoc_b'] = np.array(sorted(sb.keys()))
    out['assoc_matches'] = np.array(m, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, 'ate_cases.npz'), **out)
    print({k: v.shape for k, v in out.items() if k.endswith('stats')}, len(m), 'matches')


if __name__ == '__main__':
    main()
