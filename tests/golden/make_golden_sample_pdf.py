#!/usr/bin/env python3
"""Golden fixture for common.sample_pdf: tests/golden/sample_pdf.npz, from the reference's own
src/common.py:19-63 on seeded inputs (det and random draws; the random draw uses torch's CPU generator under a seed)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402,F401

import numpy as np  # noqa: E402
import torch  # noqa: E402

from src.common import sample_pdf  # noqa: E402


def main():
    g = torch.Generator().manual_seed(4)
    out = {}
    for case, (B, M, N) in enumerate([(7, 33, 16), (3, 5, 40), (2, 48, 8)]):
        bins = torch.sort(torch.rand(B, M, generator=g) * 4 + 0.1, dim=-1).values
        w = torch.rand(B, M - 1, generator=g)
        w[0, : (M - 1) // 2] = 0.0                       # a run of empty bins (denominator guard)
        out[f'c{case}_bins'], out[f'c{case}_w'] = bins.numpy(), w.numpy()
        out[f'c{case}_det'] = sample_pdf(bins, w, N, det=True, device='cpu').numpy()
        torch.manual_seed(100 + case)
        out[f'c{case}_rand'] = sample_pdf(bins, w, N, det=False, device='cpu').numpy()
        out[f'c{case}_n'] = np.array(N)
    np.savez_compressed(os.path.join(HERE, 'sample_pdf.npz'), **out)
    print({k: v.shape for k, v in out.items() if k.endswith('det')})


if __name__ == '__main__':
    main()
