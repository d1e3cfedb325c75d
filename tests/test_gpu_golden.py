"""HIP path (through the C ABI) against the committed golden vectors in tests/golden
(f64 ground truth; provenance in tools/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from tests.util import assert_parity

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "conv_*.npz"))))
def test_engine_matches_golden_f64(path):
    import open_headstage_amd as ohs
    g = np.load(path)
    e = ohs.ConvolutionEngine.new()
    for p in range(4):
        e.set_ir(p, g["irs"][p])
    x = g["x"]
    outs = [np.stack(e.process_block(x[0, c * 1024:(c + 1) * 1024], x[1, c * 1024:(c + 1) * 1024]))
            for c in range(x.shape[1] // 1024)]
    assert_parity(np.concatenate(outs, axis=1), g["y_f64"], os.path.basename(path))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "conv_*.npz"))))
def test_batch_matches_golden_f64(path):
    """same vectors through the batch path (time-parallel P=1 kernel or the general kernel)"""
    import torch
    import open_headstage_amd as ohs
    g = np.load(path)
    bp = ohs.BatchProcessor(1, num_bands=10)
    for p in range(4):
        bp.set_ir(p, g["irs"][p])
    y = bp.process(torch.from_numpy(g["x"][None].copy()).cuda()).cpu().numpy()[0]
    assert_parity(y, g["y_f64"], os.path.basename(path))


def test_eq_matches_golden_bits():
    import open_headstage_amd as ohs
    g = np.load(os.path.join(GOLD, "eq_11bands_3000.npz"))
    eq = ohs.StereoParametricEQ.new(len(g["enabled"]), 48000.0)
    for i in range(len(g["enabled"])):
        eq.set_band_coeffs(i, g["coeffs"][i], bool(g["enabled"][i]))
    l, r = g["x"][0].copy(), g["x"][1].copy()
    eq.process_block(l, r)
    assert np.array_equal(np.stack([l, r]).view(np.uint32), g["y_f32"].view(np.uint32))
