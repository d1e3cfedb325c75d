"""Speaker angles -> the four paths behind the C ABI (ohs_engine / ohs_batch / ohs_node_batch _set_speakers): the wiring
the reference plans (github_issues/sofa_implement_logic_select_extract_hrirs.md:5) between the CLAP parameters
az_l / el_l / az_r / el_r (src/lib.rs:120-128, discarded at :1170-1173), MySofa::get_hrtf_irs (src/sofa/loader.rs:136-199)
and ConvolutionEngine::set_ir.  The reference's data asset does not travel to the GPU box, so the SOFA file is written
here from scratch (tests/util.py: write_minimal_sofa) around the committed CIPIC subject-003 fixture: its +-30 degree
measurements sit at their real positions among decoys.  Checked: bit-identical to the Python wiring
(sofa.speaker_hrirs + sofa.load_into), only changed paths are re-loaded, untouched paths keep their history."""
import os

import numpy as np
import pytest

from tests.util import assert_parity, write_minimal_sofa

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FS = 48000.0


@pytest.fixture(scope="module")
def sofa_file(tmp_path_factory):
    g = np.load(os.path.join(GOLD, "cipic003_pm30.npz"))
    rng = np.random.default_rng(3)
    az = np.arange(0.0, 360.0, 5.0)
    pos = np.stack([az, np.zeros_like(az), np.ones_like(az)], 1)
    pos = np.concatenate([pos, np.stack([az[::3], np.full(az[::3].shape, 20.0), np.ones(az[::3].shape)], 1)])
    ir = 0.05 * rng.standard_normal((len(pos), 2, 200)) * np.exp(-np.arange(200) / 30.0)
    il, irr = int(np.argmin(np.abs(pos[:72, 0] - 30.0))), int(np.argmin(np.abs(pos[:72, 0] - 330.0)))
    ir[il, 0], ir[il, 1] = g["irs_44100"][0], g["irs_44100"][1]          # left speaker: (left ear, right ear)
    ir[irr, 0], ir[irr, 1] = g["irs_44100"][2], g["irs_44100"][3]
    path = str(tmp_path_factory.mktemp("sofa") / "cipic_like.sofa")
    write_minimal_sofa(path, ir, pos, 44100.0)
    return path, g


def _noise(seed, n):
    from open_headstage_amd import synth
    x = synth.white_noise([seed], n)[0]
    return x[0].copy(), x[1].copy()


def test_engine_set_speakers_is_the_python_wiring_bit_for_bit(sofa_file):
    import open_headstage_amd as ohs
    from open_headstage_amd import sofa
    path, g = sofa_file
    s = sofa.MySofa(path)
    a, b = ohs.ConvolutionEngine.new(), ohs.ConvolutionEngine.new()
    assert a.set_speakers(s, -30.0, 0.0, 30.0, 0.0, 1.0, FS) == 0b1111
    irs = sofa.speaker_hrirs(s, 30.0, 0.0, -30.0, 0.0, 1.0, FS)           # AES69 azimuths: positive = left
    sofa.load_into(b, irs)
    # the responses are the fixture's (resampled on this box: libm's sin / sqrt may round differently from the box
    # that wrote the fixture, hence a tolerance here and bits everywhere else)
    for p in range(4):
        assert irs[p].shape == g["irs_48000"][p].shape and np.allclose(irs[p], g["irs_48000"][p], rtol=0, atol=1e-6)
    l, r = _noise(1, 4096)
    ya, yb = np.stack(a.process_block(l, r)), np.stack(b.process_block(l, r))
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32)) and float(np.abs(ya).max()) > 0.01
    # same angles again, and a move that stays with the same nearest measurements: nothing is re-loaded
    assert a.set_speakers(s, -30.0, 0.0, 30.0, 0.0, 1.0, FS) == 0
    assert a.set_speakers(s, -31.5, 0.4, 28.9, -0.3, 1.0, FS) == 0
    l2, r2 = _noise(2, 2048)
    ya, yb = np.stack(a.process_block(l2, r2)), np.stack(b.process_block(l2, r2))
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
    # the right speaker moves to another measurement: paths Rsl, Rsr are re-loaded (their history and overlap reset,
    # convolution.rs:135-138), Lsl and Lsr keep theirs -- the Python wiring re-loads exactly those two
    assert a.set_speakers(s, -30.0, 0.0, 45.0, 0.0, 1.0, FS) == 0b1100
    new = sofa.speaker_hrirs(s, 30.0, 0.0, -45.0, 0.0, 1.0, FS)
    b.set_ir(ohs.ConvolutionPath.Rsl, new[2]); b.set_ir(ohs.ConvolutionPath.Rsr, new[3])
    l3, r3 = _noise(3, 3072)
    ya, yb = np.stack(a.process_block(l3, r3)), np.stack(b.process_block(l3, r3))
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
    # a set_ir from elsewhere makes that path count as changed
    a.set_ir(ohs.ConvolutionPath.Lsl, new[0][:50])
    assert a.set_speakers(s, -30.0, 0.0, 45.0, 0.0, 1.0, FS) == 0b0001
    # fs <= 0: the file's own samples
    c = ohs.ConvolutionEngine.new()
    assert c.set_speakers(s, -30.0, 0.0, 30.0, 0.0, 1.0, 0.0) == 0b1111
    d = ohs.ConvolutionEngine.new()
    sofa.load_into(d, [g["irs_44100"][p] for p in range(4)])
    yc, yd = np.stack(c.process_block(l, r)), np.stack(d.process_block(l, r))
    assert np.array_equal(yc.view(np.uint32), yd.view(np.uint32))


def test_batch_and_node_batch_set_speakers(sofa_file, oracle):
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import sofa, synth
    path, g = sofa_file
    s = sofa.MySofa(path)
    S, n_blocks = 5, 12
    irs = sofa.speaker_hrirs(s, 30.0, 0.0, -30.0, 0.0, 1.0, FS)
    ref = ohs.BatchProcessor(S)
    sofa.load_into(ref, irs)
    bp = ohs.BatchProcessor(S)
    assert bp.set_speakers(s, fs=FS) == 0b1111              # the plugin's defaults: az_l = -30, az_r = +30 (lib.rs:429-432)
    nb = ohs.NodeBatchProcessor(S, n_devices=1)
    assert nb.set_speakers(s, fs=FS) == 0b1111
    x = torch.from_numpy(synth.white_noise(range(S), n_blocks * 512)).cuda()
    yr = ref.process(x).clone()
    yb = bp.process(x).clone()
    yn = torch.empty_like(x)
    torch.cuda.synchronize()
    nb.process([x], [yn]); nb.sync()
    assert torch.equal(yr.view(torch.int32), yb.view(torch.int32)) and torch.equal(yr.view(torch.int32), yn.view(torch.int32))
    # against the oracle with the same four responses
    eo = oracle.ConvolutionEngine()
    for p in range(4):
        eo.set_ir(p, irs[p])
    xc = x.cpu().numpy()
    assert_parity(yb.cpu().numpy()[2], np.stack(eo.process_block(xc[2, 0], xc[2, 1])), "batch with SOFA speakers, stream 2")
    # one speaker moves: only its two paths travel (the node batch broadcasts two responses, not four)
    assert bp.set_speakers(s, az_l=-60.0, fs=FS) == 0b0011 and nb.set_speakers(s, az_l=-60.0, fs=FS) == 0b0011
    assert bp.set_speakers(s, az_l=-60.0, fs=FS) == 0 and nb.set_speakers(s, az_l=-60.0, fs=FS) == 0
    new = sofa.speaker_hrirs(s, 60.0, 0.0, -30.0, 0.0, 1.0, FS)
    ref.set_ir(0, new[0]); ref.set_ir(1, new[1])
    x2 = torch.from_numpy(synth.white_noise(range(50, 50 + S), n_blocks * 512)).cuda()
    yr, yb = ref.process(x2).clone(), bp.process(x2).clone()
    torch.cuda.synchronize()
    nb.process([x2], [yn]); nb.sync()
    assert torch.equal(yr.view(torch.int32), yb.view(torch.int32)) and torch.equal(yr.view(torch.int32), yn.view(torch.int32))
