"""Flush-to-zero modes (ohs_*_set_flush_denormals): the shipped plugin runs Plugin::process inside nih-plug's FTZ scope
(src/lib.rs:1156), the reference's unit tests -- and every other parity test here -- with IEEE denormals.
An EQ state decaying into silence takes a different path in the two modes (parametric_eq.rs:116-122): with IEEE
arithmetic it creeps through the denormal range, with FTZ it drops to zero at 1.18e-38.  The kernels follow the mode
through the MODE register; the oracle twin sets MXCSR on the calling thread.  Bit-exact in every mode with the
exact-specials kernel; the default (ring) kernel keeps its one documented deviation, which flushing makes visible: a
result flushed to -0.0 leaves the cascade as +0.0 (its pass-on lanes compute 1 * x + 0; include/ohs_hip.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ohs():
    import open_headstage_amd
    return open_headstage_amd


def _decay_bands(ohs):
    from open_headstage_amd import BandConfig, FilterType
    return [BandConfig(FilterType.LowPass, 900.0, 0.6, 0.0, True), BandConfig(FilterType.Peak, 300.0, 2.0, 6.0, True),
            BandConfig(FilterType.HighShelf, 6000.0, 0.7, -4.0, True)]


def _run_pair(ohs, oracle, bands, x, mode, exact_specials=False):
    eg = ohs.StereoParametricEQ.new(len(bands), 48000.0)
    eo = oracle.StereoParametricEQ(len(bands), 48000.0)
    for i, b in enumerate(bands):
        eg.update_band_coeffs(i, 48000.0, b)
        c, en = eg.get_band_coeffs(i)
        eo.set_band_coeffs(i, c, en)
    eg.set_flush_denormals(mode)
    eg.set_exact_specials(exact_specials)
    gl, gr = x[0].copy(), x[1].copy()
    ol, orr = x[0].copy(), x[1].copy()
    for pos in range(0, x.shape[1], 1000):          # state carried across calls
        a, b = gl[pos:pos + 1000].copy(), gr[pos:pos + 1000].copy()
        eg.process_block(a, b)
        gl[pos:pos + 1000], gr[pos:pos + 1000] = a, b
    with oracle.flush_denormals(mode):
        for pos in range(0, x.shape[1], 1000):
            a, b = ol[pos:pos + 1000].copy(), orr[pos:pos + 1000].copy()
            eo.process_block(a, b)
            ol[pos:pos + 1000], orr[pos:pos + 1000] = a, b
    return np.stack([gl, gr]), np.stack([ol, orr])


def _assert_same(g, o, exact, what):
    if exact:
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32)), what
    else:       # every value equal; the only bit patterns that may differ are the two zeros
        assert np.array_equal(g, o), what
        d = g.view(np.uint32) != o.view(np.uint32)
        assert np.all(g[d] == 0.0) and np.all(o[d] == 0.0), what


def _denormal(y):
    a = np.abs(y)
    return (a > 0) & (a < np.float32(1.1754944e-38))


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_eq_state_through_the_denormal_range(ohs, oracle, mode, exact):
    bands = _decay_bands(ohs)
    n = 6000
    x = np.zeros((2, n), np.float32)
    x[0, 0], x[1, 3] = 3e-33, -7e-34           # normal numbers; the filters' tails decay through 1e-38 ... 1e-45
    x[0, 3000:3010] = 1e-36                    # and are excited again later
    g, o = _run_pair(ohs, oracle, bands, x, mode, exact_specials=exact)
    _assert_same(g, o, exact or mode == 0, f"mode {mode}: HIP and oracle differ")
    if mode == 0:
        assert _denormal(g).sum() > 100         # the IEEE path really walks through denormals
    else:
        assert _denormal(g).sum() == 0          # flushed: nothing denormal ever comes out
        assert np.count_nonzero(g[0, 2500:3000]) == 0       # ... and the left channel's tail has dropped to exact silence


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_eq_in_every_mode_matches_the_committed_golden_bits(ohs, mode, exact):
    """The same cases against bits the oracle produced on the AUTHORING machine (tests/golden/eq_ftz_modes.npz, written by
    tools/make_golden.py under MXCSR = IEEE / FTZ / FTZ | DAZ): no dependence on how this box's CPU treats MXCSR."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "eq_ftz_modes.npz"))
    call = int(g["call"])
    for tag, xk, ck in (("y", "x", "coeffs"), ("y_in", "x_in", "coeffs_in")):
        if tag == "y_in" and mode == 1 and not exact:
            continue        # the ring kernel's documented deviation: a denormal INPUT sample is read as zero in mode 1
        eq = ohs.StereoParametricEQ.new(3, 48000.0)
        for i in range(3):
            eq.set_band_coeffs(i, g[ck][i], True)
        eq.set_flush_denormals(mode)
        eq.set_exact_specials(exact)
        x = g[xk]
        l, r = x[0].copy(), x[1].copy()
        for pos in range(0, x.shape[1], call):
            a, b = l[pos:pos + call].copy(), r[pos:pos + call].copy()
            eq.process_block(a, b)
            l[pos:pos + call], r[pos:pos + call] = a, b
        _assert_same(np.stack([l, r]), g[f"{tag}_mode{mode}"], exact or mode == 0, f"{tag}, mode {mode}, exact {exact}")


def test_modes_take_different_paths_into_silence(ohs, oracle):
    bands = _decay_bands(ohs)
    x = np.zeros((2, 4000), np.float32)
    x[0, 0] = 3e-33
    g0, _ = _run_pair(ohs, oracle, bands, x, 0)
    g1, _ = _run_pair(ohs, oracle, bands, x, 1)
    assert not np.array_equal(g0, g1)
    assert np.array_equal(g0[:, :8], g1[:, :8])         # the onset, far above the denormal range, is the same
    assert _denormal(g0).sum() > 100 and _denormal(g1).sum() == 0


def test_denormal_input_samples(ohs, oracle):
    """FTZ without DAZ reads a denormal input sample; the default (ring) EQ kernel reads it as zero (its pass-on lanes
    run 1 * x + 0 under the flush) -- the documented deviation.  The exact-specials kernel and mode 2 (DAZ: the
    reference reads it as zero too) are bit-exact."""
    bands = _decay_bands(ohs)
    bands[1].gain_db = 12.0
    x = np.zeros((2, 2000), np.float32)
    x[0, ::7] = np.float32(1.1e-38)             # denormal inputs; b0 > 1 of the peak band makes some products normal
    x[1, ::5] = np.float32(-9e-39)
    g, o = _run_pair(ohs, oracle, bands, x, 1, exact_specials=True)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    g, o = _run_pair(ohs, oracle, bands, x, 2)
    _assert_same(g, o, False, "mode 2")
    assert np.count_nonzero(g) == 0             # DAZ: nothing went in


def test_chain_and_batch_in_ftz_mode_within_parity(ohs, oracle):
    """The whole chain in mode 1 against the oracle in mode 1 (EQ bit-exact, the convolution within the usual bar):
    a burst followed by silence, so that the EQ states and the overlap tails both run out."""
    from open_headstage_amd import synth
    from tests.util import assert_parity
    taps, S, n_blocks = 512, 3, 24
    irs = synth.hrir_set(taps)
    bands = synth.eq_table()
    x = synth.white_noise(range(40, 40 + S), n_blocks * 512)
    x[:, :, 2048:] = 0.0
    import torch
    bp = ohs.BatchProcessor(S, num_bands=len(bands))
    for p in range(4):
        bp.set_ir(p, irs[p])
    coeffs = [ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db) for b in bands]
    for i, b in enumerate(bands):
        bp.set_band_coeffs(i, coeffs[i], b.enabled)
    bp.set_eq_enabled(True)
    bp.set_flush_denormals(1)
    y = bp.process(torch.from_numpy(x).cuda()).cpu().numpy()
    for s in range(S):
        eng = oracle.ConvolutionEngine()
        for p in range(4):
            eng.set_ir(p, irs[p])
        eq = oracle.StereoParametricEQ(len(bands), synth.FS)
        for i, b in enumerate(bands):
            eq.set_band_coeffs(i, coeffs[i], b.enabled)
        l, r = x[s, 0].copy(), x[s, 1].copy()
        with oracle.flush_denormals(1):
            oracle.chain_process(eng, eq, l, r, eq_enable=True, gain=1.0)
        assert_parity(y[s], np.stack([l, r]), f"FTZ chain, stream {s}")
    # the single-stream engine accepts the mode as well (and its clone inherits it)
    e = ohs.ConvolutionEngine.new()
    e.set_flush_denormals(1)
    for p in range(4):
        e.set_ir(p, irs[p])
    out = np.stack(e.process_block(x[0, 0, :1024], x[0, 1, :1024]))
    eo = oracle.ConvolutionEngine()
    for p in range(4):
        eo.set_ir(p, irs[p])
    with oracle.flush_denormals(1):
        ref = np.stack(eo.process_block(x[0, 0, :1024], x[0, 1, :1024]))
    assert_parity(out, ref, "FTZ engine")
    with pytest.raises(ohs.OhsError):
        e.set_flush_denormals(3)
