"""k_eq_ring's two forms -- four chains per wave in 16-lane DPP rows, one chain per wave with the ring closed over all 64
lanes (csrc/eq_ring64_body.hpp) -- are the same DF2T cascade (parametric_eq.rs:116-122, 166-179): bit for bit against the
oracle for every call length around the wave ring's group of 48 samples, with the state carried from call to call (the
wave ring puts a band's state into its lane one step before the band's first sample and takes it out behind its last),
for 1 .. 12 enabled bands with disabled ones in between, and in the three denormal modes.  The form is forced through the
experiments build (Tuning::eq_form); the last test runs the product library's own choice."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ohs():
    import open_headstage_amd
    return open_headstage_amd


@pytest.fixture()
def synth():
    from open_headstage_amd import synth as s
    return s


@pytest.fixture(params=[2, 1], ids=["wave_ring", "rows"])
def forced_form(request, exp_tuning, monkeypatch):
    """StereoParametricEQ objects made by the experiments library with k_eq_ring's form forced"""
    from open_headstage_amd import _ffi, dsp
    monkeypatch.setattr(dsp, "lib", _ffi.experiments_lib)
    exp_tuning.DEFAULTS.setdefault("eq_form", "0")
    exp_tuning("eq_form", request.param)
    return request.param


def _pair(ohs, oracle, bands, fs=48000.0):
    eg = ohs.StereoParametricEQ.new(len(bands), fs)
    eo = oracle.StereoParametricEQ(len(bands), fs)
    for i, b in enumerate(bands):
        eg.update_band_coeffs(i, fs, b)
        c, en = eg.get_band_coeffs(i)
        eo.set_band_coeffs(i, c, en)       # the SAME coefficient bits (coefficients are inputs of the EQ)
    return eg, eo


def _same_bits(eg, eo, x, sizes, what):
    o = 0
    for n in sizes:
        gl, gr = x[0, o:o + n].copy(), x[1, o:o + n].copy()
        ol, orr = gl.copy(), gr.copy()
        eg.process_block(gl, gr)
        eo.process_block(ol, orr)
        assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)), (what, "L", n, o)
        assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32)), (what, "R", n, o)
        o += n


@pytest.mark.parametrize("n", [1, 5, 31, 32, 33, 47, 48, 49, 95, 96, 97, 143, 144, 145, 191, 192, 193, 239, 240, 241, 500, 1024, 4099])
def test_every_call_length_around_the_group_of_48(ohs, synth, oracle, forced_form, n):
    eg, eo = _pair(ohs, oracle, synth.eq_table())
    x = synth.white_noise([31], 4 * n + 7)[0]
    _same_bits(eg, eo, x, [n, n, 7, n, n], f"form {forced_form}")


@pytest.mark.parametrize("enabled", [[0], [3], [0, 1, 2], [1, 4, 5, 9], list(range(12)), [0, 2, 4, 6, 8, 10, 11, 13]])
def test_band_counts_and_disabled_bands(ohs, synth, oracle, forced_form, enabled):
    from open_headstage_amd import BandConfig, FilterType
    bands = [BandConfig(FilterType(i % 8), 90.0 * (i + 1) ** 1.7, 0.6 + 0.15 * i, (-1.0) ** i * (1.5 + 0.5 * i), i in enabled)
             for i in range(max(enabled) + 2)]
    eg, eo = _pair(ohs, oracle, bands)
    x = synth.white_noise([32], 9000)[0]
    _same_bits(eg, eo, x, [1000, 50, 2950, 5000], f"form {forced_form}, bands {enabled}")


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_denormal_modes(ohs, synth, oracle, forced_form, mode):
    """the state decays through the subnormal range behind the signal's end (IEEE) or drops to zero there (FTZ, FTZ | DAZ):
    the same values in every mode; under a flush the two zeros may differ (the ring's pass-on lanes compute 1 * x + 0:
    include/ohs_hip.h, tests/test_gpu_ftz.py)"""
    eg, eo = _pair(ohs, oracle, synth.eq_table())
    eg.set_flush_denormals(mode)
    x = synth.white_noise([33], 64000)[0]
    x[:, 1500:] = 0.0
    o = 0
    for n in [1000, 20000, 43000]:
        gl, gr = x[0, o:o + n].copy(), x[1, o:o + n].copy()
        ol, orr = gl.copy(), gr.copy()
        eg.process_block(gl, gr)
        with oracle.flush_denormals(mode):
            eo.process_block(ol, orr)
        g, r = np.stack([gl, gr]), np.stack([ol, orr])
        if mode == 0:
            assert np.array_equal(g.view(np.uint32), r.view(np.uint32)), (forced_form, mode, n)
        else:
            assert np.array_equal(g, r), (forced_form, mode, n)
            d = g.view(np.uint32) != r.view(np.uint32)
            assert np.all(g[d] == 0.0) and np.all(r[d] == 0.0), (forced_form, mode, n)
        o += n


def test_the_product_librarys_own_choice_on_long_calls(ohs, synth, oracle):
    """long calls on few chains take the wave ring by the library's own rule (no override, product library)"""
    eg, eo = _pair(ohs, oracle, synth.eq_table())
    x = synth.white_noise([34], 50000)[0]
    _same_bits(eg, eo, x, [20000, 9000, 21000], "library's choice")


def test_documented_edge_cases_of_the_wave_ring(ohs, synth, oracle, forced_form):
    """include/ohs_hip.h: the ring kernels' pass-on lanes compute 1 * x + 0.  A -0.0 sample may come out as +0.0 (equal as
    numbers, every other sample bit-exact); a non-finite input sample makes the output non-finite from up to 16 samples
    before it in the row form, up to 64 in the wave ring (48 samples travel through a chain's pass-on lanes at a time) --
    everything earlier is bit-exact, and from that sample on the reference's output is non-finite as well."""
    eg, eo = _pair(ohs, oracle, synth.eq_table())
    n = 12000
    x = synth.white_noise([35], n)[0]
    x[:, 1000:1100] = -0.0
    gl, gr = x[0].copy(), x[1].copy(); ol, orr = x[0].copy(), x[1].copy()
    eg.process_block(gl, gr); eo.process_block(ol, orr)
    assert np.array_equal(gl, ol) and np.array_equal(gr, orr)
    nz = ol != 0
    assert np.array_equal(gl[nz].view(np.uint32), ol[nz].view(np.uint32))
    eg, eo = _pair(ohs, oracle, synth.eq_table())
    x = synth.white_noise([36], n)[0]
    k = 7013
    x[0, k] = np.nan
    gl, gr = x[0].copy(), x[1].copy(); ol, orr = x[0].copy(), x[1].copy()
    eg.process_block(gl, gr); eo.process_block(ol, orr)
    reach = 64 if forced_form == 2 else 16
    assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32))
    assert np.array_equal(gl[:k - reach].view(np.uint32), ol[:k - reach].view(np.uint32))
    assert np.isnan(ol[k:]).all() and np.isnan(gl[k:]).all()
