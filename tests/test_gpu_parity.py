"""Parity of the HIP path (through the C ABI) against the CPU oracle and the f64 ground truth.

Bars (BASELINE.json north_star):
  * convolution : RMS(gpu - oracle) and the same relative to RMS(oracle) <= 1e-6,
                  and gpu vs the f64 direct convolution <= 1e-6 as well;
  * EQ          : BIT-EXACT against the oracle's DF2T recurrence (same operations, same order,
                  no FMA contraction on either side).
Synthetic inputs are the SURVEY section 8d generators (open_headstage_amd.synth).
"""
import numpy as np
import pytest

from tests.util import assert_parity, rms_err, setup_engines

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ohs():
    import open_headstage_amd
    return open_headstage_amd


@pytest.fixture(scope="module")
def synth():
    from open_headstage_amd import synth
    return synth


# ------------------------------------------------------------------ FFT primitive
def test_fft_primitive_accuracy():
    import ctypes as C
    from open_headstage_amd._ffi import check, fp, lib
    rng = np.random.default_rng(7)
    B = 8
    z = (rng.standard_normal((B, 1024)) + 1j * rng.standard_normal((B, 1024))).astype(np.complex64)
    out = np.empty_like(z)
    check(lib().ohs_debug_fft1024(0, z.view(np.float32).ctypes.data_as(fp),
                                  out.view(np.float32).ctypes.data_as(fp), 0, B))
    ref = np.fft.fft(z.astype(np.complex128), axis=1)
    rel = np.linalg.norm(out - ref) / np.linalg.norm(ref)
    assert rel < 2.5e-7, f"forward FFT rel err {rel:.3e}"
    back = np.empty_like(z)
    check(lib().ohs_debug_fft1024(0, out.view(np.float32).ctypes.data_as(fp),
                                  back.view(np.float32).ctypes.data_as(fp), 1, B))
    rel = np.linalg.norm(back / 1024 - z) / np.linalg.norm(z)
    assert rel < 4e-7, f"round-trip rel err {rel:.3e}"
    # a pure tone lands in exactly one bin (catches layout / index-map mistakes)
    k0 = 37
    tone = np.exp(2j * np.pi * k0 * np.arange(1024) / 1024).astype(np.complex64)[None, :]
    o1 = np.empty_like(tone)
    check(lib().ohs_debug_fft1024(0, tone.view(np.float32).ctypes.data_as(fp),
                                  o1.view(np.float32).ctypes.data_as(fp), 0, 1))
    assert abs(o1[0, k0] - 1024) < 1e-2 and np.abs(np.delete(o1[0], k0)).max() < 1e-2


# ------------------------------------------------------------------ convolution engine
@pytest.mark.parametrize("taps,host_block,n_calls", [
    (128, 1024, 1),      # BASELINE config 1: one 1024 block, 4 x 128 taps
    (512, 1024, 8),      # config 2 shape
    (512, 512, 5),
    (1300, 1024, 6),     # P = 3
    (16384, 1024, 40),   # config 4 shape: P = 32, long enough to wrap the history ring
])
def test_engine_matches_oracle_and_f64(ohs, synth, oracle, taps, host_block, n_calls):
    irs = synth.hrir_set(taps)
    eg = ohs.ConvolutionEngine.new()
    eo = setup_engines(oracle, eg, irs)
    x = synth.white_noise([3], host_block * n_calls)[0]
    outs_g, outs_o = [], []
    for c in range(n_calls):
        sl = slice(c * host_block, (c + 1) * host_block)
        outs_g.append(np.stack(eg.process_block(x[0, sl], x[1, sl])))
        outs_o.append(np.stack(eo.process_block(x[0, sl], x[1, sl])))
    g = np.concatenate(outs_g, axis=1)
    o = np.concatenate(outs_o, axis=1)
    f_l, f_r = oracle.binaural_f64(x[0], x[1], irs)
    f = np.stack([f_l, f_r])
    assert_parity(g, o, f"gpu vs oracle taps={taps}")
    assert_parity(g, f, f"gpu vs f64 taps={taps}")
    eo_err = rms_err(o, f)[1]
    print(f"taps={taps}: gpu-oracle {rms_err(g, o)[1]:.2e} gpu-f64 {rms_err(g, f)[1]:.2e} oracle-f64 {eo_err:.2e}")


@pytest.mark.parametrize("sizes", [
    [256, 256, 256, 256, 256, 256],          # < 512: zeros first, then a 256-sample delay
    [100, 924, 1, 511, 512, 1000, 37, 3000],  # ragged
    [1536, 64, 64, 64, 1024],
])
def test_engine_fifo_semantics_ragged_blocks(ohs, synth, oracle, sizes):
    """convolution.rs:141-182: FIFO adapter, zero fill while fewer than n outputs are buffered."""
    irs = synth.hrir_set(700)
    eg = ohs.ConvolutionEngine.new()
    eo = setup_engines(oracle, eg, irs)
    x = synth.white_noise([11], sum(sizes))[0]
    pos = 0
    for n in sizes:
        gl, gr = eg.process_block(x[0, pos:pos + n], x[1, pos:pos + n])
        ol, orr = eo.process_block(x[0, pos:pos + n], x[1, pos:pos + n])
        pos += n
        # silence fallback must agree exactly
        assert np.array_equal(ol == 0, gl == 0) or rms_err(gl, ol)[0] < 1e-6
        if np.any(ol) or np.any(orr):
            assert_parity(np.stack([gl, gr]), np.stack([ol, orr]), f"ragged n={n}")
        else:
            assert not np.any(gl) and not np.any(gr)


def test_engine_default_is_silence_and_empty_ir_mutes(ohs, synth, oracle):
    """Fresh engine = four all-zero paths (convolution.rs:46-63); set_ir(&[]) mutes (:114-118)."""
    eg = ohs.ConvolutionEngine.new()
    x = synth.white_noise([1], 1024)[0]
    gl, gr = eg.process_block(x[0], x[1])
    assert not np.any(gl) and not np.any(gr)
    irs = synth.hrir_set(512)
    eo = setup_engines(oracle, eg, irs)
    eg.set_ir(ohs.ConvolutionPath.Rsl, [])
    eo.set_ir(oracle.RSL, [])
    assert eg.num_partitions(ohs.ConvolutionPath.Rsl) == 1
    g = np.stack(eg.process_block(x[0], x[1]))
    o = np.stack(eo.process_block(x[0], x[1]))
    assert_parity(g, o, "after muting Rsl")


def test_engine_set_ir_mid_stream_resets_only_that_path(ohs, synth, oracle):
    """convolution.rs:135-138: per-path history / overlap reset, FIFOs and other paths untouched;
    paths may have different partition counts."""
    irs = synth.hrir_set(1500)         # P = 3
    eg = ohs.ConvolutionEngine.new()
    eo = setup_engines(oracle, eg, irs)
    x = synth.white_noise([5], 1024 * 6 + 300)[0]
    pos = 0
    outs_g, outs_o = [], []

    def run(n):
        nonlocal pos
        outs_g.append(np.stack(eg.process_block(x[0, pos:pos + n], x[1, pos:pos + n])))
        outs_o.append(np.stack(eo.process_block(x[0, pos:pos + n], x[1, pos:pos + n])))
        pos += n

    run(1024); run(1024)
    new_ir = synth.hrir_set(2600)[1]   # P = 6 on one path only -> the shared ring must grow
    eg.set_ir(ohs.ConvolutionPath.Lsr, new_ir); eo.set_ir(oracle.LSR, new_ir)
    run(300)                            # leaves 300 frames in the input FIFO across the next set_ir
    short = synth.hrir_set(100)[2]
    eg.set_ir(ohs.ConvolutionPath.Rsl, short); eo.set_ir(oracle.RSL, short)
    run(1024); run(1024); run(1024); run(1024)
    g = np.concatenate(outs_g, axis=1); o = np.concatenate(outs_o, axis=1)
    assert_parity(g, o, "set_ir mid-stream")
    assert [eg.num_partitions(p) for p in range(4)] == [eo.num_partitions(p) for p in range(4)]


def test_engine_clone_is_deep(ohs, synth, oracle):
    irs = synth.hrir_set(900)
    eg = ohs.ConvolutionEngine.new()
    eo = setup_engines(oracle, eg, irs)
    x = synth.white_noise([9], 4096)[0]
    eg.process_block(x[0, :1300], x[1, :1300]); eo.process_block(x[0, :1300], x[1, :1300])
    cg, co = eg.clone(), eo.clone()
    # originals and clones continue independently with different inputs
    g1 = np.stack(eg.process_block(x[0, 1300:2324], x[1, 1300:2324]))
    o1 = np.stack(eo.process_block(x[0, 1300:2324], x[1, 1300:2324]))
    g2 = np.stack(cg.process_block(x[1, 1300:2324], x[0, 1300:2324]))
    o2 = np.stack(co.process_block(x[1, 1300:2324], x[0, 1300:2324]))
    assert_parity(g1, o1, "original after clone")
    assert_parity(g2, o2, "clone")


def test_engine_rejects_length_mismatch(ohs):
    eg = ohs.ConvolutionEngine.new()
    with pytest.raises(ValueError):
        eg.process_block(np.zeros(8, np.float32), np.zeros(8, np.float32), np.zeros(4, np.float32),
                         np.zeros(8, np.float32))


# ------------------------------------------------------------------ EQ (bit-exact)
def _mk_eqs(ohs, oracle, bands, fs=48000.0):
    eg = ohs.StereoParametricEQ.new(len(bands), fs)
    eo = oracle.StereoParametricEQ(len(bands), fs)
    for i, b in enumerate(bands):
        eg.update_band_coeffs(i, fs, b)
        # the oracle is fed the SAME coefficient bits (8b: coefficients are inputs)
        c, en = eg.get_band_coeffs(i)
        eo.set_band_coeffs(i, c, en)
    return eg, eo


@pytest.mark.parametrize("n", [1, 15, 16, 17, 512, 1024, 4099])
def test_eq_bit_exact_vs_oracle(ohs, synth, oracle, n):
    eg, eo = _mk_eqs(ohs, oracle, synth.eq_table())
    x = synth.white_noise([2], 3 * n)[0]
    for c in range(3):      # state carried across calls
        gl = x[0, c * n:(c + 1) * n].copy(); gr = x[1, c * n:(c + 1) * n].copy()
        ol = gl.copy(); orr = gr.copy()
        eg.process_block(gl, gr)
        eo.process_block(ol, orr)
        assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)), f"L differs, call {c}"
        assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32)), f"R differs, call {c}"


def test_eq_all_filter_types_disabled_bands_and_stress(ohs, synth, oracle):
    from open_headstage_amd import BandConfig, FilterType
    bands = [BandConfig(FilterType(t), 300.0 * (t + 1), 0.5 + 0.3 * t, -6.0 + 2.0 * t, t != 3)
             for t in range(8)]
    bands.append(synth.eq_stress_band())          # Peak 20 Hz Q 10 +16 dB
    bands.append(BandConfig(FilterType.Peak, 15000.0, 0.1, 16.0, True))
    eg, eo = _mk_eqs(ohs, oracle, bands)
    x = synth.white_noise([4], 6000)[0]
    gl, gr = x[0].copy(), x[1].copy()
    ol, orr = x[0].copy(), x[1].copy()
    eg.process_block(gl, gr); eo.process_block(ol, orr)
    assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32))
    assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32))
    # reset_all_bands_state zeroes s1/s2 only (parametric_eq.rs:181-188)
    eg.reset_all_bands_state(); eo.reset_all_bands_state()
    gl2, gr2 = x[0].copy(), x[1].copy(); ol2, or2 = x[0].copy(), x[1].copy()
    eg.process_block(gl2, gr2); eo.process_block(ol2, or2)
    assert np.array_equal(gl2, gl) and np.array_equal(gl2.view(np.uint32), ol2.view(np.uint32))


def test_eq_more_than_16_bands(ohs, synth, oracle):
    from open_headstage_amd import BandConfig, FilterType
    bands = [BandConfig(FilterType.Peak, 100.0 * (i + 1), 1.0, (-1.0) ** i * 2.0, i % 5 != 0)
             for i in range(23)]
    eg, eo = _mk_eqs(ohs, oracle, bands)
    x = synth.white_noise([6], 2000)[0]
    gl, gr = x[0].copy(), x[1].copy(); ol, orr = x[0].copy(), x[1].copy()
    eg.process_block(gl, gr); eo.process_block(ol, orr)
    assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32))
    assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32))


def test_eq_coefficient_errors_are_the_biquad_crates(ohs):
    """(What the coefficient values ARE is pinned on the CPU, independently of the oracle: tests/test_cpu_coeff_pins.py.)
    The error cases of biquad::Coefficients::from_params, which the reference unwrap()s (parametric_eq.rs:111):"""
    from open_headstage_amd import FilterType, _ffi, biquad_coefficients
    with pytest.raises(ohs.OhsError) as ei:
        biquad_coefficients(FilterType.Peak, 48000.0, 30000.0, 1.0, 0.0)
    assert ei.value.status == _ffi.OHS_ERR_OUTSIDE_NYQUIST
    with pytest.raises(ohs.OhsError) as ei:
        biquad_coefficients(FilterType.Peak, 48000.0, 1000.0, -1.0, 0.0)
    assert ei.value.status == _ffi.OHS_ERR_NEGATIVE_Q


def test_eq_one_ulp_in_one_coefficient_moves_the_output_beyond_1e_6(ohs, synth, oracle):
    """Why the EQ's coefficients must be the reference's BITS: with the bench's 10-band table, one coefficient of one
    band changed by one ulp moves 96 000 output samples by far more than the 1e-6 parity bar (the low-frequency
    sections have poles at 1 - 1e-3 ... 1 - 1e-2: coefficient noise is amplified by 1 / (1 + a1 + a2)).  This is what
    INTEGRATION.md section 4 rests on: a host that needs the reference's output hands ohs_eq_set_band_coeffs the
    `biquad` crate's own coefficients; ohs_eq_update_band (this library's restatement of the crate's libm) is a
    convenience whose parity with the crate is unpinned."""
    bands = synth.eq_table()
    n = 96000
    x = synth.white_noise([11], n)[0]

    def run(perturb):
        eq = ohs.StereoParametricEQ.new(len(bands), 48000.0)
        for i, b in enumerate(bands):
            eq.update_band_coeffs(i, 48000.0, b)
        if perturb is not None:
            band, k = perturb
            c, en = eq.get_band_coeffs(band)
            c = c.copy()
            c[k] = np.nextafter(c[k], np.float32(np.inf), dtype=np.float32)
            eq.set_band_coeffs(band, c, en)
        l, r = x[0].copy(), x[1].copy()
        eq.process_block(l, r)
        return np.stack([l, r]).astype(np.float64)

    ref = run(None)
    assert np.array_equal(run(None), ref)                       # the experiment itself is deterministic
    rms = float(np.sqrt(np.mean(ref ** 2)))
    moved = {}
    for name, perturb in (("band 0 a1", (0, 3)), ("band 0 b0", (0, 0)), ("band 1 a1", (1, 3)), ("band 5 a1", (5, 3))):
        d = run(perturb) - ref
        moved[name] = float(np.sqrt(np.mean(d ** 2))) / rms
    print("relative RMS change per one-ulp coefficient change:", {k: f"{v:.2e}" for k, v in moved.items()})
    assert moved["band 0 a1"] > 1e-5 and moved["band 0 b0"] > 1e-5 and moved["band 1 a1"] > 1e-5, moved
    assert all(v > 1e-6 for v in moved.values()), moved


def test_eq_frequency_response(ohs, synth, oracle):
    eg, eo = _mk_eqs(ohs, oracle, synth.eq_table())
    f = np.geomspace(20.0, 20000.0, 64).astype(np.float32)
    g = eg.calculate_frequency_response(48000.0, f)
    o = eo.calculate_frequency_response(48000.0, f)
    assert np.allclose(g, o, rtol=2e-5, atol=1e-6), np.abs(g / o - 1).max()


# ------------------------------------------------------------------ Plugin::process chain
def test_chain_eq_conv_gain(ohs, synth, oracle):
    """lib.rs:1169-1207: EQ -> convolution -> gain, once per 1024-frame host block."""
    irs = synth.hrir_set(512)
    eg = ohs.ConvolutionEngine.new()
    eo = setup_engines(oracle, eg, irs)
    qg, qo = _mk_eqs(ohs, oracle, synth.eq_table())
    x = synth.white_noise([0], 1024 * 6)[0]
    gain = np.float32(0.5011872)
    outs_g, outs_o = [], []
    for c in range(6):
        gl = x[0, c * 1024:(c + 1) * 1024].copy(); gr = x[1, c * 1024:(c + 1) * 1024].copy()
        ol, orr = gl.copy(), gr.copy()
        ohs.process_chain(eg, qg, gl, gr, eq_enable=True, output_gain=float(gain))
        oracle.chain_process(eo, qo, ol, orr, eq_enable=True, gain=float(gain))
        outs_g.append(np.stack([gl, gr])); outs_o.append(np.stack([ol, orr]))
    assert_parity(np.concatenate(outs_g, 1), np.concatenate(outs_o, 1), "chain")
    # bypass leaves the buffer untouched and freezes DSP state (lib.rs:1169)
    b = x[0, :1024].copy(); b2 = x[1, :1024].copy()
    ohs.process_chain(eg, qg, b, b2, master_bypass=True, eq_enable=True, output_gain=0.1)
    assert np.array_equal(b, x[0, :1024]) and np.array_equal(b2, x[1, :1024])


# ------------------------------------------------------------------ batch mode
def _batch_vs_oracle(ohs, synth, oracle, S, taps, n_blocks, eq_on, calls=2, gain=1.0):
    import torch
    irs = synth.hrir_set(taps)
    bands = synth.eq_table()
    bp = ohs.BatchProcessor(S, num_bands=len(bands))
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(bands):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(eq_on)
    bp.set_gain(gain)
    frames = n_blocks * 512
    x = synth.white_noise(range(S), frames * calls)
    outs = []
    for c in range(calls):
        xt = torch.from_numpy(np.ascontiguousarray(x[:, :, c * frames:(c + 1) * frames])).cuda()
        outs.append(bp.process(xt).cpu().numpy())
    g = np.concatenate(outs, axis=2)
    worst = 0.0
    for s in range(S):
        eo = oracle.ConvolutionEngine()
        for p in range(4):
            eo.set_ir(p, irs[p])
        qo = oracle.StereoParametricEQ(len(bands), synth.FS)
        for i, b in enumerate(bands):
            qo.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), b.enabled)
        l = x[s, 0].copy(); r = x[s, 1].copy()
        oracle.chain_process(eo, qo, l, r, eq_enable=eq_on, gain=gain)
        a, rel = assert_parity(g[s], np.stack([l, r]), f"batch stream {s}")
        worst = max(worst, rel)
    return worst


def test_batch_p1_with_eq(ohs, synth, oracle):
    w = _batch_vs_oracle(ohs, synth, oracle, S=9, taps=512, n_blocks=6, eq_on=True, gain=0.7)
    print("batch P=1 +EQ worst rel RMS", w)


def test_batch_p1_no_eq_inplace(ohs, synth, oracle):
    import torch
    irs = synth.hrir_set(512)
    bp = ohs.BatchProcessor(4, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    x = synth.white_noise(range(4), 2048)
    xt = torch.from_numpy(x.copy()).cuda()
    bp.process(xt, out=xt)     # d_in == d_out
    g = xt.cpu().numpy()
    for s in range(4):
        eo = oracle.ConvolutionEngine()
        for p in range(4):
            eo.set_ir(p, irs[p])
        o = np.stack(eo.process_block(x[s, 0], x[s, 1]))
        assert_parity(g[s], o, f"in-place stream {s}")


def test_batch_p1_output_bits_are_those_of_the_natural_bin_layout(ohs, synth):
    """k_conv_p1 computes in a paired bin layout (bins k and N - k in one lane: no mirror-spectrum reads).  Bin by bin
    its arithmetic is that of the natural layout, operation for operation, so the output BITS must be the ones the
    natural-layout build (-DOHS_P1_NATURAL) produced: tests/golden/p1_output_sha256.txt, written by tools/p1_hash.py
    with that build.  A deliberate change of the kernel's arithmetic needs the file regenerated (and says so here):
    last done when batch objects began to carry the MERGED overlap across call boundaries (lazy state) -- the
    multi-call shapes changed, the single-call ones did not; the experiments build's p1_lazy_state=0 still reproduces the
    older file."""
    import hashlib
    import os
    import torch
    want = {}
    with open(os.path.join(os.path.dirname(__file__), "golden", "p1_output_sha256.txt")) as f:
        for line in f:
            S, blocks, calls, digest = line.split()
            want[(int(S), int(blocks), int(calls))] = digest
    assert len(want) == 6
    dev = torch.device("cuda:0")
    for (S, blocks, calls), digest in want.items():
        bp = ohs.BatchProcessor(S, num_bands=10)
        irs = synth.hrir_set(512)
        for p in range(4):
            bp.set_ir(p, irs[p])
        bp.set_eq_enabled(False)
        bp.set_gain(0.7)
        bp.set_conv_plan(1)         # the block-512 plan's bits (from 512 streams on the library picks the hop-1536 plan by itself)
        h = hashlib.sha256()
        for c in range(calls):
            x = synth.white_noise_torch(10 * S + c, S, blocks * 512, dev)
            y = torch.empty_like(x)
            bp.process(x, out=y)
            torch.cuda.synchronize()
            h.update(y.cpu().numpy().tobytes())
        assert h.hexdigest() == digest, f"k_conv_p1 output bits changed: {S} streams x {blocks} blocks x {calls} calls"


@pytest.mark.parametrize("S,n_blocks", [(512, 64), (1024, 32), (2048, 16), (1000, 40), (4100, 16)])
def test_batch_p1_many_streams_chunk_plans_agree_bit_for_bit(ohs, synth, exp_tuning, S, n_blocks):
    """From 512 streams up a stream has 8, 4 or 2 chunks and k_conv_p1 deals a workgroup's waves out across its streams
    (p1_wave_job: every stream owns waves of all age ranks).  Where a stream is cut does not change a bit of the result,
    so the PRODUCT library's plan must equal a plan of three chunks per stream (consecutive waves; the experiments
    build's p1_target_waves) and one without any cut, over two calls that chain the state."""
    import torch
    dev = torch.device("cuda:0")
    irs = synth.hrir_set(512)
    outs = []
    for target in (None, 3 * S, S):
        if target is not None:
            exp_tuning("p1_target_waves", target)
        bp = ohs.BatchProcessor(S, num_bands=10, library=None if target is None else exp_tuning.lib)
        for p in range(4):
            bp.set_ir(p, irs[p])
        bp.set_eq_enabled(False)
        ys = []
        for c in range(2):
            x = synth.white_noise_torch(7 + c, S, n_blocks * 512, dev)
            y = torch.empty_like(x)
            bp.process(x, out=y)
            torch.cuda.synchronize()
            ys.append(y.cpu())
        outs.append(torch.cat(ys, dim=-1))
        del bp
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert torch.equal(outs[0].view(torch.int32), outs[2].view(torch.int32))


@pytest.mark.parametrize("eq_on", [False, True])
def test_batch_p1_call_boundaries_are_invisible_in_the_output_bits(ohs, synth, eq_on):
    """A batch object carries the merged overlap of its last block from one P = 1 call to the next (the lazy state of
    conv_launch) -- what the chunks of one launch hand each other -- and the EQ's state is exact anyway: cutting a
    signal into calls of 40 + 1 + 39 blocks gives the bits of one call of 80, and a per-path set_ir afterwards still
    finds the per-path overlaps (materialised from the saved last block)."""
    import torch
    dev = torch.device("cuda:0")
    S, n = 7, 80
    irs = synth.hrir_set(512)
    x = synth.white_noise_torch(21, S, n * 512, dev)

    def make():
        bp = ohs.BatchProcessor(S, num_bands=10)
        for p in range(4):
            bp.set_ir(p, irs[p])
        for i, b in enumerate(synth.eq_table()):
            bp.update_band_coeffs(i, synth.FS, b)
        bp.set_eq_enabled(eq_on)
        return bp

    one = make()
    y1 = torch.empty_like(x)
    one.process(x, out=y1)
    parts = make()
    ys = []
    for a, b in ((0, 40), (40, 41), (41, 80)):
        xi = x[:, :, a * 512:b * 512].contiguous()
        yi = torch.empty_like(xi)
        parts.process(xi, out=yi)
        ys.append(yi)
    torch.cuda.synchronize()
    y2 = torch.cat(ys, dim=-1)
    assert torch.equal(y1.view(torch.int32), y2.view(torch.int32))
    # the state at rest is still the reference's: replace one path mid-stream on both objects, same continuation
    for bp in (one, parts):
        bp.set_ir(2, irs[2][:100])
    z = synth.white_noise_torch(22, S, 10 * 512, dev)
    z1, z2 = torch.empty_like(z), torch.empty_like(z)
    one.process(z, out=z1)
    parts.process(z, out=z2)
    torch.cuda.synchronize()
    assert torch.isfinite(z1).all()
    assert torch.equal(z1.view(torch.int32), z2.view(torch.int32))


@pytest.mark.parametrize("lazy", [True, False])
def test_batch_p1_midstream_set_ir_vs_oracle(ohs, synth, oracle, exp_tuning, lazy):
    """The P = 1 batch path across per-path set_ir calls in mid-stream, against the ORACLE (not against itself): the lazy
    state (merged overlap + the last input block) is turned back into per-path overlaps by conv_materialise_state with
    the OLD spectra before they are replaced (convolution.rs:135-138 resets only that path's overlap).  A shorter
    response, a mute and a full-length one; product library (lazy state) and the experiments build with
    p1_lazy_state = 0 (per-path state after every launch)."""
    import torch
    S, n_blocks = 5, 9
    irs = synth.hrir_set(512)
    lib = None
    if not lazy:
        exp_tuning("p1_lazy_state", 0)
        exp_tuning.touched.discard("p1_lazy_state")      # (read when a batch is created; put back by hand below)
        lib = exp_tuning.lib
    try:
        bp = ohs.BatchProcessor(S, num_bands=10, library=lib)
    finally:
        if not lazy:
            exp_tuning._ffi.set_tuning("p1_lazy_state", 1)
    engs = [oracle.ConvolutionEngine() for _ in range(S)]
    for p in range(4):
        bp.set_ir(p, irs[p])
        for e in engs:
            e.set_ir(p, irs[p])
    edits = {1: (2, irs[2][:100]), 2: (1, np.zeros(0, np.float32)), 3: (0, synth.hrir_set(512)[3] * np.float32(0.7)),
             4: (1, irs[1])}
    got, ref = [], []
    for call in range(6):
        if call in edits:
            path, h = edits[call]
            bp.set_ir(path, h)
            for e in engs:
                e.set_ir(path, h)
        x = synth.white_noise(range(30 * call, 30 * call + S), n_blocks * 512)
        got.append(bp.process(torch.from_numpy(x).cuda()).cpu().numpy())
        ref.append(np.stack([np.stack(engs[s].process_block(x[s, 0], x[s, 1])) for s in range(S)]))
    for call in range(6):       # call by call: an error confined to the block after a reset cannot hide in an average
        for s in range(S):
            assert_parity(got[call][s], ref[call][s], f"lazy={lazy}, call {call}, stream {s}")
            assert_parity(got[call][s][:, :512], ref[call][s][:, :512], f"lazy={lazy}, call {call}, stream {s}, first block",
                          tol=2e-6)


def test_batch_xcd_partition_of_the_launches_changes_no_bit(ohs, synth, exp_tuning):
    """The XCD partition experiment (kernels.h: xcd_grid -- workgroups on XCDs outside the launch's set return at once, the
    others renumber themselves) must leave results untouched: EQ on XCDs [0, 2), P = 1 convolution on [2, 5) in the
    experiments build, against the product library's unpartitioned launches."""
    import torch
    dev = torch.device("cuda:0")
    S, n_blocks = 37, 70
    irs = synth.hrir_set(512)
    outs = []
    for part in (False, True):
        if part:
            exp_tuning("p1_xcd", "2,3")
            exp_tuning("eq_xcd", "0,2")
        bp = ohs.BatchProcessor(S, num_bands=10, library=exp_tuning.lib if part else None)
        for p in range(4):
            bp.set_ir(p, irs[p])
        for i, b in enumerate(synth.eq_table()):
            bp.update_band_coeffs(i, synth.FS, b)
        bp.set_eq_enabled(True)
        ys = []
        for c in range(2):
            x = synth.white_noise_torch(3 + c, S, n_blocks * 512, dev)
            y = torch.empty_like(x)
            bp.process(x, out=y)
            torch.cuda.synchronize()
            ys.append(y.cpu())
        outs.append(torch.cat(ys, dim=-1))
        del bp
    assert torch.isfinite(outs[0]).all() and float(outs[0].abs().max()) > 0
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))


def test_batch_p1_random_shapes_default_plan_equals_one_chunk_per_stream(ohs, synth, exp_tuning):
    """Seeded sweep over stream counts, block counts and call sequences: the launch plan conv_p1_chunks picks (chunk count
    by rounds / cost rule, waves dealt out across streams, boundary tails by the chunks' own waves or by the pre-pass,
    lazy state across calls) against the plainest plan there is -- one wave per stream, no cut -- bit for bit, in place."""
    import torch
    rng = np.random.default_rng(20261004)
    dev = torch.device("cuda:0")
    irs = synth.hrir_set(512)
    shapes = [(1, 1), (2, 3), (16, 16), (17, 33), (255, 20), (256, 16), (257, 70), (300, 5)]
    shapes += [(int(rng.integers(1, 700)), int(rng.integers(1, 130))) for _ in range(14)]
    for S, n in shapes:
        calls = [n, max(1, n // 3), n]
        outs = []
        for plain in (False, True):
            if plain:
                exp_tuning("p1_target_waves", S)
            bp = ohs.BatchProcessor(S, num_bands=10, library=exp_tuning.lib if plain else None)
            for p in range(4):
                bp.set_ir(p, irs[p])
            bp.set_eq_enabled(False)
            bp.set_gain(0.9)
            ys = []
            for c, nb in enumerate(calls):
                x = synth.white_noise_torch(100 + c, S, nb * 512, dev)
                bp.process(x, out=x)            # in place
                ys.append(x.cpu())
            torch.cuda.synchronize()
            outs.append(torch.cat(ys, dim=-1))
            del bp
        assert torch.isfinite(outs[0]).all(), (S, n)
        assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), (S, n)


def test_batch_long_ir_p32(ohs, synth, oracle):
    w = _batch_vs_oracle(ohs, synth, oracle, S=3, taps=16384, n_blocks=20, eq_on=False, calls=2)
    print("batch P=32 worst rel RMS", w)
    # in place (EQ on => the convolution runs in place on the EQ output), odd segment sizes
    w = _batch_vs_oracle(ohs, synth, oracle, S=2, taps=6000, n_blocks=37, eq_on=True, calls=2, gain=0.9)
    print("batch P=12 +EQ worst rel RMS", w)


def test_batch_mixed_partitions_and_midstream_set_ir(ohs, synth, oracle):
    """Different partition counts per path, a per-path set_ir between calls (transitional gating ->
    general kernel, then back to the time-parallel kernels), segments longer than one tile."""
    import torch
    S = 2
    irs = [synth.hrir_set(3000)[0], synth.hrir_set(700)[1], synth.hrir_set(5000)[2], synth.hrir_set(512)[3]]
    bp = ohs.BatchProcessor(S, num_bands=10)
    engs = [oracle.ConvolutionEngine() for _ in range(S)]
    for p in range(4):
        bp.set_ir(p, irs[p])
        for e in engs:
            e.set_ir(p, irs[p])
    x = synth.white_noise(range(S), 512 * 75)
    plan = [(0, 23), (23, 30), (30, 37), (37, 75)]     # block ranges per call
    outs = []
    for k, (b0, b1) in enumerate(plan):
        if k == 1:      # reset one path mid-stream with a longer IR
            new = synth.hrir_set(4100)[1]
            bp.set_ir(ohs.ConvolutionPath.Lsr, new)
            for e in engs:
                e.set_ir(oracle.LSR, new)
        xt = torch.from_numpy(np.ascontiguousarray(x[:, :, b0 * 512:b1 * 512])).cuda()
        outs.append(bp.process(xt).cpu().numpy())
    g = np.concatenate(outs, axis=2)
    for s_ in range(S):
        o = np.stack(engs[s_].process_block(x[s_, 0], x[s_, 1])) if False else None
    # oracle: same call pattern (set_ir at the same point)
    engs = [oracle.ConvolutionEngine() for _ in range(S)]
    for p in range(4):
        for e in engs:
            e.set_ir(p, irs[p])
    for s_ in range(S):
        ref = []
        for k, (b0, b1) in enumerate(plan):
            if k == 1:
                engs[s_].set_ir(oracle.LSR, synth.hrir_set(4100)[1])
            ref.append(np.stack(engs[s_].process_block(x[s_, 0, b0 * 512:b1 * 512], x[s_, 1, b0 * 512:b1 * 512])))
        assert_parity(g[s_], np.concatenate(ref, axis=1), f"mixed P stream {s_}")


def test_batch_reset_and_linearity(ohs, synth, oracle):
    """Size-independent properties: reset => same output again; linearity in the input."""
    import torch
    irs = synth.hrir_set(512)
    bp = ohs.BatchProcessor(2, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    x = torch.from_numpy(synth.white_noise(range(2), 4096)).cuda()
    y1 = bp.process(x).clone()
    bp.reset()
    y2 = bp.process(x).clone()
    assert torch.equal(y1, y2)
    bp.reset()
    y3 = bp.process(0.5 * x)
    assert torch.allclose(y3, 0.5 * y1, rtol=0, atol=1e-6)


def test_synth_torch_generator_matches_numpy(synth):
    import torch
    a = synth.white_noise(range(5, 8), 1000, offset=17)
    b = synth.white_noise_torch(5, 3, 1000, torch.device("cuda:0"), offset=17).cpu().numpy()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ------------------------------------------------------------------ edge cases of the fast paths
@pytest.mark.parametrize("taps,n_blocks", [(512, 1), (512, 2), (512, 3), (512, 5), (512, 67),
                                           (2000, 1), (2000, 3), (2000, 4), (2000, 9), (2000, 17), (2000, 32),
                                           (2000, 33), (2000, 50)])
def test_batch_small_block_counts_every_kernel_path(ohs, synth, oracle, taps, n_blocks):
    """P=1: chunks <= n_blocks; P>1: general kernel below 4 blocks, time-parallel kernels from 4 (time tiles of 16 or
    32 blocks, partially filled: 9 -> 16, 17 and 32 -> 32, 33 -> 3 x 16, 50 -> 2 x 32), three consecutive calls chaining
    the state."""
    _batch_vs_oracle(ohs, synth, oracle, S=3, taps=taps, n_blocks=n_blocks, eq_on=False, calls=3)


@pytest.mark.parametrize("nb", [1, 2, 4, 5, 8, 9, 11, 12, 13, 15, 16, 17, 28, 32])
def test_eq_band_counts_bit_exact(ohs, synth, oracle, nb):
    """<= 12 bands: ring kernel (12 = the last band's output lands in the I/O port bank); 13..16:
    conveyor kernel (16 = the full DPP row, no output shift); 17 / 28 / 32 = two passes (16 + ring or
    16 + conveyor)."""
    from open_headstage_amd import BandConfig, FilterType
    bands = [BandConfig(FilterType.Peak, 60.0 * (i + 1.5), 0.9 + 0.05 * i, (-1.0) ** i * (1.0 + 0.2 * i), True)
             for i in range(nb)]
    eg, eo = _mk_eqs(ohs, oracle, bands)
    x = synth.white_noise([8], 5000)[0]
    gl, gr = x[0].copy(), x[1].copy(); ol, orr = x[0].copy(), x[1].copy()
    eg.process_block(gl, gr); eo.process_block(ol, orr)
    assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32))
    assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32))


def test_batch_eq_long_stream_bit_exact_then_conv(ohs, synth, oracle):
    """The asm steady-state groups + hand-counted vmcnt of k_eq_pass over a long stream (several
    hundred groups), chained across two calls, with the EQ||convolution overlap active."""
    w = _batch_vs_oracle(ohs, synth, oracle, S=5, taps=512, n_blocks=130, eq_on=True, calls=2, gain=1.0)
    print("long stream worst rel RMS", w)


def test_batch_strided_buffers(ohs, synth, oracle):
    """process_ptr with strides larger than the processed region (a window into a bigger buffer)."""
    import torch
    S, frames, pad = 3, 2048, 384
    irs = synth.hrir_set(512)
    bp = ohs.BatchProcessor(S, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    x = synth.white_noise(range(S), frames)
    big = torch.full((S, 2, frames + pad), 7.0, device="cuda")
    big[:, :, :frames] = torch.from_numpy(x).cuda()
    out = torch.full_like(big, -3.0)
    bp.process_ptr(big.data_ptr(), out.data_ptr(), frames // 512, 2 * (frames + pad), frames + pad,
                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert np.all(o[:, :, frames:] == -3.0), "wrote outside the processed region"
    for s_ in range(S):
        eo = oracle.ConvolutionEngine()
        for p in range(4):
            eo.set_ir(p, irs[p])
        qo = oracle.StereoParametricEQ(10, synth.FS)
        for i, b in enumerate(synth.eq_table()):
            qo.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
        l, r = x[s_, 0].copy(), x[s_, 1].copy()
        oracle.chain_process(eo, qo, l, r, eq_enable=True, gain=1.0)
        assert_parity(o[s_, :, :frames], np.stack([l, r]), f"strided stream {s_}")


def test_batch_strides_beyond_32_bit_lane_offsets(ohs, synth, oracle):
    """The ring-form EQ addresses a lane's samples by a 32-bit byte offset from its wave's base (the offset spans the
    wave's two streams): a stream stride of 2^30 floats (4 GiB) does not fit, the launcher must take the conveyor form
    (64-bit addresses) -- same bits -- and the convolution its own 64-bit stream offsets."""
    import torch
    S, frames = 2, 1536
    ss, cs = 1 << 30, 4096
    irs = synth.hrir_set(512)
    bp = ohs.BatchProcessor(S, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    x = synth.white_noise(range(S), frames)
    n = (S - 1) * ss + cs + frames
    big = torch.zeros(n, device="cuda")
    out = torch.zeros(n, device="cuda")
    for s_ in range(S):
        for c in range(2):
            big[s_ * ss + c * cs: s_ * ss + c * cs + frames] = torch.from_numpy(x[s_, c]).cuda()
    bp.process_ptr(big.data_ptr(), out.data_ptr(), frames // 512, ss, cs, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for s_ in range(S):
        eo = oracle.ConvolutionEngine()
        for p in range(4):
            eo.set_ir(p, irs[p])
        qo = oracle.StereoParametricEQ(10, synth.FS)
        for i, b in enumerate(synth.eq_table()):
            qo.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
        l, r = x[s_, 0].copy(), x[s_, 1].copy()
        oracle.chain_process(eo, qo, l, r, eq_enable=True, gain=1.0)
        got = np.stack([out[s_ * ss + c * cs: s_ * ss + c * cs + frames].cpu().numpy() for c in range(2)])
        assert_parity(got, np.stack([l, r]), f"far-apart stream {s_}")
    del big, out
    torch.cuda.empty_cache()


def test_autoeq_profile_drives_the_eq_bit_exact(ohs, synth, oracle):
    """f3 end to end: CSV -> bands -> coefficients -> EQ kernel, bit-exact against the oracle fed the
    same parsed rows."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "autoeq_sample.csv")
    bands = ohs.parse_autoeq_csv(path)
    eg = ohs.StereoParametricEQ.new(10, 48000.0)
    ohs.apply_bands(eg, bands, 48000.0)
    eo = oracle.StereoParametricEQ(10, 48000.0)
    for i, b in enumerate(bands):
        eo.update_band_coeffs(i, 48000.0, int(b.filter_type), b.frequency, b.q, b.gain, b.enabled)
    x = synth.white_noise([12], 4096)[0]
    gl, gr = x[0].copy(), x[1].copy(); ol, orr = x[0].copy(), x[1].copy()
    eg.process_block(gl, gr); eo.process_block(ol, orr)
    assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32))
    assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32))


def test_real_hrirs_cipic_subject_003(ohs, synth, oracle):
    """f1 end to end with measured HRIRs (CIPIC subject 003, +/-30 deg, resampled to 48 kHz; fixture
    made by tools/make_sofa_fixture.py from the reference's data asset): engine and batch vs oracle / f64."""
    import os
    import torch
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cipic003_pm30.npz"))
    irs = list(g["irs_48000"])
    eg = ohs.ConvolutionEngine.new()
    eo = setup_engines(oracle, eg, irs)
    x = synth.white_noise([21], 8192)[0]
    outs_g = [np.stack(eg.process_block(x[0, c * 1024:(c + 1) * 1024], x[1, c * 1024:(c + 1) * 1024])) for c in range(8)]
    o = np.stack(eo.process_block(x[0], x[1]))
    gg = np.concatenate(outs_g, axis=1)
    f = np.stack(oracle.binaural_f64(x[0], x[1], irs))
    assert_parity(gg, o, "CIPIC engine vs oracle")
    assert_parity(gg, f, "CIPIC engine vs f64")
    bp = ohs.BatchProcessor(1, num_bands=10)
    from open_headstage_amd import sofa
    sofa.load_into(bp, irs)
    y = bp.process(torch.from_numpy(x[None].copy()).cuda()).cpu().numpy()[0]
    assert_parity(y, f, "CIPIC batch vs f64")


@pytest.mark.parametrize("nb", [1, 5, 10, 12, 13, 16])
def test_eq_sample_routing_identity_filters_on_a_ramp(ohs, nb):
    """Routing check of the systolic EQ kernel (conveyor refills / drains at every chunk and group
    boundary): identity bands (b0 = 1) must return the ramp exactly, for sizes around every boundary."""
    eg = ohs.StereoParametricEQ.new(nb, 48000.0)
    for i in range(nb):
        eg.set_band_coeffs(i, [1.0, 0.0, 0.0, 0.0, 0.0], True)
    for n in (127, 128, 129, 255, 256, 257, 300, 383, 384, 385, 400, 511, 512, 640, 1000, 2049):
        l = np.arange(n, dtype=np.float32)
        r = (100000.0 - np.arange(n)).astype(np.float32)
        el, er = l.copy(), r.copy()
        eg.process_block(l, r)
        assert np.array_equal(l, el) and np.array_equal(r, er), (nb, n, np.nonzero(l != el)[0][:8])


def test_eq_ring_documented_edge_cases_signed_zero_and_non_finite_input(ohs, synth, oracle):
    """DESIGN.md 4.5 / LABNOTES.md: the ring kernel's pass-on lanes run `1*x + 0` in the asm groups.  (a) A -0.0
    sample may come out as +0.0: equal as numbers, every other sample bit-exact.  (b) A non-finite
    input sample: the reference (and the oracle) is non-finite from that sample on; this kernel
    may additionally lose up to 16 samples BEFORE it -- everything earlier is bit-exact."""
    eg, eo = _mk_eqs(ohs, oracle, synth.eq_table())
    n = 4096
    x = synth.white_noise([12], n)[0]
    x[:, 1000:1100] = -0.0
    gl, gr = x[0].copy(), x[1].copy(); ol, orr = x[0].copy(), x[1].copy()
    eg.process_block(gl, gr); eo.process_block(ol, orr)
    assert np.array_equal(gl, ol) and np.array_equal(gr, orr)           # numeric equality (-0 == +0)
    nz = ol != 0
    assert np.array_equal(gl[nz].view(np.uint32), ol[nz].view(np.uint32))

    eg, eo = _mk_eqs(ohs, oracle, synth.eq_table())
    x = synth.white_noise([13], n)[0]
    k = 2500
    x[0, k] = np.nan
    gl, gr = x[0].copy(), x[1].copy(); ol, orr = x[0].copy(), x[1].copy()
    eg.process_block(gl, gr); eo.process_block(ol, orr)
    assert np.array_equal(gr.view(np.uint32), orr.view(np.uint32))      # the other channel is untouched
    assert np.array_equal(gl[:k - 16].view(np.uint32), ol[:k - 16].view(np.uint32))
    assert np.isnan(ol[k:]).all() and np.isnan(gl[k:]).all()


@pytest.mark.parametrize("nb_on", [1, 7, 10, 12])
def test_eq_exact_specials_mode_is_bit_exact_on_signed_zeros_and_around_non_finite_input(ohs, synth, oracle, nb_on):
    """ohs_eq_set_exact_specials / ohs_batch_set_eq_exact_specials: the conveyor kernel moves samples by DPP
    moves only, so -0.0 keeps its sign bit and a non-finite input sample changes nothing before itself --
    every bit equals the oracle's, also the NaN pattern after the bad sample's position."""
    import torch
    bands = synth.eq_table()[:nb_on]
    n = 4096
    for case in ("negzero", "nan", "inf"):
        eg, eo = _mk_eqs(ohs, oracle, bands)
        eg.set_exact_specials(True)
        x = synth.white_noise([40 + nb_on], n)[0]
        if case == "negzero":
            x[:, 700:900] = -0.0
            x[1, :64] = -0.0
        elif case == "nan":
            x[0, 2500] = np.nan
        else:
            x[1, 1234] = np.inf
        gl, gr = x[0].copy(), x[1].copy(); ol, orr = x[0].copy(), x[1].copy()
        eg.process_block(gl, gr); eo.process_block(ol, orr)
        for g, o, ch in ((gl, ol, "L"), (gr, orr, "R")):
            fin = np.isfinite(o)
            assert np.array_equal(np.isfinite(g), fin), (case, ch)
            assert np.array_equal(g[fin].view(np.uint32), o[fin].view(np.uint32)), (case, ch)
    # batch entry point: on ordinary samples the two EQ kernels agree bit for bit, so the whole chain does
    S = 3
    irs = synth.hrir_set(512)
    outs = []
    for exact in (False, True):
        bp = ohs.BatchProcessor(S, num_bands=len(bands))
        for p in range(4):
            bp.set_ir(p, irs[p])
        for i, b in enumerate(bands):
            bp.update_band_coeffs(i, synth.FS, b)
        bp.set_eq_enabled(True)
        bp.set_eq_exact_specials(exact)
        x = synth.white_noise(range(S), 80 * 512)
        outs.append(bp.process(torch.from_numpy(x).cuda()).cpu().numpy())
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_batch_deferred_join_is_bit_identical_and_mixes_with_everything(ohs, synth, oracle):
    """ohs_batch_process_deferred pipelines back-to-back calls (the next call's EQ starts while the last
    convolution chunk of the previous one finishes).  Same bits as ohs_batch_process, whatever follows a
    deferred call: the same geometry again, another buffer, a short call, a call without EQ, set_ir."""
    import torch
    S, taps, n_blocks = 6, 512, 96
    frames = n_blocks * 512
    irs = synth.hrir_set(taps)
    bands = synth.eq_table()

    def make():
        bp = ohs.BatchProcessor(S, num_bands=len(bands))
        for p in range(4):
            bp.set_ir(p, irs[p])
        for i, b in enumerate(bands):
            bp.update_band_coeffs(i, synth.FS, b)
        bp.set_eq_enabled(True)
        return bp

    x = torch.from_numpy(synth.white_noise(range(500, 500 + S), frames * 6)).cuda()
    seg = [x[:, :, k * frames:(k + 1) * frames].contiguous() for k in range(6)]
    short = seg[4][:, :, :20 * 512].contiguous()

    def run(bp, deferred):
        outs = []
        y = torch.empty_like(seg[0])
        for k in range(3):                       # same buffers, back to back: the pipelined case
            bp.process(seg[k], out=y, deferred=deferred)
            if deferred:
                bp.join()                        # consume on the stream before the buffer is reused
            outs.append(y.clone())
        ys = [torch.empty_like(seg[0]) for _ in range(2)]
        bp.process(seg[3], out=ys[0], deferred=deferred)       # deferred, then another output buffer
        bp.process(seg[3], out=ys[1], deferred=deferred)
        bp.process(short, out=None if not deferred else None, deferred=deferred)    # a short (non-overlapped) call
        bp.set_eq_enabled(False)
        y5 = bp.process(seg[5], deferred=deferred)             # no EQ: convolution on the caller's stream
        bp.set_eq_enabled(True)
        if deferred:
            bp.join()
        torch.cuda.synchronize()
        outs += [ys[0].clone(), ys[1].clone(), y5.clone()]
        irs2 = synth.hrir_set(300)
        bp.set_ir(2, irs2[2])                                   # synchronises by itself
        ylast = bp.process(seg[0], deferred=deferred)
        if deferred:
            bp.sync()                                           # host-side wait, includes the pending call
        torch.cuda.synchronize()
        outs.append(ylast.clone())
        return outs

    a = run(make(), False)
    b = run(make(), True)
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), f"deferred differs from strict at output {i}"

    # the pure pipeline (no join between calls, same buffers) against the oracle for the last call
    bp = make()
    y = torch.empty_like(seg[0])
    for k in range(3):
        bp.process(seg[k], out=y, deferred=True)
    bp.join()
    torch.cuda.synchronize()
    assert torch.equal(y, a[2])


def test_batch_eq_in_place_with_a_partial_last_wave(ohs, synth, oracle):
    """EQ + convolution in place (d_in == d_out) when the chain count is not a multiple of 4: the idle rows of
    the last EQ wave must shadow a row of their OWN wave (lock-step).  Shadowing a chain of another wave --
    here stream 0, handled by a wave on another CU -- read samples that wave had already overwritten as soon
    as the two drifted apart under the overlapped convolution (found by tests/test_gpu_fuzz.py)."""
    import torch
    S = 255                                   # 510 chains: the last wave has two idle rows
    bands = synth.eq_table()
    irs = synth.hrir_set(512)
    n = 512 * 128
    x0 = synth.white_noise([900], n)[0]
    eo = oracle.ConvolutionEngine()
    for p in range(4):
        eo.set_ir(p, irs[p])
    qo = oracle.StereoParametricEQ(len(bands), synth.FS)
    for i, b in enumerate(bands):
        qo.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
    l, r = x0[0].copy(), x0[1].copy()
    oracle.chain_process(eo, qo, l, r, eq_enable=True, gain=1.0)
    ref0 = np.stack([l, r])
    bp = ohs.BatchProcessor(S, num_bands=len(bands))
    for i, b in enumerate(bands):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    for p in range(4):
        bp.set_ir(p, irs[p])
    base = synth.white_noise_torch(901, 3, n, torch.device("cuda:0"))
    for rep in range(3):
        bp.reset()
        xt = base[torch.arange(S, device="cuda") % 3].contiguous()
        xt[0] = torch.from_numpy(x0).cuda()
        y = bp.process(xt, out=xt)
        assert_parity(y[0].cpu().numpy(), ref0, f"in-place chain, stream 0, repetition {rep}")
        assert torch.equal(y[4], y[7]) and torch.equal(y[5], y[254])       # 4 = 7 = 1 mod 3, 5 = 254 = 2 mod 3


def test_handles_on_concurrent_host_threads(ohs, synth, oracle):
    """One handle = one thread at a time, but different handles run on different host threads at once
    (ctypes releases the GIL inside the library): four engine + EQ pairs, each driven by its own thread."""
    import threading
    irs = synth.hrir_set(512)
    bands = synth.eq_table()
    n_threads, calls, n = 4, 60, 1024
    results = [None] * n_threads
    errors = []

    def worker(tid):
        try:
            eg = ohs.ConvolutionEngine.new()
            for p in range(4):
                eg.set_ir(p, irs[p])
            qg = ohs.StereoParametricEQ.new(len(bands), synth.FS)
            for i, b in enumerate(bands):
                qg.update_band_coeffs(i, synth.FS, b)
            x = synth.white_noise([700 + tid], calls * n)[0]
            out = np.empty_like(x)
            for c in range(calls):
                l, r = x[0, c * n:(c + 1) * n].copy(), x[1, c * n:(c + 1) * n].copy()
                ohs.process_chain(eg, qg, l, r, eq_enable=True, output_gain=0.9)
                out[0, c * n:(c + 1) * n], out[1, c * n:(c + 1) * n] = l, r
            results[tid] = out
        except Exception as e:       # noqa: BLE001 -- reported below
            errors.append((tid, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    for tid in range(n_threads):
        eo = oracle.ConvolutionEngine()
        for p in range(4):
            eo.set_ir(p, irs[p])
        qo = oracle.StereoParametricEQ(len(bands), synth.FS)
        for i, b in enumerate(bands):
            qo.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
        x = synth.white_noise([700 + tid], calls * n)[0]
        ref = np.empty_like(x)
        for c in range(calls):
            l, r = x[0, c * n:(c + 1) * n].copy(), x[1, c * n:(c + 1) * n].copy()
            oracle.chain_process(eo, qo, l, r, eq_enable=True, gain=0.9)
            ref[0, c * n:(c + 1) * n], ref[1, c * n:(c + 1) * n] = l, r
        assert_parity(results[tid], ref, f"thread {tid}")
