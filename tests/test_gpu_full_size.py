"""BASELINE.json's full-size configurations, checked through size-independent properties
(the oracle would take minutes at these sizes): stream independence (bit-identical outputs for
identical inputs), linearity, impulse responses, determinism, plus an oracle check of a prefix."""
import numpy as np
import pytest

from tests.util import assert_parity

pytestmark = pytest.mark.gpu


def test_config3_full_size_256_streams_eq_conv():
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    from oracle import ohs_oracle as orc

    S, frames, taps = 256, 480256, 512
    irs = synth.hrir_set(taps)
    bands = synth.eq_table()
    bp = ohs.BatchProcessor(S, num_bands=len(bands))
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(bands):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    dev = torch.device("cuda:0")
    base = synth.white_noise_torch(0, 4, frames, dev)                 # streams 0..3: distinct noise
    x = base.repeat(S // 4, 1, 1).contiguous()                        # stream s = stream s % 4
    x[4:8] *= 0.5                                                      # streams 4..7: half amplitude
    y = bp.process(x)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    # independence: identical inputs => bit-identical outputs, wherever the stream sits in the batch
    for s in range(8, S):
        assert torch.equal(y[s], y[s % 4]), f"stream {s} differs from stream {s % 4}"
    # linearity (EQ and convolution are linear; f32 rounding only)
    for s in range(4):
        d = (y[4 + s] - 0.5 * y[s]).double()
        rel = float(torch.sqrt((d * d).mean()) / torch.sqrt((y[s].double() ** 2).mean()))
        assert rel <= 1e-6, f"linearity stream {s}: {rel:.3e}"
    # |y| stays bounded: L1-normalised HRIRs, EQ gain <= +4 dB shelving/peaks on white noise
    assert float(y.abs().max()) < 8.0
    # prefix of stream 0 against the oracle
    n = 20 * 1024
    eng = orc.ConvolutionEngine()
    for p in range(4):
        eng.set_ir(p, irs[p])
    eq = orc.StereoParametricEQ(len(bands), synth.FS)
    for i, b in enumerate(bands):
        eq.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
    xs = base[0, :, :n].cpu().numpy()
    l, r = xs[0].copy(), xs[1].copy()
    orc.chain_process(eng, eq, l, r, eq_enable=True, gain=1.0)
    assert_parity(y[0, :, :n].cpu().numpy(), np.stack([l, r]), "config 3 stream 0 prefix")
    # determinism: same state + same input => same bits
    bp.reset()
    y2 = bp.process(x)
    assert torch.equal(y, y2)


def test_config4_full_size_impulse_responses():
    """64 streams, 4 x 16384 taps (P = 32): an impulse per stream at a different position must come
    back as the HRIRs themselves (out_l = lsl + rsl, out_r = lsr + rsr), across two calls."""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth

    S, taps, n_blocks = 64, 16384, 96
    frames = n_blocks * 512
    irs = synth.hrir_set(taps)
    bp = ohs.BatchProcessor(S, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    x = torch.zeros((S, 2, frames), device="cuda")
    pos_l = [17 * s + 3 for s in range(S)]
    pos_r = [900 + 29 * s for s in range(S)]
    for s in range(S):
        x[s, 0, pos_l[s]] = 1.0
        x[s, 1, pos_r[s]] = -0.5
    half = frames // 2
    y = torch.cat([bp.process(x[:, :, :half].contiguous()), bp.process(x[:, :, half:].contiguous())], dim=2)
    y = y.cpu().numpy()
    for s in (0, 1, 31, 63):
        exp = np.zeros((2, frames))
        exp[0, pos_l[s]:pos_l[s] + taps] += irs[0][: frames - pos_l[s]]
        exp[1, pos_l[s]:pos_l[s] + taps] += irs[1][: frames - pos_l[s]]
        exp[0, pos_r[s]:pos_r[s] + taps] += -0.5 * irs[2].astype(np.float64)[: frames - pos_r[s]]
        exp[1, pos_r[s]:pos_r[s] + taps] += -0.5 * irs[3].astype(np.float64)[: frames - pos_r[s]]
        assert_parity(y[s], exp, f"config 4 impulse stream {s}")


def test_config5_per_node_stream_count_on_one_gpu():
    """BASELINE config 5 shards 2048 streams over 8 GPUs; here all 2048 sit on one device (4096 EQ chains =
    one wave per SIMD, 4 block ranges per stream in the convolution): streams 0, 1023 and 2047 against the
    oracle, every other stream against the stream it repeats, across two calls."""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    from oracle import ohs_oracle as orc

    S, n_blocks, taps = 2048, 24, 512
    frames = n_blocks * 512
    irs = synth.hrir_set(taps)
    bands = synth.eq_table()
    bp = ohs.BatchProcessor(S, num_bands=len(bands))
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(bands):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    bp.set_gain(0.7)
    dev = torch.device("cuda:0")
    base = synth.white_noise_torch(40, 3, 2 * frames, dev)            # three distinct streams
    x = base[torch.arange(S, device=dev) % 3].contiguous()           # stream s = base[s % 3]
    ys = [bp.process(x[:, :, c * frames:(c + 1) * frames].contiguous()) for c in range(2)]
    y = torch.cat(ys, dim=2)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    for s in range(3, S):
        assert torch.equal(y[s], y[s % 3]), f"stream {s} differs from stream {s % 3}"
    for s in (0, 1023, 2047):
        eng = orc.ConvolutionEngine()
        for p in range(4):
            eng.set_ir(p, irs[p])
        eq = orc.StereoParametricEQ(len(bands), synth.FS)
        for i, b in enumerate(bands):
            eq.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
        xs = base[s % 3].cpu().numpy()
        l, r = xs[0].copy(), xs[1].copy()
        orc.chain_process(eng, eq, l, r, eq_enable=True, gain=0.7)
        assert_parity(y[s].cpu().numpy(), np.stack([l, r]), f"config 5 stream {s}")


def test_config5_stream_count_with_calls_long_enough_for_the_hop_plan():
    """2048 streams, 96 blocks per call: the library serves the convolution with its hop-1536 plan here (from 512
    streams and 48 blocks on) -- three hop ranges per stream, two rounds of waves.  Streams 0, 1023, 2047 against the
    oracle across two calls, every other stream against the stream it repeats."""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    from oracle import ohs_oracle as orc

    S, n_blocks, taps = 2048, 96, 512
    frames = n_blocks * 512
    irs = synth.hrir_set(taps)
    bands = synth.eq_table()
    bp = ohs.BatchProcessor(S, num_bands=len(bands))
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(bands):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    bp.set_gain(0.7)
    dev = torch.device("cuda:0")
    base = synth.white_noise_torch(77, 3, 2 * frames, dev)
    x = base[torch.arange(S, device=dev) % 3].contiguous()
    ys = [bp.process(x[:, :, c * frames:(c + 1) * frames].contiguous()) for c in range(2)]
    y = torch.cat(ys, dim=2)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    for s in range(3, S):
        assert torch.equal(y[s], y[s % 3]), f"stream {s} differs from stream {s % 3}"
    for s in (0, 1023, 2047):
        eng = orc.ConvolutionEngine()
        for p in range(4):
            eng.set_ir(p, irs[p])
        eq = orc.StereoParametricEQ(len(bands), synth.FS)
        for i, b in enumerate(bands):
            eq.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
        xs = base[s % 3].cpu().numpy()
        l, r = xs[0].copy(), xs[1].copy()
        orc.chain_process(eng, eq, l, r, eq_enable=True, gain=0.7)
        assert_parity(y[s].cpu().numpy(), np.stack([l, r]), f"config 5, hop plan, stream {s}")
    # ... and it is the hop plan that ran: the bits of plan 2, not of plan 1
    outs = {}
    for plan in (1, 2):
        b2 = ohs.BatchProcessor(S, num_bands=len(bands))
        for p in range(4):
            b2.set_ir(p, irs[p])
        for i, b in enumerate(bands):
            b2.update_band_coeffs(i, synth.FS, b)
        b2.set_eq_enabled(True)
        b2.set_gain(0.7)
        b2.set_conv_plan(plan)
        outs[plan] = b2.process(x[:, :, :frames].contiguous())[:3].clone()
        del b2
    torch.cuda.synchronize()
    assert torch.equal(ys[0][:3].view(torch.int32), outs[2].view(torch.int32))
    assert not torch.equal(outs[1].view(torch.int32), outs[2].view(torch.int32))
