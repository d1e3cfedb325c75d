"""The overlap-save plan of the one-partition batch path (ohs_batch_set_conv_plan(2): hop 1536 / FFT 2048,
csrc/conv_os_kernels.hip) against the oracle, the f64 ground truth and the block-512 plan: every hop-count remainder (the
last window holds 512, 1024 or 1536 new frames), every chunk count the launcher picks (1, 2, 3, 4, 6, 12 hop ranges per stream),
in place, with the EQ in front, across calls that alternate between the two plans (they hand the same state over: the
merged overlap and the last input block), and across a per-path set_ir in mid-stream."""
import numpy as np
import pytest

from tests.util import assert_parity, rms_err

pytestmark = pytest.mark.gpu


def _make(ohs, synth, S, plan, taps=512, eq=False, gain=1.0):
    bp = ohs.BatchProcessor(S, num_bands=10)
    irs = synth.hrir_set(taps)
    for p in range(4):
        bp.set_ir(p, irs[p])
    if eq:
        for i, b in enumerate(synth.eq_table()):
            bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(eq)
    bp.set_gain(gain)
    bp.set_conv_plan(plan)
    return bp, irs


@pytest.mark.parametrize("n_blocks", [3, 4, 5, 6, 7, 8, 9, 31, 66])
def test_overlap_save_plan_matches_the_oracle_and_f64(oracle, n_blocks):
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S, calls = 5, 3
    bp, irs = _make(ohs, synth, S, 2, gain=0.9)
    x = synth.white_noise(range(S), n_blocks * 512 * calls)
    outs = []
    for c in range(calls):
        xt = torch.from_numpy(np.ascontiguousarray(x[:, :, c * n_blocks * 512:(c + 1) * n_blocks * 512])).cuda()
        bp.process(xt, out=xt)          # in place
        outs.append(xt.cpu().numpy())
    g = np.concatenate(outs, axis=2)
    for s in range(S):
        eo = oracle.ConvolutionEngine()
        for p in range(4):
            eo.set_ir(p, irs[p])
        ref = np.stack(eo.process_block(x[s, 0], x[s, 1])) * np.float32(0.9)
        assert_parity(g[s], ref, f"{n_blocks} blocks x {calls} calls, stream {s}")
    yl, yr = oracle.binaural_f64(x[0, 0], x[0, 1], irs)
    a, r = rms_err(g[0], 0.9 * np.stack([yl, yr])[:, :g.shape[2]])
    print(f"hop-1536 plan vs f64 direct convolution: {r:.2e}")
    assert r <= 1e-6


# (hop ranges per stream the launcher picks IN PLACE with twelve waves per CU on 256 CUs -- divisors of 12: api_conv.hip;
# OUT OF PLACE any count goes: 5 for 600 x 60, 10 for 300 x 100, 34 -- one hop each -- for 37 x 100)
@pytest.mark.parametrize("S,n_blocks,want_chunks", [(2048, 12, 1), (1536, 40, 2), (1000, 24, 3), (600, 60, 4), (300, 100, 6),
                                                  (37, 100, 12)])
def test_every_chunk_count_of_the_launcher(S, n_blocks, want_chunks):
    """against the block-512 plan on all streams (both are 2-3e-7 from the truth), in place and out of place"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    dev = torch.device("cuda:0")
    a, _ = _make(ohs, synth, S, 1)
    b, _ = _make(ohs, synth, S, 2)
    c, _ = _make(ohs, synth, S, 2)
    for call in range(2):
        x = synth.white_noise_torch(9 * call, S, n_blocks * 512, dev)
        ya = torch.empty_like(x)
        a.process(x, out=ya)
        yb = x.clone()
        b.process(yb, out=yb)               # in place: a divisor of twelve hop ranges per stream
        yc = torch.empty_like(x)
        c.process(x, out=yc)                # out of place: the count that fills the chip
        torch.cuda.synchronize()
        # a hop is computed on its own: how a stream's hops are dealt to waves does not show in the bits
        assert torch.equal(yb.view(torch.int32), yc.view(torch.int32)), call
        for y2, what in ((yb, "in place"), (yc, "out of place")):
            d = (ya - y2).double()
            rel = float(torch.sqrt((d * d).mean()) / torch.sqrt((ya.double() ** 2).mean()))
            per_stream = torch.sqrt((d * d).mean(dim=(1, 2))) / torch.sqrt((ya.double() ** 2).mean(dim=(1, 2)))
            assert rel <= 6e-7 and float(per_stream.max()) <= 1e-6, (what, call, rel, float(per_stream.max()))


def test_the_two_plans_hand_the_same_state_over_and_a_midstream_set_ir(oracle):
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S = 4
    bp, irs = _make(ohs, synth, S, 0, eq=True, gain=0.7)
    bands = synth.eq_table()
    engs, eqs = [], []
    for s in range(S):
        eo = oracle.ConvolutionEngine()
        for p in range(4):
            eo.set_ir(p, irs[p])
        qo = oracle.StereoParametricEQ(len(bands), synth.FS)
        for i, b in enumerate(bands):
            qo.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), b.enabled)
        engs.append(eo); eqs.append(qo)
    seq = [(2, 7), (1, 5), (2, 4), (2, 10), (1, 3), (2, 68)]       # (plan, blocks)
    for k, (plan, nb) in enumerate(seq):
        if k == 3:      # a shorter response on one path in mid-stream: the lazy state is materialised with the OLD spectra
            h = irs[1][:200]
            bp.set_ir(1, h)
            for e in engs:
                e.set_ir(1, h)
        bp.set_conv_plan(plan)
        x = synth.white_noise(range(10 * k, 10 * k + S), nb * 512)
        y = bp.process(torch.from_numpy(x).cuda()).cpu().numpy()
        for s in range(S):
            l, r = x[s, 0].copy(), x[s, 1].copy()
            oracle.chain_process(engs[s], eqs[s], l, r, eq_enable=True, gain=0.7)
            assert_parity(y[s], np.stack([l, r]), f"call {k} (plan {plan}, {nb} blocks), stream {s}")


def test_plan_argument_and_fallbacks():
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    bp, irs = _make(ohs, synth, 2, 2)
    with pytest.raises(ohs.OhsError):
        bp.set_conv_plan(3)
    # two blocks are fewer than a hop needs, odd strides cannot take 8-byte accesses: the block-512 kernel serves them
    ref, _ = _make(ohs, synth, 2, 1)
    x = torch.from_numpy(synth.white_noise(range(2), 2 * 512)).cuda()
    assert torch.equal(bp.process(x).view(torch.int32), ref.process(x).view(torch.int32))
    # longer responses are not this plan's business: plan 2 means block 2048 / FFT 4096 for them (tests/test_gpu_conv_lb.py)
    bp.set_ir(0, synth.hrir_set(1300)[0]); ref.set_ir(0, synth.hrir_set(1300)[0])
    # (the call right behind this set_ir: one-partition objects keep no input history, only the last block's input -- which is
    # all the other paths' tails need; the set_ir seeds the history from it, so the block-2048 plan serves this call already)
    def rel(ya, yb):
        d = (ya - yb).double()
        return float(torch.sqrt((d * d).mean()) / torch.sqrt((yb.double() ** 2).mean()))
    for seed in (5, 8):
        x = torch.from_numpy(synth.white_noise(range(seed, seed + 2), 12 * 512)).cuda()
        ya, yb = bp.process(x), ref.process(x)
        assert bp.last_conv_plan()[0] == "block2048" and ref.last_conv_plan()[0] == "block512_tp"
        assert rel(ya, yb) <= 1e-6


def test_the_library_picks_the_hop_plan_where_it_pays(oracle):
    """ohs_batch_set_conv_plan(0), the default: the hop-1536 plan for calls of 48 blocks or more on 512 streams or more --
    256 or more when the convolution runs out of place, i.e. with the EQ off and distinct buffers -- and the block-512
    plan otherwise (profiles/r04_conv_plan_ab_run3.txt, _run4.txt); BASELINE configs[2] (256 streams, EQ on) keeps the
    bits the golden hash records.  Told by the bits: each choice must equal the explicit plan it stands for."""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    dev = torch.device("cuda:0")
    #     streams, blocks per call, EQ, in place, the plan the library must pick
    for S, n_blocks, eq, in_place, want in ((512, 60, True, False, 2), (512, 40, True, False, 1), (256, 60, True, False, 1),
                                            (700, 48, True, False, 2), (256, 60, False, False, 2), (256, 60, False, True, 1),
                                            (200, 60, False, False, 1), (256, 40, False, False, 1), (512, 60, False, True, 2)):
        base = synth.white_noise_torch(90 + S, 3, 2 * n_blocks * 512, dev)
        x = base[torch.arange(S, device=dev) % 3].contiguous()
        out = {}
        for plan in (0, 1, 2):
            bp, irs = _make(ohs, synth, S, plan, eq=eq, gain=0.7)
            ys = []
            for c in range(2):
                xc = x[:, :, c * n_blocks * 512:(c + 1) * n_blocks * 512].contiguous()
                if in_place:
                    bp.process(xc, out=xc)
                    ys.append(xc)
                else:
                    ys.append(bp.process(xc))
            out[plan] = torch.cat(ys, dim=2)
        torch.cuda.synchronize()
        assert torch.equal(out[0].view(torch.int32), out[want].view(torch.int32)), (S, n_blocks, eq, in_place, want)
        assert not torch.equal(out[1].view(torch.int32), out[2].view(torch.int32))
        if want == 2:           # ... and what it picked is within the parity bar of the oracle
            for s in (0, 1, 2):
                eng = oracle.ConvolutionEngine()
                for p in range(4):
                    eng.set_ir(p, irs[p])
                bands = synth.eq_table()
                eqo = oracle.StereoParametricEQ(len(bands), synth.FS)
                for i, b in enumerate(bands):
                    eqo.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), True)
                xs = base[s].cpu().numpy()
                l, r = xs[0].copy(), xs[1].copy()
                oracle.chain_process(eng, eqo, l, r, eq_enable=eq, gain=0.7)
                assert_parity(out[0][s].cpu().numpy(), np.stack([l, r]), f"auto plan, {S} streams, stream {s}")
