"""The block-2048 / FFT-4096 plan for long impulse responses (ohs_batch_set_conv_plan(2) with taps > 512;
csrc/conv_lb_kernels.hip) against the oracle -- the restatement of the reference's uniform block-512 partitioning
(src/dsp/convolution.rs:120-132,236-289) -- and against f64 direct convolution: every remainder of the call length modulo
the 2048-frame block, calls that continue the previous call's window grid and calls that do not, several segments per
call, paths of different lengths and a muted one, in place, with the EQ in front, across calls that alternate between the
plans (the block-512 state is rebuilt from the input history and back), across a per-path set_ir in mid-stream, and at
BASELINE config 4's shape."""
import numpy as np
import pytest

from tests.util import assert_parity, rms_err

pytestmark = pytest.mark.gpu


def _oracle_engines(oracle, irs, S):
    engs = []
    for _ in range(S):
        e = oracle.ConvolutionEngine()
        for p in range(4):
            e.set_ir(p, irs[p])
        engs.append(e)
    return engs


def _make(ohs, S, irs, plan, gain=1.0):
    bp = ohs.BatchProcessor(S, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    bp.set_eq_enabled(False)
    bp.set_gain(gain)
    bp.set_conv_plan(plan)
    return bp


@pytest.mark.parametrize("taps,blocks", [(16384, [9, 4, 7, 8, 8, 1, 2, 40]), (4097, [3, 5, 16, 16]), (3600, [6, 6, 12])])
def test_block_2048_plan_matches_the_oracle_and_f64(oracle, taps, blocks):
    """call lengths of every remainder modulo 4 blocks: the 2048-frame blocks sit on the stream's absolute grid, a call that ends
    inside one computes it with zeros behind its frames and the next call recomputes it from the input history; in place"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S = 3
    irs = synth.hrir_set(taps)
    bp = _make(ohs, S, irs, 2, gain=0.9)
    engs = _oracle_engines(oracle, irs, S)
    total = sum(blocks) * 512
    x = synth.white_noise(range(S), total)
    got, pos = [], 0
    for k, nb in enumerate(blocks):
        xt = torch.from_numpy(np.ascontiguousarray(x[:, :, pos:pos + nb * 512])).cuda()
        bp.process(xt, out=xt)          # in place
        assert bp.last_conv_plan()[0] == "block2048", (k, bp.last_conv_plan())
        got.append(xt.cpu().numpy())
        pos += nb * 512
    g = np.concatenate(got, axis=2)
    for s in range(S):
        ref = np.stack(engs[s].process_block(x[s, 0], x[s, 1])) * np.float32(0.9)
        assert_parity(g[s], ref, f"taps {taps}, stream {s}")
    yl, yr = oracle.binaural_f64(x[0, 0], x[0, 1], irs)
    a, r = rms_err(g[0], 0.9 * np.stack([yl, yr]))
    print(f"block-2048 plan vs f64 direct convolution, {taps} taps: {r:.2e}")
    assert r <= 1e-6


@pytest.mark.parametrize("seed", [11, 12])
def test_random_call_lengths_continue_one_another(oracle, seed):
    """calls of 1 .. 9 blocks in random order (the plugin's 1024-frame block is 2): every call starts and ends somewhere on the
    512 grid inside the absolute grid's 2048-frame blocks, continues the ring of window spectra the previous call left and
    recomputes the block it starts in; a reset and a plan-1 call in between force a fresh start from the input history"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    rng = np.random.default_rng(seed)
    S, taps = 2, int(rng.integers(3600, 9000))
    irs = synth.hrir_set(taps)
    bp = _make(ohs, S, irs, 0)
    engs = _oracle_engines(oracle, irs, S)
    for k in range(34):
        nb = int(rng.integers(1, 10))
        if k == 20:
            bp.set_conv_plan(1)
        if k == 21:
            bp.set_conv_plan(0)
        if k == 27:
            bp.reset()
            engs = _oracle_engines(oracle, irs, S)
        x = synth.white_noise(range(1000 * seed + 10 * k, 1000 * seed + 10 * k + S), nb * 512)
        xt = torch.from_numpy(x).cuda()
        y = (bp.process(xt, out=xt) if k % 3 == 0 else bp.process(xt)).cpu().numpy()
        assert bp.last_conv_plan()[0] == ("block512_tp" if k == 20 else "block2048"), (k, bp.last_conv_plan())
        for s in range(S):
            ref = np.stack(engs[s].process_block(x[s, 0], x[s, 1]))
            assert_parity(y[s], ref, f"seed {seed}, call {k} ({nb} blocks), stream {s}")


def test_short_calls_through_the_block_512_kernels_incl_a_per_path_set_ir(oracle):
    """calls of 1, 2 and 3 blocks on a long response are served by the time-parallel block-512 kernels
    (MAC tiles of 4 blocks) when the reference blocking is asked for -- until round 5 they took the sequential kernel.
    Sequences of such calls, a per-path set_ir in mid-stream (short calls right behind it go through the sequential kernel,
    which carries the per-path gates, until a call reaches past every old block), and long calls in between."""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S, taps = 4, 16384
    irs = synth.hrir_set(taps)
    bp = _make(ohs, S, irs, 1)
    engs = _oracle_engines(oracle, irs, S)
    seq = [1, 2, 3, 2, 1, 40, 2, 2, 3, 1, "set_ir", 2, 1, 3, 36, 2, 3, 1]
    k = 0
    for item in seq:
        if item == "set_ir":
            h = synth.hrir_set(9000)[1]
            bp.set_ir(1, h)
            for e in engs:
                e.set_ir(1, h)
            continue
        nb = item
        x = synth.white_noise(range(7000 + 10 * k, 7000 + 10 * k + S), nb * 512)
        y = bp.process(torch.from_numpy(x).cuda()).cpu().numpy()
        fam = bp.last_conv_plan()[0]
        assert fam in ("block512_tp", "sequential"), (k, fam)
        if k < 10:
            assert fam == "block512_tp", (k, nb, fam)       # every call in front of the set_ir, the 1-block ones included
        for s in range(S):
            ref = np.stack(engs[s].process_block(x[s, 0], x[s, 1]))
            assert_parity(y[s], ref, f"call {k} ({nb} blocks, {fam}), stream {s}")
        k += 1
    assert bp.last_conv_plan()[0] == "block512_tp"


def test_paths_of_different_lengths_and_a_muted_path(oracle):
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S = 2
    full = synth.hrir_set(9000)
    irs = [full[0], full[1][:5000], full[2][:700], np.zeros(0, np.float32)]
    bp = _make(ohs, S, irs, 2)
    engs = _oracle_engines(oracle, irs, S)
    x = synth.white_noise(range(7, 7 + S), 2 * 44 * 512)
    ys = []
    for c in range(2):
        xt = torch.from_numpy(np.ascontiguousarray(x[:, :, c * 44 * 512:(c + 1) * 44 * 512])).cuda()
        ys.append(bp.process(xt).cpu().numpy())
        assert bp.last_conv_plan()[0] == "block2048"
    g = np.concatenate(ys, axis=2)
    for s in range(S):
        assert_parity(g[s], np.stack(engs[s].process_block(x[s, 0], x[s, 1])), f"mixed lengths, stream {s}")


def test_several_segments_per_call(oracle):
    """A launch pair serves at most 256 blocks of 2048 per stream (131 072 windows over all streams: the ring stays within
    4 GiB), so a 1 102-block call is two segments, the second one short and ragged; streams 0, 19 and 39 against the oracle, every
    stream against the stream it repeats"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    dev = torch.device("cuda:0")
    S, nb, taps = 40, 1102, 4200
    irs = synth.hrir_set(taps)
    bp = _make(ohs, S, irs, 2)
    base = synth.white_noise_torch(21, 3, 2 * nb * 512, dev)
    x = base[torch.arange(S, device=dev) % 3].contiguous()
    ys = []
    for c in range(2):          # (in place: out of place a call this long would go to the block-8192 kernel)
        xc = x[:, :, c * nb * 512:(c + 1) * nb * 512].contiguous()
        bp.process(xc, out=xc)
        ys.append(xc)
    y = torch.cat(ys, dim=2)
    torch.cuda.synchronize()
    assert bp.last_conv_plan()[0] == "block2048"
    for s in range(3, S):
        assert torch.equal(y[s], y[s % 3]), f"stream {s} differs from stream {s % 3}"
    engs = _oracle_engines(oracle, irs, 3)
    xb = base.cpu().numpy()
    for s in (0, 19, 39):
        ref = np.stack(engs[s % 3].process_block(xb[s % 3, 0], xb[s % 3, 1]))
        assert_parity(y[s].cpu().numpy(), ref, f"multi-segment, stream {s}")
        engs[s % 3] = _oracle_engines(oracle, irs, 1)[0]


def test_the_plans_hand_the_state_over_and_a_midstream_set_ir(oracle):
    """One continuous signal through a sequence of calls under changing plans: block 2048 -> block 512 time-parallel (24
    blocks, 2 blocks) -> block 2048 ..., with the EQ in front, and a per-path set_ir in mid-stream -- that path
    must not see the frames before its reset (convolution.rs:135-138), the others must: the reset is carried out as "every
    path forgets" plus pending tails (what the old frames still owe the output through the surviving paths), so the call
    right behind it is block 2048 as well."""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S, taps = 3, 8192
    irs = synth.hrir_set(taps)
    bands = synth.eq_table()
    bp = ohs.BatchProcessor(S, num_bands=len(bands))
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(bands):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    bp.set_gain(0.7)
    engs = _oracle_engines(oracle, irs, S)
    eqs = []
    for s in range(S):
        qo = oracle.StereoParametricEQ(len(bands), synth.FS)
        for i, b in enumerate(bands):
            qo.set_band_coeffs(i, ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db), b.enabled)
        eqs.append(qo)
    #      (plan, blocks, kernel family expected of the call's last launch)
    seq = [(2, 20, "block2048"), (1, 24, "block512_tp"), (2, 8, "block2048"), (1, 2, "block512_tp"), (2, 16, "block2048"),
           (2, 16, "block2048"),        # <- set_ir(path 2) in front of this one: pending tails carry the other paths' past
           (2, 20, "block2048"), (0, 3, "block2048"), (0, 40, "block2048")]
    for k, (plan, nb, want) in enumerate(seq):
        if k == 5:
            h = synth.hrir_set(6000)[2]
            bp.set_ir(2, h)
            for e in engs:
                e.set_ir(2, h)
        bp.set_conv_plan(plan)
        x = synth.white_noise(range(10 * k, 10 * k + S), nb * 512)
        y = bp.process(torch.from_numpy(x).cuda()).cpu().numpy()
        assert bp.last_conv_plan()[0] == want, (k, bp.last_conv_plan(), want)
        for s in range(S):
            l, r = x[s, 0].copy(), x[s, 1].copy()
            oracle.chain_process(engs[s], eqs[s], l, r, eq_enable=True, gain=0.7)
            assert_parity(y[s], np.stack([l, r]), f"call {k} (plan {plan}, {nb} blocks), stream {s}")


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_per_path_resets_in_quick_succession_pending_tails(oracle, seed):
    """A per-path set_ir on an object that keeps the input history = "every path forgets" + pending tails, one row per path
    (conv_lb_tail_route).  Resets that come while tails are still pending -- the same path again, another path, a mute, a
    response that changes the history's reach (and with it the rows' length), all four paths one after the other -- and
    calls of 1 .. 20 blocks in between, some shorter than the tails, in place, under every plan: one continuous signal
    against the oracle, call by call."""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    rng = np.random.default_rng(seed)
    S = 3
    taps0 = [int(rng.integers(4000, 12000)) for _ in range(4)]
    irs = [synth.hrir_set(t)[p] for p, t in enumerate(taps0)]
    bp = _make(ohs, S, irs, 0, gain=0.8)
    engs = _oracle_engines(oracle, irs, S)
    lengths = [0, 300, 512, 3000, 4100, 9000, 16384, 20000]
    for k in range(40):
        r = rng.random()
        if k in (17, 18, 19, 20):               # every path re-loaded, one call each, no processing in between
            todo = [(k - 17, int(rng.choice(lengths[3:])))]
        elif r < 0.45:
            todo = [(int(rng.integers(0, 4)), int(rng.choice(lengths)))]
            if rng.random() < 0.3:
                todo.append((int(rng.integers(0, 4)), int(rng.choice(lengths))))
        else:
            todo = []
        for path, t in todo:
            h = synth.hrir_set(max(t, 1))[path][:t]
            bp.set_ir(path, h)
            for e in engs:
                e.set_ir(path, h)
        if k in (17, 18, 19):
            continue
        bp.set_conv_plan(int(rng.integers(0, 3)))
        nb = int(rng.choice([1, 2, 3, 5, 8, 20]))
        x = synth.white_noise(range(5000 * seed + 10 * k, 5000 * seed + 10 * k + S), nb * 512)
        xt = torch.from_numpy(x).cuda()
        y = (bp.process(xt, out=xt) if k % 2 else bp.process(xt)).cpu().numpy()
        for s in range(S):
            ref = np.stack(engs[s].process_block(x[s, 0], x[s, 1])) * np.float32(0.8)
            assert_parity(y[s], ref, f"seed {seed}, call {k} ({nb} blocks, {bp.last_conv_plan()[0]}), stream {s}")


def test_reset_and_a_set_ir_of_all_paths_forget_the_input_history(oracle):
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S, taps = 2, 5000
    irs = synth.hrir_set(taps)
    bp = _make(ohs, S, irs, 2)
    x = synth.white_noise(range(S), 24 * 512)
    xt = torch.from_numpy(x).cuda()
    y0 = bp.process(xt).cpu().numpy()
    bp.reset()
    y1 = bp.process(xt).cpu().numpy()           # the same frames from zeroed state: the same bits
    assert np.array_equal(y0.view(np.int32), y1.view(np.int32))
    for p in range(4):                          # every path re-loaded: each forgets its past (convolution.rs:135-138)
        bp.set_ir(p, irs[p])
    y2 = bp.process(xt).cpu().numpy()
    assert bp.last_conv_plan()[0] == "block2048"
    engs = _oracle_engines(oracle, irs, S)
    for s in range(S):
        ref = np.stack(engs[s].process_block(x[s, 0], x[s, 1]))
        assert_parity(y2[s], ref, f"after set_ir of all paths, stream {s}")
        assert_parity(y0[s], ref, f"first run, stream {s}")


def test_the_library_picks_block_2048_for_long_calls_on_long_responses():
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    dev = torch.device("cuda:0")
    for taps, nb, want in ((16384, 64, "block2048"), (16384, 12, "block2048"), (16384, 2, "block2048"),
                           (3000, 64, "block2048"), (1024, 2, "block2048"), (513, 7, "block2048"), (512, 64, "block512_p1"),
                           (512, 7, "block512_p1"), (4096, 16, "block2048")):
        bp = _make(ohs, 4, synth.hrir_set(taps), 0)
        x = synth.white_noise_torch(3, 4, nb * 512, dev)
        bp.process(x)
        torch.cuda.synchronize()
        assert bp.last_conv_plan()[0] == want, (taps, nb, bp.last_conv_plan())
    # unaligned buffers / strides that are not multiples of 4 floats: served by the block-512 kernels, same signal
    bp = _make(ohs, 2, synth.hrir_set(8192), 2)
    ref = _make(ohs, 2, synth.hrir_set(8192), 1)
    n = 32 * 512
    buf = torch.zeros(2 * 2 * (n + 1) + 1, device=dev)
    src = synth.white_noise_torch(5, 2, n, dev)
    view = buf[1:1 + 2 * 2 * (n + 1)].view(2, 2, n + 1)[:, :, :n]
    view.copy_(src)
    out = torch.empty_like(buf)
    bp.process_ptr(view.data_ptr(), out[1:].data_ptr(), n // 512, 2 * (n + 1), n + 1,
                   torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    assert bp.last_conv_plan()[0] == "block512_tp"
    y = out[1:1 + 2 * 2 * (n + 1)].view(2, 2, n + 1)[:, :, :n]
    assert torch.equal(y.contiguous().view(torch.int32), ref.process(src).view(torch.int32))


def test_config4_shape_both_plans(oracle):
    """BASELINE config 4: 64 streams, 4 x 16 384 taps, 256-block calls: the library's choice out of place (the block-8192 kernel)
    and in place (the block-2048 plan) against the block-512 plan on all streams and against the oracle on three, over two calls
    (the second one continues from what the first one left)"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    dev = torch.device("cuda:0")
    S, nb, taps = 64, 256, 16384
    irs = synth.hrir_set(taps)
    a, b, c2 = _make(ohs, S, irs, 1), _make(ohs, S, irs, 0), _make(ohs, S, irs, 0)
    x = synth.white_noise_torch(50, S, 2 * nb * 512, dev)
    ya = torch.cat([a.process(x[:, :, c * nb * 512:(c + 1) * nb * 512].contiguous()) for c in range(2)], dim=2)
    yb = torch.cat([b.process(x[:, :, c * nb * 512:(c + 1) * nb * 512].contiguous()) for c in range(2)], dim=2)
    ycs = []
    for c in range(2):
        xc = x[:, :, c * nb * 512:(c + 1) * nb * 512].contiguous()
        c2.process(xc, out=xc)
        ycs.append(xc)
    yc = torch.cat(ycs, dim=2)
    torch.cuda.synchronize()
    assert a.last_conv_plan()[0] == "block512_tp" and b.last_conv_plan() == ("block8192", 4) and c2.last_conv_plan() == ("block2048", 16)
    xs = x.cpu().numpy()
    for name, yy in (("block 8192", yb), ("block 2048", yc)):
        d = (ya - yy).double()
        per_stream = torch.sqrt((d * d).mean(dim=(1, 2))) / torch.sqrt((ya.double() ** 2).mean(dim=(1, 2)))
        print(f"{name} vs block 512 at config 4: worst stream {float(per_stream.max()):.2e}")
        assert float(per_stream.max()) <= 1e-6
        for s in (0, 31, 63):
            e = _oracle_engines(oracle, irs, 1)[0]
            ref = np.stack(e.process_block(xs[s, 0], xs[s, 1]))
            assert_parity(yy[s].cpu().numpy(), ref, f"config 4, {name}, stream {s}")
