"""Per-stream EQ tables in batch mode (ohs_batch_set_stream_eq_band_coeffs; SURVEY 8b "per-stream or shared EQ tables"): in the
reference every plugin instance owns its bands (src/dsp/parametric_eq.rs:125-164), so N listeners with N AutoEQ profiles are
N StereoParametricEQ objects.  Here: 9 streams x 9 different 10-band tables -- different coefficients, different bands
disabled, different NUMBERS of bands enabled, one stream with none -- in one batch, BIT-EXACT against 9 oracle EQs.

How bit-exactness is told behind the convolution: the batch runs EQ -> convolution -> gain, and the convolution is a
deterministic function of the bits it is fed.  The 9 oracle EQs' outputs go through a second batch whose EQ is off; the
per-stream batch's output must equal that bit for bit -- one differing EQ output bit would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(ohs, synth, S, nb=10):
    """S different tables of nb bands: (coeffs [S][nb][5], enabled [S][nb])"""
    from open_headstage_amd.dsp import BandConfig, FilterType
    rng = np.random.default_rng(77)
    coeffs = np.zeros((S, nb, 5), np.float32)
    en = np.zeros((S, nb), bool)
    types = [FilterType.Peak, FilterType.LowShelf, FilterType.HighShelf]
    for s in range(S):
        for b in range(nb):
            ft = types[(s + b) % 3] if b not in (0, nb - 1) else (FilterType.LowShelf if b == 0 else FilterType.HighShelf)
            fc = float(40.0 * 2.0 ** (b * 0.85 + 0.07 * s))
            coeffs[s, b] = ohs.biquad_coefficients(ft, synth.FS, min(fc, 18000.0), float(0.5 + 0.3 * ((s + 2 * b) % 5)),
                                                   float(rng.uniform(-9, 9)))
            en[s, b] = True
        en[s, s % nb] = False                       # every stream disables a different band
    en[3, 4:] = False                               # stream 3: four bands (three enabled)
    if S > 5:
        en[5, :] = False                            # stream 5: no band at all -- the identity
    if S > 7:
        en[7, :] = True                             # stream 7: all ten, with the stress peak of SURVEY 8d
        coeffs[7, 2] = ohs.biquad_coefficients(FilterType.Peak, synth.FS, 20.0, 10.0, 16.0)
    return coeffs, en


def _oracle_eq(oracle, synth, coeffs, en, x):
    """x [2][n] through one oracle StereoParametricEQ with this table, continuously"""
    q = oracle.StereoParametricEQ(coeffs.shape[0], synth.FS)
    for b in range(coeffs.shape[0]):
        q.set_band_coeffs(b, coeffs[b], bool(en[b]))
    l, r = x[0].copy(), x[1].copy()
    q.process_block(l, r)
    return np.stack([l, r])


def _conv_only(ohs, synth, S, irs, gain):
    bp = ohs.BatchProcessor(S, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    bp.set_eq_enabled(False)
    bp.set_gain(gain)
    bp.set_conv_plan(1)
    return bp


@pytest.mark.parametrize("blocks", [[20, 7], [70, 66], [130, 101]])
def test_nine_streams_nine_tables_bit_exact_against_nine_oracle_eqs(oracle, blocks):
    """two calls each (the EQ state carries over); [70, 66]: calls long enough for the EQ || convolution overlap, i.e. six EQ
    launches per call, each reading the tables again; [130, 101]: launches of 8 192 samples or more, which take the EQ's
    wave-ring form (one chain per wave, csrc/eq_ring64_body.hpp) with every chain's own table"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S = 9
    irs = synth.hrir_set(512)
    coeffs, en = _tables(ohs, synth, S)
    bp = ohs.BatchProcessor(S, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    bp.set_conv_plan(1)
    bp.set_gain(0.8)
    bp.set_eq_enabled(True)
    for s in range(S):
        for b in range(10):
            bp.set_stream_band_coeffs(s, b, coeffs[s, b], bool(en[s, b]))
    ref = _conv_only(ohs, synth, S, irs, 0.8)
    total = sum(blocks) * 512
    x = synth.white_noise(range(300, 300 + S), total)
    xe = np.stack([_oracle_eq(oracle, synth, coeffs[s], en[s], x[s]) for s in range(S)])       # nine oracle EQs
    pos = 0
    for nb in blocks:
        sl = slice(pos, pos + nb * 512)
        y = bp.process(torch.from_numpy(np.ascontiguousarray(x[:, :, sl])).cuda())
        yr = ref.process(torch.from_numpy(np.ascontiguousarray(xe[:, :, sl])).cuda())
        torch.cuda.synchronize()
        for s in range(S):
            assert torch.equal(y[s].view(torch.int32), yr[s].view(torch.int32)), f"stream {s}, call at frame {pos}"
        pos += nb * 512
    assert np.array_equal(xe[5].view(np.uint32), x[5].view(np.uint32))      # (the stream without bands: the oracle passes it through)


def test_shared_calls_set_every_stream_and_the_shared_table_comes_back(oracle):
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S, n = 5, 12 * 512
    irs = synth.hrir_set(512)
    coeffs, en = _tables(ohs, synth, S)
    bands = synth.eq_table()
    bp = ohs.BatchProcessor(S, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    bp.set_conv_plan(1)
    bp.set_eq_enabled(True)
    for i, b in enumerate(bands):
        bp.update_band_coeffs(i, synth.FS, b)               # the shared table first ...
    shared = np.stack([ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db) for b in bands])
    bp.set_stream_band_coeffs(2, 3, coeffs[2, 3], True)     # ... then ONE band of ONE stream: the others keep the shared table
    with pytest.raises(ohs.OhsError):
        bp.set_stream_band_coeffs(S, 0, coeffs[0, 0], True)
    bp.set_stream_band_coeffs(1, 99, coeffs[0, 0], True)    # a band index past the last one is ignored (parametric_eq.rs:144-164)
    ref = _conv_only(ohs, synth, S, irs, 1.0)
    x = synth.white_noise(range(40, 40 + S), 3 * n)
    tabs = [shared.copy() for _ in range(S)]
    tabs[2][3] = coeffs[2, 3]
    ens = [np.ones(10, bool) for _ in range(S)]
    qs = []
    for s in range(S):
        q = oracle.StereoParametricEQ(10, synth.FS)
        for b in range(10):
            q.set_band_coeffs(b, tabs[s][b], True)
        qs.append(q)

    def step(k):
        xs = np.ascontiguousarray(x[:, :, k * n:(k + 1) * n])
        xe = np.empty_like(xs)
        for s in range(S):
            l, r = xs[s, 0].copy(), xs[s, 1].copy()
            qs[s].process_block(l, r)
            xe[s, 0], xe[s, 1] = l, r
        y = bp.process(torch.from_numpy(xs).cuda())
        yr = ref.process(torch.from_numpy(xe).cuda())
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int32), yr.view(torch.int32)), k

    step(0)
    bp.set_band_coeffs(3, coeffs[4, 3], False)              # the shared call now sets band 3 of EVERY stream (state kept)
    for q in qs:
        q.set_band_coeffs(3, coeffs[4, 3], False)
    step(1)
    bp.share_eq_table()                                     # one table again: what the shared calls have set
    for q in qs:
        for b in range(10):
            q.set_band_coeffs(b, shared[b] if b != 3 else coeffs[4, 3], b != 3)
    step(2)


@pytest.mark.parametrize("mode", ["exact_specials", "thirteen_bands", "forty_bands"])
def test_tables_one_ring_pass_cannot_take(oracle, mode):
    """the exact-specials mode (conveyor kernel: -0.0 and non-finite input exact; its table is a kernel argument, so every stream
    is a launch sequence of its own) and a stream with more than 12 enabled bands (a second ring pass for all streams, in which
    the streams with 12 bands or fewer hand their samples on): bit-exact all the same; a stream without bands is still copied"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    nb = {"exact_specials": 10, "thirteen_bands": 16, "forty_bands": 40}[mode]     # (forty: 39 enabled on most streams = four passes)
    S, n = 4, 9 * 512
    irs = synth.hrir_set(512)
    coeffs, en = _tables(ohs, synth, S, nb)
    en[1, :] = False
    if mode == "thirteen_bands":
        en[2, :] = True
        en[2, :3] = False                                   # stream 2: 13 enabled
    bp = ohs.BatchProcessor(S, num_bands=nb)
    for p in range(4):
        bp.set_ir(p, irs[p])
    bp.set_conv_plan(1)
    bp.set_eq_enabled(True)
    bp.set_eq_exact_specials(mode == "exact_specials")
    for s in range(S):
        for b in range(nb):
            bp.set_stream_band_coeffs(s, b, coeffs[s, b], bool(en[s, b]))
    ref = _conv_only(ohs, synth, S, irs, 1.0)
    x = synth.white_noise(range(60, 60 + S), 2 * n)
    if mode == "exact_specials":
        x[0, 0, 100:120] = -0.0                             # what the ring form would turn into +0.0
    xe = np.stack([_oracle_eq(oracle, synth, coeffs[s], en[s], x[s]) for s in range(S)])
    for k in range(2):
        y = bp.process(torch.from_numpy(np.ascontiguousarray(x[:, :, k * n:(k + 1) * n])).cuda())
        yr = ref.process(torch.from_numpy(np.ascontiguousarray(xe[:, :, k * n:(k + 1) * n])).cuda())
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int32), yr.view(torch.int32)), k


def test_node_batch_routes_a_stream_to_the_device_that_holds_it(exp_tuning):
    """ohs_node_batch_set_stream_eq_band_coeffs takes the JOB's stream id: three device slots (sharing the one GPU of the
    box: experiments build), nine streams, the same nine tables as one batch of nine -- the same bits"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import _ffi, synth
    dev = torch.device("cuda:0")
    S, slots, n = 9, 3, 16 * 512
    exp_tuning("node_shared_device_rehearsal", "1")
    L = _ffi.experiments_lib()
    irs = synth.hrir_set(512)
    coeffs, en = _tables(ohs, synth, S)
    nb = ohs.NodeBatchProcessor(S, num_bands=10, devices=[0] * slots, library=L)
    bp = ohs.BatchProcessor(S, num_bands=10, library=L)
    nb.set_tables(irs)
    nb.set_conv_plan(1); bp.set_conv_plan(1)
    for p in range(4):
        bp.set_ir(p, irs[p])
    nb.set_eq_enabled(True); bp.set_eq_enabled(True)
    for s in range(S):
        for b in range(10):
            nb.set_stream_band_coeffs(s, b, coeffs[s, b], bool(en[s, b]))
            bp.set_stream_band_coeffs(s, b, coeffs[s, b], bool(en[s, b]))
    with pytest.raises(ohs.OhsError):
        nb.set_stream_band_coeffs(S, 0, coeffs[0, 0], True)
    x = synth.white_noise_torch(17, S, n, dev)
    shards = [nb.shard(i) for i in range(slots)]
    xs = [x[f:f + c].contiguous() for _, f, c in shards]
    ys = [torch.empty_like(a) for a in xs]
    torch.cuda.synchronize()
    nb.process(xs, ys); nb.sync()
    y = bp.process(x)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(ys).view(torch.int32), y.view(torch.int32))
