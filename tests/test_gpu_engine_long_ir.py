"""One engine, a response of many partitions, the plugin's host blocks (the reference's own use of `process_block`,
convolution.rs:141-182, with a room response: convolution.rs:120-132, 261-273).

The engine kernels compute a block's sum over the partitions that pair with blocks ALREADY in the ring ahead of time --
behind the previous call's completion word (kernels.h: ConvGeneralArgs::pre) -- so that a call only waits for the
partitions that pair with its own blocks.  These tests drive that split through everything that can invalidate what was
computed ahead: changing call lengths (fewer / more blocks than were prepared, calls that complete no block), long
calls that leave the engine kernels, per-path set_ir in mid-stream (the reference's per-path reset, convolution.rs:135-138),
different partition counts per path, clone, the switch between the launch-per-call and the resident form, the denormal
mode.  Reference: the oracle (f32 restatement) at the parity bar, and f64 direct convolution for the steady case."""
import numpy as np
import pytest

from tests.util import assert_parity, setup_engines

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ohs():
    import open_headstage_amd
    return open_headstage_amd


@pytest.fixture()
def synth():
    from open_headstage_amd import synth as s
    return s


def _run(ge, eo, x, sizes, o=0):
    g, r = [], []
    for n in sizes:
        l, rr = x[0, o:o + n].copy(), x[1, o:o + n].copy()
        g.append(np.stack(ge.process_block(l, rr)))
        r.append(np.stack(eo.process_block(l, rr)))
        o += n
    return np.concatenate(g, 1), np.concatenate(r, 1), o


@pytest.mark.parametrize("realtime", [False, True])
@pytest.mark.parametrize("taps", [1024, 16384])
def test_plugin_blocks_on_a_room_response(ohs, synth, oracle, taps, realtime):
    ge = ohs.ConvolutionEngine.new()
    irs = synth.hrir_set(taps)
    eo = setup_engines(oracle, ge, irs)
    ge.set_realtime(realtime)
    x = synth.white_noise([21], 1024 * 48)[0]
    g, r, _ = _run(ge, eo, x, [1024] * 48)
    assert_parity(g, r, f"{taps} taps, 1024-frame calls, realtime={realtime}")
    # f64 direct convolution of the same frames (no FIFO latency at n % 512 == 0)
    ref = np.zeros((2, x.shape[1]))
    xd = x.astype(np.float64)
    h = [np.asarray(v, np.float64) for v in irs]
    ref[0] = np.convolve(xd[0], h[0])[:x.shape[1]] + np.convolve(xd[1], h[2])[:x.shape[1]]
    ref[1] = np.convolve(xd[0], h[1])[:x.shape[1]] + np.convolve(xd[1], h[3])[:x.shape[1]]
    assert_parity(g, ref, f"{taps} taps vs f64 direct convolution, realtime={realtime}")


@pytest.mark.parametrize("realtime", [False, True])
def test_changing_call_lengths_and_partition_counts_per_path(ohs, synth, oracle, realtime):
    ge = ohs.ConvolutionEngine.new()
    irs = [synth.hrir_set(t)[p] for p, t in enumerate((5000, 512, 16384, 1500))]
    eo = setup_engines(oracle, ge, irs)
    ge.set_realtime(realtime)
    # 1 block prepared -> 3 needed; 4 prepared -> 1 needed; calls that complete no block; 8 blocks (the launch-per-call
    # kernel's limit); 10 000 frames (leaves the engine kernels: what was computed ahead is stale afterwards)
    sizes = [512, 1536, 2048, 512, 100, 100, 400, 1024, 4096, 1024, 10000, 1024, 1024, 37, 1000, 1024, 3072, 1024]
    x = synth.white_noise([22], sum(sizes))[0]
    g, r, _ = _run(ge, eo, x, sizes)
    assert_parity(g, r, f"ragged calls, realtime={realtime}")


@pytest.mark.parametrize("realtime", [False, True])
def test_per_path_set_ir_in_mid_stream_forgets_what_was_computed_ahead(ohs, synth, oracle, realtime):
    ge = ohs.ConvolutionEngine.new()
    irs = synth.hrir_set(8192)
    eo = setup_engines(oracle, ge, irs)
    ge.set_realtime(realtime)
    x = synth.white_noise([23], 1024 * 64)[0]
    g1, r1, o = _run(ge, eo, x, [1024] * 20)
    # a path gets another response of the same length: the counter does not move, the sums computed ahead belong to the
    # old response and to blocks that path may no longer see (convolution.rs:135-138)
    new = synth.hrir_set(8192)[2][::-1].copy()
    ge.set_ir(1, new); eo.set_ir(1, new)
    g2, r2, o = _run(ge, eo, x, [1024] * 20, o)
    # ... a shorter one, then silence on a path (len 0: one zero partition, convolution.rs:114-118)
    short = synth.hrir_set(700)[0]
    ge.set_ir(2, short); eo.set_ir(2, short)
    ge.set_ir(0, np.zeros(0, np.float32)); eo.set_ir(0, np.zeros(0, np.float32))
    g3, r3, o = _run(ge, eo, x, [1024] * 20, o)
    assert_parity(np.concatenate([g1, g2, g3], 1), np.concatenate([r1, r2, r3], 1), f"set_ir in mid-stream, realtime={realtime}")


def test_clone_mode_switch_and_denormal_mode(ohs, synth, oracle):
    ge = ohs.ConvolutionEngine.new()
    irs = synth.hrir_set(4096)
    eo = setup_engines(oracle, ge, irs)
    x = synth.white_noise([24], 1024 * 80)[0]
    g, r, o = _run(ge, eo, x, [1024] * 10)
    assert_parity(g, r, "before the clone")
    gc, oc = ge.clone(), eo.clone()
    ge.set_realtime(True)                   # the original goes on in the resident form, the clone launch per call
    ga, ra, o2 = _run(ge, eo, x, [1024] * 10, o)
    gb, rb, _ = _run(gc, oc, x, [1024, 512, 1536, 1024], o)
    assert_parity(ga, ra, "original after the clone (resident)")
    assert_parity(gb, rb, "clone")
    ge.set_realtime(False)                  # back: the launch-per-call kernel finds what the resident one prepared
    ga, ra, o2 = _run(ge, eo, x, [1024] * 6, o2)
    assert_parity(ga, ra, "resident -> launch per call")
    ge.set_flush_denormals(1)               # (sums prepared in the old mode are dropped; plain signals: same values)
    ga, ra, o2 = _run(ge, eo, x, [1024] * 6, o2)
    assert_parity(ga, ra, "after a change of the denormal mode")


@pytest.mark.parametrize("taps", [1300, 16384])
def test_resident_and_launch_per_call_agree_bit_for_bit(ohs, synth, taps):
    irs = synth.hrir_set(taps)
    a, b = ohs.ConvolutionEngine.new(), ohs.ConvolutionEngine.new()
    for p in range(4):
        a.set_ir(p, irs[p]); b.set_ir(p, irs[p])
    a.set_realtime(True)
    sizes = [1024, 1024, 512, 2048, 100, 1024, 1536, 4096, 1024, 300, 1024, 9000, 1024, 1024]
    x = synth.white_noise([25], sum(sizes))[0]
    o = 0
    for n in sizes:
        l, r = x[0, o:o + n].copy(), x[1, o:o + n].copy()
        ga, gb = np.stack(a.process_block(l, r)), np.stack(b.process_block(l, r))
        assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32)), (taps, n, o)
        o += n


@pytest.mark.parametrize("realtime,n", [(False, 1024), (True, 1024), (True, 256), (False, 1536)])
def test_an_impulse_reads_every_partition_back(ohs, synth, realtime, n):
    """The reference's identity / delay known-answer tests (convolution.rs:317-383) on a response of 32 partitions: an
    impulse in the left channel must come back as Lsl's taps in the left ear and Lsr's in the right, tap for tap -- every
    partition paired with the right block, whether its product was computed in the call or ahead of it."""
    taps = 16384
    irs = synth.hrir_set(taps)
    e = ohs.ConvolutionEngine.new()
    for p in range(4):
        e.set_ir(p, irs[p])
    e.set_realtime(realtime)
    pos = 700                                       # not on a block boundary
    total = ((pos + taps + 2048) // n + 1) * n
    x = np.zeros((2, total), np.float32)
    x[0, pos] = 1.0
    outs = []
    for c in range(total // n):
        outs.append(np.stack(e.process_block(x[0, c * n:(c + 1) * n].copy(), x[1, c * n:(c + 1) * n].copy())))
    y = np.concatenate(outs, 1)
    lat = 0 if n % 512 == 0 else int(e.latency_samples(n))
    for ear, path in ((0, 0), (1, 1)):
        got = y[ear, lat + pos:lat + pos + taps]
        want = np.asarray(irs[path], np.float32)
        assert np.abs(got - want).max() <= 2e-6 * max(1.0, float(np.abs(want).max())), (ear, float(np.abs(got - want).max()))
    assert np.abs(y[:, :lat + pos]).max() <= 1e-6 and np.abs(y[:, lat + pos + taps:]).max() <= 1e-6


@pytest.mark.parametrize("taps", [512, 3000, 16384])
def test_host_blocks_of_up_to_8192_frames(ohs, synth, oracle, taps):
    """Host blocks of 4 097 .. 8 192 frames are served as two calls of the one-launch kernel (few partitions) or by the
    staged path (many; and whenever the call answers with silence, convolution.rs:176-181): the reference's FIFO
    semantics either way -- ragged sizes, a first call that cannot pop its frames, the remainder carried over."""
    ge = ohs.ConvolutionEngine.new()
    irs = synth.hrir_set(taps)
    eo = setup_engines(oracle, ge, irs)
    sizes = [5000, 8192, 8192, 4608, 300, 8192, 6000, 4097, 1024, 8191, 8192, 211, 7000]
    x = synth.white_noise([26], sum(sizes))[0]
    g, r, _ = _run(ge, eo, x, sizes)
    assert np.array_equal(g[:, :5000], np.zeros((2, 5000), np.float32)) and np.array_equal(r[:, :5000], np.zeros((2, 5000)))
    assert_parity(g, r, f"{taps} taps, host blocks of up to 8192 frames")
