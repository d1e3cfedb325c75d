"""The block-8192 / FFT-16384 kernel for long out-of-place calls on long impulse responses (csrc/conv_xb_kernels.hip: the library's
choice for responses of 2 .. 32 partitions of 512 taps on calls of 128 blocks or more -- with more than 16 partitions on 32 streams
or more), and its sixteen-wave first form (csrc/experiments/conv_xb16_kernels.hip, experiments build, Tuning::xb_form = 1), checked
against the oracle -- the restatement of the reference's uniform block-512 partitioning
(src/dsp/convolution.rs:120-132,236-289) -- and against f64 direct convolution: one and two partitions of 8192 taps, calls that
end inside an 8192-frame block, calls that follow calls served by the other plans (the state at rest is the input history all of
them share), a per-path set_ir in mid-stream, paths of different lengths and a muted one."""
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


_LIB = {"library": None}


@pytest.fixture(autouse=True, params=["product", "sixteen_waves"])
def _form(exp_tuning, request):
    """product: libohs_hip.so as shipped; sixteen_waves: the experiments library with the first form selected.  Both with the
    two-partition kernel allowed on any number of streams (the tests use 2 .. 3)."""
    from open_headstage_amd import _ffi
    if request.param == "product":
        _LIB["library"] = None
    else:
        _LIB["library"] = _ffi.experiments_lib()
        exp_tuning("xb_form", 1)
        exp_tuning("xb_min_streams_p2", 1)
    yield
    _LIB["library"] = None


def _make(ohs, S, irs, plan=0, gain=1.0):
    bp = ohs.BatchProcessor(S, num_bands=10, library=_LIB["library"])
    for p in range(4):
        bp.set_ir(p, irs[p])
    bp.set_eq_enabled(False)
    bp.set_gain(gain)
    bp.set_conv_plan(plan)
    return bp


def _streams(taps):
    """the library takes the kernel for two partitions of 8192 taps from 32 streams on: 34 streams that repeat three signals
    (the experiments library of the sixteen-wave form has that threshold lowered: 3 streams do)"""
    return 34 if taps > 8192 and _LIB["library"] is None else 3


def _signals(synth, S, first_id, frames):
    base = synth.white_noise(range(first_id, first_id + 3), frames)
    return base, np.ascontiguousarray(base[np.arange(S) % 3])


@pytest.mark.parametrize("taps,blocks", [(16384, [144, 3, 131, 16, 160]), (9000, [129, 128]), (8192, [130, 1, 177]), (6700, [128, 200]),
                                         (1024, [128, 7, 150])])
def test_block_8192_kernel_matches_the_oracle_and_f64(oracle, taps, blocks):
    """long calls take the block-8192 kernel, short ones in between the block-2048 plan: each continues from the input history the
    other leaves; call lengths that end inside an 8192-frame block (zeros behind the call, nothing stored there)"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S = _streams(taps)
    irs = synth.hrir_set(taps)
    bp = _make(ohs, S, irs, 0, gain=0.9)
    engs = _oracle_engines(oracle, irs, 3)
    total = sum(blocks) * 512
    base, x = _signals(synth, S, 0, total)
    got, pos = [], 0
    for k, nb in enumerate(blocks):
        xt = torch.from_numpy(np.ascontiguousarray(x[:, :, pos:pos + nb * 512])).cuda()
        y = bp.process(xt)
        assert bp.last_conv_plan()[0] == ("block8192" if nb >= 128 else "block2048"), (k, bp.last_conv_plan())
        got.append(y.cpu().numpy())
        pos += nb * 512
    g = np.concatenate(got, axis=2)
    for s in range(3, S):
        assert np.array_equal(g[s], g[s % 3]), f"stream {s} differs from stream {s % 3}"
    for s in range(3):
        ref = np.stack(engs[s].process_block(base[s, 0], base[s, 1])) * np.float32(0.9)
        assert_parity(g[s], ref, f"taps {taps}, stream {s}")
    yl, yr = oracle.binaural_f64(base[0, 0], base[0, 1], irs)
    a, r = rms_err(g[0], 0.9 * np.stack([yl, yr]))
    print(f"block-8192 kernel vs f64 direct convolution, {taps} taps: {r:.2e}")
    assert r <= 1e-6


def test_what_the_library_does_not_give_the_kernel(oracle):
    """in-place calls (a run's first windows read frames another workgroup's blocks would overwrite), calls shorter than 128
    blocks, two partitions of 8192 taps on fewer than 32 streams (block 2048 is faster there), more than two partitions"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    if _LIB["library"] is not None:
        pytest.skip("the product's rule")
    for taps, S, nb, in_place, want in ((16384, 34, 160, True, "block2048"), (16384, 34, 127, False, "block2048"),
                                        (16384, 8, 160, False, "block2048"), (16385, 34, 160, False, "block2048"),
                                        (8192, 2, 160, False, "block8192"), (16384, 32, 128, False, "block8192")):
        irs = synth.hrir_set(taps)
        bp = _make(ohs, S, irs)
        x = torch.from_numpy(synth.white_noise(range(S), nb * 512)).cuda()
        if in_place:
            bp.process(x, out=x)
        else:
            bp.process(x)
        assert bp.last_conv_plan()[0] == want, (taps, S, nb, in_place, bp.last_conv_plan())


def test_different_lengths_a_muted_path_and_a_midstream_set_ir(oracle):
    """paths of 16384 / 700 / 0 / 9000 taps; then path 1 is replaced in mid-stream: the pending tails carry the other paths' past
    over the reset, whatever kernel serves the calls behind it"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    S = _streams(16384)
    hb = synth.hrir_set(16384)
    irs = [hb[0], hb[1][:700], hb[2][:0], hb[3][:9000]]
    bp = _make(ohs, S, irs)
    engs = _oracle_engines(oracle, irs, 3)
    base, x = _signals(synth, S, 40, 3 * 140 * 512)
    got, ref = [], []
    for k in range(3):
        if k == 1:
            h = synth.hrir_set(12000)[1]
            bp.set_ir(1, h)
            for e in engs:
                e.set_ir(1, h)
        sl = slice(k * 140 * 512, (k + 1) * 140 * 512)
        got.append(bp.process(torch.from_numpy(np.ascontiguousarray(x[:, :, sl])).cuda()).cpu().numpy())
        assert bp.last_conv_plan()[0] == "block8192", (k, bp.last_conv_plan())
        ref.append(np.stack([np.stack(engs[s].process_block(base[s, 0, sl].copy(), base[s, 1, sl].copy())) for s in range(3)]))
    g, r = np.concatenate(got, axis=2), np.concatenate(ref, axis=2)
    for s in range(3, S):
        assert np.array_equal(g[s], g[s % 3]), f"stream {s} differs from stream {s % 3}"
    for s in range(3):
        assert_parity(g[s], r[s], f"stream {s}")


def test_many_streams_every_run_length(oracle):
    """70 streams (not a multiple of the 8 XCDs) x 19 blocks of 8192: runs of several blocks per workgroup, the last one short;
    every stream against the stream it repeats, three against the oracle"""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    dev = torch.device("cuda:0")
    S, nb = 70, 19 * 16 - 5
    irs = synth.hrir_set(16384)
    bp = _make(ohs, S, irs)
    base = synth.white_noise_torch(77, 3, nb * 512, dev)
    x = base[torch.arange(S, device=dev) % 3].contiguous()
    y = bp.process(x)
    torch.cuda.synchronize()
    assert bp.last_conv_plan()[0] == "block8192"
    for s in range(3, S):
        assert torch.equal(y[s], y[s % 3]), f"stream {s} differs from stream {s % 3}"
    engs = _oracle_engines(oracle, irs, 3)
    xb = base.cpu().numpy()
    for s in range(3):
        assert_parity(y[s].cpu().numpy(), np.stack(engs[s].process_block(xb[s, 0], xb[s, 1])), f"stream {s}")
