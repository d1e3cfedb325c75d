"""A per-path set_ir in mid-stream on the long-IR batch path (BASELINE configs[3] shape: 64 streams, 4 x 16384 taps =
32 partitions, 256-block calls).

The reference resets only THAT path's history (convolution.rs:135-138), the other three keep seeing the blocks
processed before.  Until round 4 the call after such a set_ir went through k_conv_general as a whole (one wave per
stream, blocks in sequence: ~290 x the steady call).  Now the time-parallel kernels take it: the old blocks are hidden
behind a zeroed gap of the ring and k_conv_tp_old adds, under the per-path gate, what each path may still see of them.
Checked: parity with the oracle across several resets (longer, shorter, muted), and the cost of the call after a reset.

Round 5: the steady call of this shape is served by the large-block plans (block 8192 out of place: 0.12 ms, block 2048 in place:
0.135, instead of 0.25), and an object that keeps
the input history (every batch object with a response of two partitions or more) carries a per-path set_ir out as "every
path forgets its past" + pending tails -- what the frames in front of the reset still owe the output through the surviving
paths, computed once at set_ir time and added to the calls that follow (api_conv.hip: conv_lb_tail_route).  The call
behind the reset is block 2048 as well: 0.16-0.175 ms = 1.3 x the steady call.  The gated block-512 kernels of round 4
still serve objects told to keep the reference blocking (ohs_batch_set_conv_plan(1)) and calls whose buffers the block-2048
kernels cannot address (odd strides, 4-byte alignment).
"""
import numpy as np
import pytest

from tests.util import assert_parity

pytestmark = pytest.mark.gpu

S, TAPS, NB = 64, 16384, 256
CHECK = (0, 31, 63)         # streams compared with the oracle (the oracle runs 32 partitions x 4 paths per block)


def _irs(synth):
    return synth.hrir_set(TAPS)


def _edits(synth):
    """(call index, path, new impulse response): applied before that call"""
    return [
        (1, 1, synth.hrir_set(12000)[1]),                   # one path, shorter than the others (P = 24)
        (2, 2, np.zeros(0, np.float32)),                    # mute (one zero partition)
        (2, 0, synth.hrir_set(700)[0]),                     # two partitions
        (3, 3, synth.hrir_set(TAPS)[0] * np.float32(0.5)),  # full length again
    ]


def test_midstream_set_ir_on_the_long_ir_path_matches_the_oracle(oracle):
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    dev = torch.device("cuda:0")
    irs = _irs(synth)
    bp = ohs.BatchProcessor(S, num_bands=10)
    engs = {s: oracle.ConvolutionEngine() for s in CHECK}
    for p in range(4):
        bp.set_ir(p, irs[p])
        for e in engs.values():
            e.set_ir(p, irs[p])
    calls = 4
    frames = NB * 512
    got = {s: [] for s in CHECK}
    ref = {s: [] for s in CHECK}
    for c in range(calls):
        for (at, path, h) in _edits(synth):
            if at == c:
                bp.set_ir(path, h)
                for e in engs.values():
                    e.set_ir(path, h)
        x = synth.white_noise_torch(1000 * c, S, frames, dev)
        y = torch.empty_like(x)
        bp.process(x, out=y)
        torch.cuda.synchronize()
        xc, yc = x.cpu().numpy(), y.cpu().numpy()
        for s in CHECK:
            got[s].append(yc[s])
            ref[s].append(np.stack(engs[s].process_block(xc[s, 0], xc[s, 1])))
    for s in CHECK:
        g, r = np.concatenate(got[s], axis=1), np.concatenate(ref[s], axis=1)
        assert_parity(g, r, f"stream {s} over {calls} calls with per-path set_ir between them")
        # and call by call, so that an error confined to the blocks after a reset cannot hide in the average
        for c in range(calls):
            assert_parity(got[s][c][:, :40 * 512], ref[s][c][:, :40 * 512], f"stream {s}, first 40 blocks of call {c}")


def test_the_call_after_a_per_path_set_ir_costs_at_most_three_steady_calls():
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    dev = torch.device("cuda:0")
    irs = _irs(synth)
    bp = ohs.BatchProcessor(S, num_bands=10)
    for p in range(4):
        bp.set_ir(p, irs[p])
    x = synth.white_noise_torch(5, S, NB * 512, dev)
    y = torch.empty_like(x)

    def timed_call():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bp.process(x, out=y)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    # (one call under the block-512 plan and one reset first, so that every buffer either route may want exists and no timed
    # call pays for an allocation)
    bp.set_conv_plan(1)
    timed_call()
    bp.set_conv_plan(0)
    bp.set_ir(0, irs[0])
    timed_call()
    for _ in range(3):
        timed_call()
    steady = sorted(timed_call() for _ in range(7))[3]
    assert bp.last_conv_plan()[0] == "block8192"        # (out of place, 256 blocks: the single-kernel plan; in place: block 2048)
    after = []
    for k in range(6):
        bp.set_ir(k % 4, irs[(k + 1) % 4])
        after.append(timed_call())
        timed_call()                    # (back to the gate-free plan before the next reset)
    worst = max(after)
    print(f"steady call {steady:.3f} ms; call after a per-path set_ir: {['%.3f' % v for v in after]} ms "
          f"(worst {worst / steady:.2f} x)")
    assert worst <= 2.0 * steady, (steady, after)


@pytest.mark.parametrize("plan", [1, 0])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gated_time_parallel_calls_small_shapes_vs_oracle(oracle, seed, plan):
    """Small, odd shapes: per-path partition counts that differ, calls of exactly Pmax blocks (the shortest the gated
    time-parallel kernels take), shorter ones in between, resets before the previous reset has aged out, a ring that
    wraps.  plan 1: the reference blocking with the per-path gate; plan 0: what the library picks (block 2048 + pending
    tails while some response has two partitions or more)."""
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import synth
    rng = np.random.default_rng(seed)
    Sx = 3
    taps = [int(rng.integers(513, 5200)) for _ in range(4)]
    irs = [synth.hrir_set(t)[p] for p, t in enumerate(taps)]
    bp = ohs.BatchProcessor(Sx, num_bands=10)
    bp.set_conv_plan(plan)
    engs = [oracle.ConvolutionEngine() for _ in range(Sx)]
    for p in range(4):
        bp.set_ir(p, irs[p])
        for e in engs:
            e.set_ir(p, irs[p])
    pmax = lambda: max(int(bp_P[p]) for p in range(4))          # noqa: E731
    bp_P = [(t + 511) // 512 for t in taps]
    got, ref, served = [], [], set()
    for step in range(9):
        if step % 2 == 1:       # a per-path reset before every other call
            path = int(rng.integers(0, 4))
            t_new = int(rng.integers(0, 5200))
            h = synth.hrir_set(max(t_new, 1))[path][:t_new]
            bp.set_ir(path, h)
            for e in engs:
                e.set_ir(path, h)
            bp_P[path] = max(1, (t_new + 511) // 512)
        n = [pmax(), pmax() + 1, 3, 2 * pmax() + 5, 1][int(rng.integers(0, 5))]
        x = synth.white_noise(range(100 * step, 100 * step + Sx), n * 512)
        got.append(bp.process(torch.from_numpy(x).cuda()).cpu().numpy())
        served.add(bp.last_conv_plan()[0])
        ref.append(np.stack([np.stack(engs[s].process_block(x[s, 0], x[s, 1])) for s in range(Sx)]))
    assert ("block2048" in served) == (plan == 0), served
    g, r = np.concatenate(got, axis=2), np.concatenate(ref, axis=2)
    for s in range(Sx):
        assert_parity(g[s], r[s], f"seed {seed} stream {s}")
