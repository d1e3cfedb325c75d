"""Randomised operation sequences: the HIP path (C ABI) against the CPU oracle, call by call.

The fixed-case parity tests pin each kernel path; these sequences mix them the way a host would --
ragged host blocks, impulse responses replaced in mid-stream (per path, different lengths), EQ bands
retuned / toggled / reset between calls, engine clones -- so that every hand-over of state between the
single-launch engine call, the general kernel, the one-partition fast path, the time-parallel long-IR
path and the two EQ kernels gets exercised.  Bars as in test_gpu_parity.py: EQ bit-exact, everything that
passes through an FFT <= 1e-6 RMS (absolute and relative) against the oracle.  Odd seeds run the engine in
real-time mode (resident kernel + mailbox, round 2) and toggle it in mid-sequence, so that the hand-over of
state between the resident kernel and every launched path -- in both directions, around set_ir, clone and the
chain's EQ -- is part of the mix.
"""
import os

import numpy as np
import pytest

from tests.util import assert_parity

pytestmark = pytest.mark.gpu

# OHS_FUZZ_SEEDS=200 python -m pytest tests/test_gpu_fuzz.py -m gpu   for a longer soak
_EXTRA = int(os.environ.get("OHS_FUZZ_SEEDS", "0"))
_FIRST = int(os.environ.get("OHS_FUZZ_FIRST", "0"))      # with OHS_FUZZ_SEEDS: skip seeds below this one


@pytest.fixture(scope="module")
def ohs():
    import open_headstage_amd
    return open_headstage_amd


@pytest.fixture(scope="module")
def synth():
    from open_headstage_amd import synth
    return synth


def _random_ir(rng, taps):
    """Decaying noise burst after a short delay, L1 norm 0.45 (two paths per ear keep |y| <= 0.9 |x|)."""
    if taps == 0:
        return np.zeros(0, np.float32)
    h = rng.standard_normal(taps) * np.exp(-np.arange(taps) / max(taps / 5.0, 1.0))
    h[min(int(rng.integers(0, 40)), taps - 1)] += 2.0
    return (0.45 * h / np.abs(h).sum()).astype(np.float32)


def _random_band(ohs, rng):
    from open_headstage_amd import BandConfig, FilterType
    return BandConfig(FilterType(int(rng.integers(0, 8))), float(np.exp(rng.uniform(np.log(30.0), np.log(18000.0)))),
                      float(rng.uniform(0.3, 3.0)), float(rng.uniform(-6.0, 6.0)), bool(rng.random() < 0.8))


def _sync_band(qg, qo, i):
    c, en = qg.get_band_coeffs(i)          # the oracle is fed the SAME coefficient bits
    qo.set_band_coeffs(i, c, en)


@pytest.mark.parametrize("seed", range(_FIRST if _EXTRA else 0, max(6, _EXTRA)))
def test_fuzz_engine_eq_chain(ohs, synth, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    eg, eo = ohs.ConvolutionEngine.new(), oracle.ConvolutionEngine()
    nb = int(rng.choice([3, 10, 12, 14, 20]))       # ring kernel, conveyor kernel, two passes
    qg, qo = ohs.StereoParametricEQ.new(nb, synth.FS), oracle.StereoParametricEQ(nb, synth.FS)
    for i in range(nb):
        qg.update_band_coeffs(i, synth.FS, _random_band(ohs, rng))
        _sync_band(qg, qo, i)
    for p in range(4):
        h = _random_ir(rng, int(rng.choice([64, 512, 700])))
        eg.set_ir(p, h); eo.set_ir(p, h)
    sizes = [1, 7, 64, 511, 512, 513, 1024, 1024, 1024, 1500, 2048, 3000, 4096, 5000, 8192, 9000]
    stream = synth.white_noise([200 + seed], 45 * 9000)[0]        # 45 steps of <= 9000 frames
    pos = 0
    got, ref = [], []
    rt = bool(seed & 1)
    if rt:
        eg.set_realtime(True)
    for step in range(45):
        if (seed & 1) and step in (13, 27, 36):                   # leave / re-enter real-time mode in mid-stream
            rt = not rt
            eg.set_realtime(rt)
        op = rng.random()
        if op < 0.62:                   # the lib.rs:1169-1207 chain on one host block
            n = int(rng.choice(sizes))
            l, r = stream[0, pos:pos + n].copy(), stream[1, pos:pos + n].copy()
            pos += n
            ol, orr = l.copy(), r.copy()
            eq_on = bool(rng.random() < 0.7)
            gain = float(np.float32(rng.uniform(0.2, 1.0)))
            ohs.process_chain(eg, qg, l, r, eq_enable=eq_on, output_gain=gain)
            oracle.chain_process(eo, qo, ol, orr, eq_enable=eq_on, gain=gain)
            got.append(np.stack([l, r])); ref.append(np.stack([ol, orr]))
            assert np.abs(got[-1] - ref[-1]).max() < 2e-5, (seed, step, n)
        elif op < 0.70:                 # EQ alone: bit-exact
            n = int(rng.choice(sizes))
            l, r = stream[0, pos:pos + n].copy(), stream[1, pos:pos + n].copy()
            pos += n
            ol, orr = l.copy(), r.copy()
            qg.process_block(l, r); qo.process_block(ol, orr)
            assert np.array_equal(l.view(np.uint32), ol.view(np.uint32)), (seed, step, n)
            assert np.array_equal(r.view(np.uint32), orr.view(np.uint32)), (seed, step, n)
        elif op < 0.80:                 # one path gets a new impulse response (set_ir resets that path only)
            p = int(rng.integers(0, 4))
            h = _random_ir(rng, int(rng.choice([0, 1, 100, 512, 513, 1500, 2048, 5000])))   # (several partitions: sums computed ahead)
            eg.set_ir(p, h); eo.set_ir(p, h)
        elif op < 0.90:                 # retune / toggle a band (state is kept: parametric_eq.rs:112)
            i = int(rng.integers(0, nb))
            qg.update_band_coeffs(i, synth.FS, _random_band(ohs, rng))
            _sync_band(qg, qo, i)
        elif op < 0.95:
            qg.reset_all_bands_state(); qo.reset_all_bands_state()
        else:                           # continue on deep copies (a clone starts outside real-time mode)
            eg, eo = eg.clone(), eo.clone()
            if rt:
                eg.set_realtime(True)
    if got:
        assert_parity(np.concatenate(got, 1), np.concatenate(ref, 1), f"fuzz chain seed {seed}")


@pytest.mark.parametrize("seed", range(_FIRST if _EXTRA else 0, max(4, _EXTRA)))
def test_fuzz_batch(ohs, synth, oracle, seed):
    import torch
    rng = np.random.default_rng(2000 + seed)
    S = int(rng.integers(1, 6)) if seed % 5 else int(rng.integers(6, 24))
    bands = synth.eq_table()
    bp = ohs.BatchProcessor(S, num_bands=len(bands))
    eos = [oracle.ConvolutionEngine() for _ in range(S)]
    qos = [oracle.StereoParametricEQ(len(bands), synth.FS) for _ in range(S)]
    shared = {}                         # the shared EQ table as the shared calls have set it: band -> (coeffs, enabled)
    for i, b in enumerate(bands):
        bp.update_band_coeffs(i, synth.FS, b)
        c = ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db)
        shared[i] = (c, b.enabled)
        for q in qos:
            q.set_band_coeffs(i, c, b.enabled)
    long_ir = bool(seed % 2)            # odd seeds: partitions > 1 (sequential, time-parallel and block-2048 kernels)
    lengths = [300, 512, 2000, 5000, 9000] if long_ir else [64, 300, 512]
    cur_irs = [None] * 4
    for p in range(4):
        h = _random_ir(rng, int(rng.choice(lengths)))
        cur_irs[p] = h
        bp.set_ir(p, h)
        for e in eos:
            e.set_ir(p, h)
    x = synth.white_noise(range(300 + 10 * seed, 300 + 10 * seed + S), 512 * 130 * 9)    # 9 calls of <= 130 blocks
    pos = 0
    got, ref = [], []
    for call in range(9):
        if call and rng.random() < 0.35:    # mid-stream: one path replaced -> per-path gating in the general kernel
            p = int(rng.integers(0, 4))
            h = _random_ir(rng, int(rng.choice(lengths + [0])))
            cur_irs[p] = h
            bp.set_ir(p, h)
            for e in eos:
                e.set_ir(p, h)
        if call and rng.random() < 0.15:    # retune one band for every stream (state kept)
            i = int(rng.integers(0, len(bands)))
            nb_ = _random_band(ohs, rng)
            bp.update_band_coeffs(i, synth.FS, nb_)
            c = ohs.biquad_coefficients(nb_.filter_type, synth.FS, nb_.center_freq, nb_.q, nb_.gain_db)
            shared[i] = (c, nb_.enabled)
            for q in qos:
                q.set_band_coeffs(i, c, nb_.enabled)
        if call and rng.random() < 0.3:     # ONE stream gets a band of its own (every plugin instance owns its bands) ...
            s_ = int(rng.integers(0, S))
            i = int(rng.integers(0, len(bands)))
            nb_ = _random_band(ohs, rng)
            c = ohs.biquad_coefficients(nb_.filter_type, synth.FS, nb_.center_freq, nb_.q, nb_.gain_db)
            bp.set_stream_band_coeffs(s_, i, c, nb_.enabled)
            qos[s_].set_band_coeffs(i, c, nb_.enabled)
        if call and rng.random() < 0.08:    # ... and back to the one shared table (what the shared calls have set)
            bp.share_eq_table()
            for q in qos:
                for i, (c, en) in shared.items():
                    q.set_band_coeffs(i, c, en)
        if call and rng.random() < 0.1:     # zero all DSP state, keep IRs and EQ table
            bp.reset()
            for q in qos:
                q.reset_all_bands_state()
            irs_now = cur_irs
            eos = [oracle.ConvolutionEngine() for _ in range(S)]
            for e in eos:
                for p in range(4):
                    e.set_ir(p, irs_now[p])
        eq_on = bool(rng.random() < 0.6)
        gain = float(np.float32(rng.uniform(0.3, 1.0)))
        bp.set_eq_enabled(eq_on); bp.set_gain(gain)
        # the plan of the one-partition path: library's choice, block 512 / FFT 1024, hop 1536 / FFT 2048 -- they hand the
        # same state over, so any sequence of them must stay on the oracle (no effect while a path has more partitions)
        bp.set_conv_plan(int(rng.integers(0, 3)))
        nblk = int(rng.choice([1, 2, 3, 5, 9, 17, 40, 64, 70, 130]))
        n = nblk * 512
        xt = torch.from_numpy(np.ascontiguousarray(x[:, :, pos:pos + n])).cuda()
        mode = int(rng.integers(0, 4))
        if mode == 0:                       # out of place, strict
            y = bp.process(xt).cpu().numpy()
        elif mode == 1:                     # in place
            y = bp.process(xt, out=xt).cpu().numpy()
        elif mode == 2:                     # deferred join (ohs_batch_process_deferred / ohs_batch_join)
            yt = bp.process(xt, deferred=True)
            bp.join()
            y = yt.cpu().numpy()
        else:                               # padded strides through the pointer interface
            pad_c, pad_s = int(rng.integers(0, 3)) * 64, int(rng.integers(0, 3)) * 128
            cs_ = n + pad_c
            ss_ = 2 * cs_ + pad_s
            buf = torch.full((S * ss_,), float("nan"), device="cuda")
            view = torch.as_strided(buf, (S, 2, n), (ss_, cs_, 1))
            view.copy_(xt)
            obuf = torch.zeros_like(buf)
            bp.process_ptr(buf.data_ptr(), obuf.data_ptr(), nblk, ss_, cs_,
                           torch.cuda.current_stream().cuda_stream)
            y = torch.as_strided(obuf, (S, 2, n), (ss_, cs_, 1)).cpu().numpy()
        o = np.empty_like(y)
        for s in range(S):
            l, r = x[s, 0, pos:pos + n].copy(), x[s, 1, pos:pos + n].copy()
            oracle.chain_process(eos[s], qos[s], l, r, eq_enable=eq_on, gain=gain)
            o[s, 0], o[s, 1] = l, r
        pos += n
        assert np.abs(y - o).max() < 2e-5, (seed, call, nblk, mode, eq_on, S, [int(v) for v in np.argwhere(np.abs(y - o) > 2e-5)[:6].ravel()])
        got.append(y); ref.append(o)
    assert_parity(np.concatenate(got, 2), np.concatenate(ref, 2), f"fuzz batch seed {seed}")


@pytest.mark.parametrize("seed", range(_FIRST if _EXTRA else 0, max(4, _EXTRA // 4)))
def test_fuzz_node_batch_slots_against_one_batch(ohs, synth, exp_tuning, seed):
    """Random operation sequences through ohs_node_batch_* with 2 ... 4 device slots (all of them device 0: the
    experiments build's rehearsal mode) against ONE batch holding all the streams, bit for bit: sharding, per-slot
    tables, per-slot worker threads and pointers, the host-buffer form's row offsets, table changes and resets in
    mid-stream.  (The single batch is itself pinned to the oracle by test_fuzz_batch.)"""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7000 + seed)
    L = exp_tuning.lib
    exp_tuning("node_shared_device_rehearsal", "1")
    slots = int(rng.integers(2, 5))
    S = int(rng.integers(slots, 14))
    bands = synth.eq_table()
    coeffs = np.stack([ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db) for b in bands])
    enabled = np.array([b.enabled for b in bands])
    long_ir = bool(seed % 2)
    lengths = [300, 512, 2000, 5000] if long_ir else [64, 300, 512]
    irs = [_random_ir(rng, int(rng.choice(lengths))) for _ in range(4)]
    nb = ohs.NodeBatchProcessor(S, num_bands=len(bands), devices=[0] * slots, library=L)
    bp = ohs.BatchProcessor(S, num_bands=len(bands), library=L)
    nb.set_tables(irs, coeffs, enabled)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i in range(len(bands)):
        bp.set_band_coeffs(i, coeffs[i], bool(enabled[i]))
    shards = [nb.shard(i) for i in range(slots)]
    assert sum(c for _, _, c in shards) == S
    for step in range(14):
        op = rng.random()
        if op < 0.55:
            n = 512 * int(rng.choice([1, 2, 7, 16, 40, 70]))
            x = synth.white_noise_torch(int(rng.integers(0, 1 << 20)), S, n, dev)
            if rng.random() < 0.7:          # device-resident form
                xs = [x[f:f + c].contiguous() for _, f, c in shards]
                ys = [torch.empty_like(a) for a in xs]
                torch.cuda.synchronize()
                nb.process(xs, ys); nb.sync()
                y = torch.cat(ys)
                y_one = bp.process(x)
            else:                           # host-buffer form: the same time chunks on both sides (with partitions > 1 the
                cb = int(rng.choice([0, 4, 16]))    # long-IR kernels round differently across a call boundary than inside)
                y = torch.from_numpy(nb.process_host(x.cpu().numpy(), chunk_blocks=cb)).to(dev)
                y_one = torch.from_numpy(bp.process_host(x.cpu().numpy(), chunk_blocks=cb)).to(dev)
            torch.cuda.synchronize()
            assert torch.equal(y.view(torch.int32), y_one.view(torch.int32)), (seed, step, slots, S, n)
        elif op < 0.70:
            p = int(rng.integers(0, 4))
            h = _random_ir(rng, int(rng.choice([0, 1, 100, 512, 513] + ([1500, 4000] if long_ir else []))))
            nb.set_ir(p, h); bp.set_ir(p, h)
        elif op < 0.78:
            i = int(rng.integers(0, len(bands)))
            c = ohs.biquad_coefficients(*_band_args(_random_band(ohs, rng), synth.FS))
            en = bool(rng.random() < 0.8)
            nb.set_band_coeffs(i, c, en); bp.set_band_coeffs(i, c, en)
        elif op < 0.86:
            on = bool(rng.random() < 0.7)
            nb.set_eq_enabled(on); bp.set_eq_enabled(on)
        elif op < 0.92:
            g = float(np.float32(rng.uniform(0.2, 1.0)))
            nb.set_gain(g); bp.set_gain(g)
        elif op < 0.96:
            plan = int(rng.integers(0, 3))
            nb.set_conv_plan(plan); bp.set_conv_plan(plan)
        else:
            nb.reset(); bp.reset()


def _band_args(b, fs):
    return b.filter_type, fs, b.center_freq, b.q, b.gain_db
