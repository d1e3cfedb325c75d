"""The batch mode over the GPUs of a node behind the C ABI (ohs_node_batch_*): on this one-GPU box the RCCL
communicator has size 1 -- ncclCommInitAll, the grouped ncclBroadcast of the packed tables and the unpacking on the
device all run -- and the results must be bit-identical to the single-device batch (ohs_batch_process_host) fed the
same tables: streams do not interact, so sharding changes nothing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ohs():
    import open_headstage_amd
    return open_headstage_amd


@pytest.fixture(scope="module")
def synth():
    from open_headstage_amd import synth as s
    return s


def _tables(ohs, synth, taps):
    irs = synth.hrir_set(taps)
    bands = synth.eq_table()
    coeffs = np.stack([ohs.biquad_coefficients(b.filter_type, synth.FS, b.center_freq, b.q, b.gain_db) for b in bands])
    enabled = np.array([b.enabled for b in bands])
    return irs, coeffs, enabled


@pytest.mark.parametrize("taps", [512, 1300])
def test_node_batch_one_device_bit_identical_to_the_device_batch(ohs, synth, oracle, taps):
    S, n_blocks = 7, 96
    irs, coeffs, enabled = _tables(ohs, synth, taps)
    nb = ohs.NodeBatchProcessor(S, num_bands=len(coeffs), n_devices=1)
    assert nb.shard(0) == (0, 0, S)
    assert nb.rccl_info() == (1, True)              # a real communicator, from the loaded librccl.so
    nb.set_tables(irs, coeffs, enabled)
    nb.set_eq_enabled(True)
    nb.set_gain(0.7)
    x = synth.white_noise(range(S), n_blocks * 512)
    y_node = nb.process_host(x.copy(), chunk_blocks=32)
    bp = ohs.BatchProcessor(S, num_bands=len(coeffs))
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i in range(len(coeffs)):
        bp.set_band_coeffs(i, coeffs[i], bool(enabled[i]))
    bp.set_eq_enabled(True)
    bp.set_gain(0.7)
    y_one = bp.process_host(x.copy(), chunk_blocks=32)
    assert np.array_equal(y_node.view(np.uint32), y_one.view(np.uint32))
    # and within the parity bar of the oracle (stream 3)
    eng = oracle.ConvolutionEngine()
    for p in range(4):
        eng.set_ir(p, irs[p])
    eq = oracle.StereoParametricEQ(len(coeffs), synth.FS)
    for i in range(len(coeffs)):
        eq.set_band_coeffs(i, coeffs[i], bool(enabled[i]))
    l, r = x[3, 0].copy(), x[3, 1].copy()
    oracle.chain_process(eng, eq, l, r, eq_enable=True, gain=0.7)
    ref = np.stack([l, r]).astype(np.float64)
    err = np.sqrt(np.mean((y_node[3] - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
    assert err <= 1e-6, err
    # a second call continues the streams; one path changed in mid-stream through the broadcast path
    nb.set_ir(2, irs[2][:200])
    bp.set_ir(2, irs[2][:200])
    x2 = synth.white_noise(range(100, 100 + S), n_blocks * 512)
    assert np.array_equal(nb.process_host(x2.copy(), chunk_blocks=32).view(np.uint32),
                          bp.process_host(x2.copy(), chunk_blocks=32).view(np.uint32))
    # reset returns both to the same fresh state (tables kept: path 2 stays the shortened response)
    nb.reset(); bp.reset()
    y3 = nb.process_host(x.copy(), chunk_blocks=32)
    assert np.array_equal(y3.view(np.uint32), bp.process_host(x.copy(), chunk_blocks=32).view(np.uint32))
    assert not np.array_equal(y3, y_one)


def test_node_batch_argument_checks(ohs):
    from open_headstage_amd import _ffi
    with pytest.raises(ohs.OhsError) as ei:
        ohs.NodeBatchProcessor(1, n_devices=2)          # fewer streams than devices
    assert ei.value.status == _ffi.OHS_ERR_INVALID_ARG
    with pytest.raises(ohs.OhsError):
        ohs.NodeBatchProcessor(8, devices=[0, 0])       # a device listed twice
    with pytest.raises(ohs.OhsError):
        ohs.NodeBatchProcessor(8, devices=[99])         # no such device
