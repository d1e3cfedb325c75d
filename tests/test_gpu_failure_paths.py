"""Error paths of the batch handle.  A HIP failure in the middle of ohs_batch_process leaves the per-stream state advanced
for some time chunks only; the call must join its internal stream, drop the timing spans it opened, report OHS_ERR_HIP,
and the handle must refuse every later processing call until ohs_batch_reset -- never carry on from half-advanced state.
The failure is injected through the EXPERIMENTS build's debug entry point (ohs_debug_inject_batch_failure: the k-th time
chunk from now fails as if HIP had refused a launch); the product library has no such path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(ohs, synth, S, library):
    bp = ohs.BatchProcessor(S, num_bands=10, library=library)
    irs = synth.hrir_set(512)
    for p in range(4):
        bp.set_ir(p, irs[p])
    for i, b in enumerate(synth.eq_table()):
        bp.update_band_coeffs(i, synth.FS, b)
    bp.set_eq_enabled(True)
    return bp


@pytest.mark.parametrize("eq_on,kth", [(True, 1), (True, 3), (True, 6), (False, 1)])
def test_a_failure_in_mid_call_fails_the_handle_until_reset(exp_tuning, eq_on, kth):
    import torch
    import open_headstage_amd as ohs
    from open_headstage_amd import _ffi, synth
    L = exp_tuning.lib
    S, n_blocks = 8, 128                    # >= 64 blocks with the EQ on: six time chunks, EQ || convolution on two streams
    dev = torch.device("cuda:0")
    x = synth.white_noise_torch(3, S, n_blocks * 512, dev)
    bp = _make(ohs, synth, S, L)
    bp.set_eq_enabled(eq_on)
    y_fresh = bp.process(x).clone()         # what a fresh object answers to x
    bp.reset()
    bp.set_profiling(True)
    y = torch.empty_like(x)
    bp.process(x, out=y)                    # one good call: its spans stay readable
    L.ohs_debug_inject_batch_failure(kth)
    with pytest.raises(ohs.OhsError) as ei:
        bp.process(x, out=y)
    assert ei.value.status == _ffi.OHS_ERR_HIP and "injected failure" in str(ei.value)
    torch.cuda.synchronize()                # nothing dangling on either stream
    # refused from now on, through every processing entry point, with the reason
    for call in (lambda: bp.process(x, out=y), lambda: bp.process(x, out=y, deferred=True),
                 lambda: bp.process_host(x.cpu().numpy())):
        with pytest.raises(ohs.OhsError) as ei:
            call()
        assert ei.value.status == _ffi.OHS_ERR_HIP and "half-advanced" in str(ei.value) and "injected failure" in str(ei.value)
    # the timing spans of the failed call are gone, those of the good one are intact
    eq_ms, conv_ms, n_calls, eq_l, conv_l = bp.profile_read()
    assert eq_ms >= 0 and conv_ms > 0 and eq_l == (6 if eq_on else 0) and conv_l == (6 if eq_on else 1)
    bp.set_profiling(False)
    # ohs_batch_reset is the way back: zeroed state, tables kept -- the bits of a fresh object
    bp.reset()
    y2 = bp.process(x).clone()
    torch.cuda.synchronize()
    assert torch.equal(y2.view(torch.int32), y_fresh.view(torch.int32))
    # other handles of the library were never affected
    other = _make(ohs, synth, S, L)
    other.set_eq_enabled(eq_on)
    assert torch.equal(other.process(x).view(torch.int32), y_fresh.view(torch.int32))
