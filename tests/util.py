"""Shared helpers for the parity tests."""
import numpy as np

RMS_TOL = 1e-6   # BASELINE.json north_star: <= 1e-6 RMS float error vs the reference CPU path


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0


def rms_err(test, ref):
    """(absolute RMS error, relative RMS error) over all frames / channels."""
    test = np.asarray(test, np.float64)
    ref = np.asarray(ref, np.float64)
    e = rms(test - ref)
    r = rms(ref)
    return e, (e / r if r > 0 else e)


def assert_parity(test, ref, what, tol=RMS_TOL):
    a, r = rms_err(test, ref)
    assert a <= tol and r <= tol, f"{what}: abs RMS {a:.3e}, rel RMS {r:.3e} (tol {tol:.0e})"
    return a, r


def setup_engines(oracle, gpu_engine, irs):
    eo = oracle.ConvolutionEngine()
    for p, h in enumerate(irs):
        eo.set_ir(p, h)
        gpu_engine.set_ir(p, h)
    return eo
