import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights():
    """Folded fp32 synthetic weights (seed 0) keyed by reference state-dict names."""
    from detail_tts_amd.weights import synthetic_state_dict, select_inference_params
    return select_inference_params(synthetic_state_dict(0))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


_TOL_SCALE = None


def _tol_scale():
    """The tight gates below were measured on ONE platform (MI355X / gfx950, ROCm 7.x: profiles/r05_measured_errors.txt).  They sit near
    fp32 rounding, so another ROCm's expf / tanh or another CU count may move the measurement with no regression behind it:
    elsewhere the limits are widened tenfold (still 2 - 4 orders below north_star's 1e-3), and DTTS_TOL_SCALE overrides either way."""
    global _TOL_SCALE
    if _TOL_SCALE is None:
        env = os.environ.get("DTTS_TOL_SCALE")
        if env:
            _TOL_SCALE = float(env)
        else:
            scale = 1.0
            try:
                import torch
                if torch.cuda.is_available():
                    arch = getattr(torch.cuda.get_device_properties(0), "gcnArchName", "")
                    hip = getattr(torch.version, "hip", None) or ""
                    if not (arch.startswith("gfx950") and hip.split(".")[0] == "7"):
                        scale = 10.0
            except Exception:
                pass
            _TOL_SCALE = scale
    return _TOL_SCALE


def tol(name, value, limit):
    """assert value < limit, and append (name, value, limit) to $DTTS_TEST_LOG when set: the limits of the waveform-level gates are kept at
    ~20 x what is measured on the MI355X (a limit 5 orders above the measurement proves nothing), so the measurements are
    recorded run by run (profiles/r05_measured_errors.txt).  Off the platform the limits were measured on they are scaled (_tol_scale)."""
    value, limit = float(value), float(limit) * _tol_scale()
    log = os.environ.get("DTTS_TEST_LOG")
    if log:
        with open(log, "a") as f:
            f.write(f"{name}\t{value:.3e}\t{limit:.1e}\n")
    assert value < limit, (name, value, limit)
    return value
