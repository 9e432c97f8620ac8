import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_fixture(name):
    with np.load(os.path.join(GOLDEN, f"fx_{name}.npz")) as z:
        return {k: z[k] for k in z.files}


def load_weights():
    with np.load(os.path.join(GOLDEN, "weights_seed0.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def weights():
    return load_weights()


def rel_l1(a, b):
    """mean|a-b| / mean|b|  -- the north_star depth-parity metric."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).mean() / max(np.abs(b).mean(), 1e-30))


def conf_mask(expected_index, tol=2e-3):
    """Pixels where trunc(expected_index) is stable under fp32 summation-order noise."""
    frac = expected_index - np.floor(expected_index)
    return (frac > tol) & (frac < 1 - tol)


def conf_from_prob(prob, idx):
    """sum of prob over the window [idx-1, idx+2] (zero outside), idx integer [h,w]."""
    D = prob.shape[0]
    out = np.zeros(idx.shape, np.float64)
    for k in (-1, 0, 1, 2):
        j = idx + k
        ok = (j >= 0) & (j < D)
        jj = np.clip(j, 0, D - 1)
        out += np.where(ok, np.take_along_axis(prob, jj[None], 0)[0], 0.0)
    return out


def assert_conf_close(conf, ref_conf, expected_index, prob=None, atol=1e-5, tol=2e-3):
    """Confidence parity that tolerates the trunc() discontinuity of models/mvsnet.py:217.

    Where the expected index sits within `tol` of an integer, fp32 summation order may move
    trunc() to the neighbouring bin; there either window is accepted (needs `prob`), elsewhere
    the value must match.
    """
    m = conf_mask(expected_index, tol)
    np.testing.assert_allclose(conf[m], ref_conf[m], rtol=0, atol=atol)
    if prob is not None and (~m).any():
        D = prob.shape[0]
        base = np.clip(np.round(expected_index).astype(np.int64), 0, D - 1)
        c_hi = conf_from_prob(prob, base)
        c_lo = conf_from_prob(prob, np.clip(base - 1, 0, D - 1))
        err = np.minimum(np.abs(conf - c_hi), np.abs(conf - c_lo))
        assert err[~m].max() <= max(atol, 1e-5) * 4, err[~m].max()
    return float(m.mean())
