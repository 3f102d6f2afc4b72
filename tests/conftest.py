import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    # materialise once: NpzFile re-decompresses the member on every [] access
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def world_npz():
    return load_golden("world_stock.npz")


def pack_obs_rows(obs):
    """[n, 100] float32 state_PathPlan rows -> [n, 20] int32 packed rows (include/uavenv.h, UAVENV_OBS_PACKED): the test-side
    inverse of uavenv_obs_unpack.  Raises if a row is not packed-representable (a flag column that is not 0 / 1, a non-zero
    pad column)."""
    import numpy as np
    obs = np.ascontiguousarray(obs, dtype=np.float32)
    n = obs.shape[0]
    flag_cols = list(range(11, 86)) + list(range(90, 95))
    f = obs[:, flag_cols]
    if not np.all((f == 0.0) | (f == 1.0)) or np.any(obs[:, 95:] != 0.0):
        raise ValueError("rows are not packed-representable")
    out = np.zeros((n, 20), dtype=np.uint32)
    for c in flag_cols:
        out[:, c >> 5] |= (obs[:, c] != 0).astype(np.uint32) << np.uint32(c & 31)
    out[:, 4:15] = obs[:, 0:11].view(np.uint32)
    out[:, 15:19] = obs[:, 86:90].view(np.uint32)
    return out.view(np.int32)
