import glob
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


def golden_names(prefix=None, exclude_prefix=None):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    if prefix is None:
        names = [n for n in names if not n.startswith(("rlgr_", "pipeline_", "voxres_"))]   # have their own tests
    if prefix is not None:
        names = [n for n in names if n.startswith(prefix)]
    if exclude_prefix is not None:
        names = [n for n in names if not n.startswith(exclude_prefix)]
    return names


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: z[k] for k in z.files}
    if "level_len" in g:                      # unpack the concatenated reference lists
        ll = g["level_len"].astype(np.int64)
        tot = int(ll.sum())
        flags = np.unpackbits(g["flags_cat"])[:tot].astype(bool)
        off = np.concatenate([[0], np.cumsum(ll)])
        g["List"] = [g["list_cat"][off[i]:off[i + 1]].astype(np.int64) for i in range(len(ll))]
        g["Flags"] = [flags[off[i]:off[i + 1]] for i in range(len(ll))]
        g["weights"] = [g["weights_cat"][off[i]:off[i + 1]].astype(np.int64) for i in range(len(ll))]
    return g


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc
