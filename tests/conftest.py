import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_reference():
    return os.path.isdir(os.path.join(REFERENCE, "catfish"))


@pytest.fixture(scope="session")
def ckpt_weights():
    """The 74 inference tensors of ckpnt-30000 (exported fixture, see make_network_golden.py)."""
    with np.load(os.path.join(GOLDEN, "ckpnt-30000-inference.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_read():
    with np.load(os.path.join(GOLDEN, "golden_read_4096_seed0.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def graph_golden():
    """Outputs of the reference's own MetaGraphDef, interpreted in numpy (make_graph_golden.py)."""
    with np.load(os.path.join(GOLDEN, "graph_golden.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def hp():
    return dict(batch_size=256, optimizer_choice="RMSProp", learning_rate=0.001, layer_size=64,
                n_layers=3, keep_prob=0.8, layer_size_res=32, n_layers_res=2)
