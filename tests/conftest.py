import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Returns (state_dict of torch tensors, dict of other arrays, kwargs dict or None)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    state, arrays = {}, {}
    for k in z.files:
        if k.startswith("w::"):
            state[k[3:]] = torch.from_numpy(z[k])
        else:
            arrays[k] = z[k]
    kwargs = json.loads(str(arrays.pop("kwargs"))) if "kwargs" in arrays else None
    return state, arrays, kwargs


def onehot(pos, n):
    x = torch.zeros((len(pos), n), dtype=torch.float32)
    x[torch.arange(len(pos)), torch.as_tensor(np.asarray(pos), dtype=torch.long)] = 1.0
    return x


@pytest.fixture(scope="session")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def kernel_option(gpu):
    """set(name, value): ncf_set_option for the duration of one test (kernel-variant overrides; restored to "auto")."""
    from deeprecommendation_amd import native
    touched = []

    def set_(name, value):
        touched.append(name)
        native.set_option(name, "auto" if value is None else value)

    yield set_
    for name in touched:
        native.set_option(name, "auto")
