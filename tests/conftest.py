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


class Golden:
    """One tests/golden/epic_<name>.npz produced by oracle/make_golden.py from the reference."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, f"epic_{name}.npz"), allow_pickle=False)
        self.hp = json.loads(str(self.z["hp_json"]))
        self.keys = [str(k) for k in self.z["_keys"]]
        self.state = {k: torch.from_numpy(self.z["sd/" + k]) for k in self.keys}
        # frequency table of the machine that recorded the vectors (see oracle/fm_ref.py::cosine_encoding)
        self.freqs = torch.from_numpy(self.z["freqs"])

    def get(self, key, default=None):
        if key in self.z.files:
            return torch.from_numpy(self.z[key])
        return default

    def grads(self, tag):
        pre = tag + "grad/"
        return {k[len(pre):]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith(pre)}


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(params=["jetnet30", "jetnet150", "cond_gl", "cond_jetclass"])
def golden(request):
    return load_golden(request.param)
