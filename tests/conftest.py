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


@pytest.fixture(params=["jetnet30", "jetnet150", "cond_gl", "cond_jetclass", "addtime", "addtime_notl", "relu", "noact"])
def golden(request):
    return load_golden(request.param)


class TfGolden:
    """tests/golden/tf_<name>.npz (BASELINE cfg 4 family).  The large fixture does not store the weights: they are
    re-derived from the stored seed (oracle/seeded.py) and checked against the stored |w| sum."""

    FILE = "tf_{}.npz"

    def __init__(self, name):
        from oracle.seeded import seeded_state

        self.name = name
        self.z = np.load(os.path.join(GOLDEN, self.FILE.format(name)), allow_pickle=False)
        self.hp = json.loads(str(self.z["hp_json"]))
        self.keys = [str(k) for k in self.z["_keys"]]
        self.freqs = torch.from_numpy(self.z["freqs"])
        shapes = {k: tuple(s) for k, s in json.loads(str(self.z["_shapes_json"])).items()}
        self.subsampled = "sd/" + next(iter(shapes)) not in self.z.files
        if self.subsampled:
            new = seeded_state(shapes, int(self.z["seed"]))
            assert abs(sum(float(np.abs(v).sum(dtype=np.float64)) for v in new.values()) - float(self.z["abs_sum"])) < 1e-6
            self.state = {k: torch.from_numpy(v) for k, v in new.items()}
        else:
            self.state = {k: torch.from_numpy(self.z["sd/" + k]) for k in shapes}
        self.state["flows.0.frequencies"] = 2 ** torch.arange(self.hp["frequencies"]) * torch.pi

    get = Golden.get
    grads = Golden.grads

    def pick(self, g):
        """Bring a full gradient to what the fixture stores."""
        from oracle.seeded import subsample

        return torch.from_numpy(subsample(g.detach().cpu().numpy())) if self.subsampled else g.detach().cpu()


class WideGolden(TfGolden):
    """tests/golden/epicw_<name>.npz: EPiC at the JetClass width (BASELINE cfg 5), weights re-derived from the seed."""

    FILE = "epicw_{}.npz"


def load_wide_golden(name):
    key = "epicw_" + name
    if key not in _cache:
        _cache[key] = WideGolden(name)
    return _cache[key]


@pytest.fixture(params=["small", "jetclass", "sincos", "lhco128", "plain", "plainw"])
def wide_golden(request):
    return load_wide_golden(request.param)


class EpicSeededGolden(TfGolden):
    """tests/golden/epic_<name>.npz in the seed-derived format (weights re-derived from the stored seed): jet-resident EPiC configurations
    recorded by the generic recorders of oracle/make_golden.py."""

    FILE = "epic_{}.npz"


def load_epic_seeded_golden(name):
    key = "epics_" + name
    if key not in _cache:
        _cache[key] = EpicSeededGolden(name)
    return _cache[key]


class CaGolden(TfGolden):
    """tests/golden/ca_<name>.npz: the cross-attention encoder (model "droid_fullcrossattention"), seed-derived weights."""

    FILE = "ca_{}.npz"


def load_ca_golden(name):
    key = "ca_" + name
    if key not in _cache:
        _cache[key] = CaGolden(name)
    return _cache[key]


@pytest.fixture(params=["small", "lhco", "plain"])
def ca_golden(request):
    return load_ca_golden(request.param)


class MdmaGolden(TfGolden):
    """tests/golden/mdma_<name>.npz: the MDMA model (model "mdma"), seed-derived weights."""

    FILE = "mdma_{}.npz"


def load_mdma_golden(name):
    key = "mdma_" + name
    if key not in _cache:
        _cache[key] = MdmaGolden(name)
    return _cache[key]


@pytest.fixture(params=["small", "yaml", "tcat", "tloc", "tglob", "cond", "condcat", "lcat"])
def mdma_golden(request):
    return load_mdma_golden(request.param)


def load_tf_golden(name):
    key = "tf_" + name
    if key not in _cache:
        _cache[key] = TfGolden(name)
    return _cache[key]


@pytest.fixture(params=["small", "lhco", "sincos", "plain"])
def tf_golden(request):
    return load_tf_golden(request.param)
