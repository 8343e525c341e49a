import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # -m gpu tests must never silently pass without a device
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """npz fixture with helpers to pull reference-keyed state_dicts (tools/gen_golden.py)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    def __getitem__(self, k):
        return self.z[k]

    def t(self, k):
        return torch.from_numpy(np.asarray(self.z[k]))

    def keys(self):
        return list(self.z.keys())

    def state_dict(self, prefix):
        out = {}
        for k in self.z.keys():
            if k.startswith(prefix):
                out[k[len(prefix):]] = torch.from_numpy(np.asarray(self.z[k]))
        return out


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def oracle():
    from raw_ops import RawOps
    return RawOps("oracle")


@pytest.fixture(scope="session")
def hip():
    from raw_ops import RawOps
    return RawOps("hip")


@pytest.fixture
def tuning(monkeypatch):
    """set(**env): override M355_* tuning knobs for one test.  The library caches them at load time, so the
    environment is re-read now (m355_reload_tuning) and again when the test ends."""
    from segmentation_pipeline_amd import _lib

    def set_(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, str(v))
        _lib.reload_tuning()
    yield set_
    monkeypatch.undo()
    _lib.reload_tuning()
