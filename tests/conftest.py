"""Shared test plumbing: path setup, the ``gpu`` marker, golden loaders."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (os.path.join(ROOT, 'gym-mapf_amd'), os.path.join(ROOT, 'oracle'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

try:    # torch wheels bundle their own HIP runtime: it has to be in the process BEFORE libmapf_hip.so pulls in the system one,
    import torch  # noqa: F401    # or a later device-mode test reports "No HIP GPUs" -- whichever test file happens to run first
except ImportError:
    pass

TRAJECTORY_SETS = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith('.npz'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def set_tune(monkeypatch, **kv):
    """MAPF_TUNE -- the library's ONE kernel-dispatch override ("key=value,...", read when a handle is created; keys in
    include/mapf_hip.h): merge `kv` into the current string for the rest of the test; a value of None removes the key."""
    cur = dict(item.split('=', 1) for item in os.environ.get('MAPF_TUNE', '').split(',') if item)
    for k, v in kv.items():
        if v is None:
            cur.pop(k, None)
        else:
            cur[k] = str(v)
    if cur:
        monkeypatch.setenv('MAPF_TUNE', ','.join('%s=%s' % item for item in cur.items()))
    else:
        monkeypatch.delenv('MAPF_TUNE', raising=False)


def load_trajectory_set(name):
    with open(os.path.join(GOLDEN, name + '.json')) as f:
        meta = json.load(f)
    data = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    return meta, data


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(params=TRAJECTORY_SETS)
def trajectory_set(request):
    return load_trajectory_set(request.param)
