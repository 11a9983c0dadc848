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

TRAJECTORY_SETS = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith('.npz'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


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
