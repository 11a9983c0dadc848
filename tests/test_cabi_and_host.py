"""CPU-side checks: the C ABI library loads and exports exactly what include/mapf_hip.h declares,
the product fails loudly without a GPU (no CPU fallback), and the kernels that can be dispatched
compile without register spills.  No compute is launched here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from gym_mapf_amd import _native as nat
from gym_mapf_amd.envs.grid import MapfGrid
from gym_mapf_amd.envs.vec_env import OptimizationCriteria, VecMapfEnv

HEADER = os.path.join(ROOT, 'include', 'mapf_hip.h')
CSRC = os.path.join(ROOT, 'gym-mapf_amd', 'csrc')


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mapf_[a-z_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = nat.load()
    declared = _declared_functions()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(nat.SIGNATURES) == declared          # the ctypes table covers the header, no more, no less
    assert b'gfx950' in lib.mapf_version()


def test_struct_layouts_match_header():
    # sizes the C side checks via struct_size; offsets follow from natural alignment of the field list
    assert ctypes.sizeof(nat.MapfDesc) == 112
    assert nat.MapfDesc.nbr.offset == 40 and nat.MapfDesc.fail_prob.offset == 64 and nat.MapfDesc.stream.offset == 104
    assert ctypes.sizeof(nat.MapfRolloutIO) == 88
    text = open(HEADER).read()
    for name, value in (('MAPF_MAX_AGENTS', nat.MAPF_MAX_AGENTS), ('MAPF_FLAG_DEVICE_PTRS', nat.MAPF_FLAG_DEVICE_PTRS),
                        ('MAPF_FLAG_THREAD_PER_ENV', nat.MAPF_FLAG_THREAD_PER_ENV),
                        ('MAPF_FLAG_LANE_GROUP', nat.MAPF_FLAG_LANE_GROUP), ('MAPF_STEP_AUTO_RESET', nat.MAPF_STEP_AUTO_RESET)):
        m = re.search(r'#define\s+%s\s+(0x[0-9a-fA-F]+|\d+)u?' % name, text)
        assert m and int(m.group(1), 0) == value, name


def test_argument_validation_happens_before_any_device_work():
    lib = nat.load()
    h = ctypes.c_void_p()
    desc = nat.MapfDesc(struct_size=4)
    assert lib.mapf_create(ctypes.byref(desc), ctypes.byref(h)) == nat.MAPF_EINVAL
    assert b'struct_size' in lib.mapf_last_error()
    assert lib.mapf_create(None, ctypes.byref(h)) == nat.MAPF_EINVAL
    assert lib.mapf_step(None, None, None, None, None, None, None, None, None, 0) == nat.MAPF_EINVAL
    assert lib.mapf_destroy(None) == nat.MAPF_EINVAL
    nbr = np.zeros((4, 5), np.uint16) + np.arange(4, dtype=np.uint16)[:, None]
    cells = np.zeros(2, np.uint16)
    desc = nat.MapfDesc(struct_size=ctypes.sizeof(nat.MapfDesc), n_cells=4, n_agents=200, n_envs=1,
                        nbr=nbr.ctypes.data, start=cells.ctypes.data, goal=cells.ctypes.data)
    assert lib.mapf_create(ctypes.byref(desc), ctypes.byref(h)) == nat.MAPF_EUNSUPPORTED
    desc.n_agents = 2
    bad = nbr.copy(); bad[1, 2] = 9
    desc.nbr = bad.ctypes.data
    assert lib.mapf_create(ctypes.byref(desc), ctypes.byref(h)) == nat.MAPF_EINVAL and b'nbr' in lib.mapf_last_error()


def test_no_cpu_fallback_without_a_gpu():
    if nat.device_count() > 0:
        pytest.skip('a GPU is present')
    grid = MapfGrid(['....', '....'])
    with pytest.raises(nat.MapfNativeError) as err:
        VecMapfEnv(grid, 2, ((0, 0), (1, 1)), ((0, 3), (1, 2)), 0.1, -1.0, 1.0, -1.0, OptimizationCriteria.SoC, n_envs=4)
    assert err.value.code == nat.MAPF_ENODEVICE and 'no CPU fallback' in str(err.value)
    from gym_mapf_amd.envs.mapf_env import MapfEnv
    env = MapfEnv(grid, 2, ((0, 0), (1, 1)), ((0, 3), (1, 2)), 0.1, -1.0, 1.0, -1.0, OptimizationCriteria.SoC)
    assert env.s == env.locations_to_state(((0, 0), (1, 1)))       # host-side tables work anywhere
    with pytest.raises(nat.MapfNativeError):
        env.step(0)                                                # ... stepping needs the device


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'gym-mapf_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'mapf_oracle' not in text and 'c_oracle' not in text and 'import philox' not in text, f


def test_dispatched_lane_group_kernels_have_no_register_spills(tmp_path):
    """The lane-group family serves every agent count; its code objects must not spill (spilled
    SGPRs/VGPRs both cost time and were the one place a miscompile was ever observed)."""
    text = ''
    procs = []
    for unit in ('mapf_lg_kernels', 'mapf_lg_rollout', 'mapf_lq_rollout', 'mapf_transitions'):
        out = tmp_path / (unit + '.s')
        procs.append((out, subprocess.Popen(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
                                             '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only',
                                             os.path.join(CSRC, unit + '.hip'), '-o', str(out)], stderr=subprocess.DEVNULL)))
    for out, proc in procs:
        assert proc.wait() == 0
        text += out.read_text()
    kernels = re.findall(r'\.name:\s+(_ZN4mapf\w+)', text)
    assert len(kernels) >= 100
    spills = [int(x) for x in re.findall(r'\.(?:sgpr|vgpr)_spill_count:\s+(\d+)', text)]
    scratch = [int(x) for x in re.findall(r'\.private_segment_fixed_size:\s+(\d+)', text)]
    assert spills and all(v == 0 for v in spills) and all(v == 0 for v in scratch)
