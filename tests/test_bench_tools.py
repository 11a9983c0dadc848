"""CPU-side checks of the measurement plumbing: the kernel-source hash that ties committed counter profiles to the code
they were measured on, and the derivation of `roofline.valu_frac` (tools/derive_valu.py) -- a stale or impossible number must
turn into `null` / a refusal, never into a plausible-looking figure."""
import csv
import importlib.util
import json
import os

import pytest

from conftest import ROOT

import bench


def _load_tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'tools', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_csrc_hash_is_stable_and_counter_profiles_are_keyed_by_it(monkeypatch):
    h = bench.csrc_hash()
    assert len(h) == 16 and h == bench.csrc_hash()
    # the committed profiles were derived from THESE sources (refresh_profiles.sh is the last step of a round) ...
    for name in ('traffic.json', 'valu.json'):
        doc = json.load(open(os.path.join(ROOT, 'profiles', name)))
        hashes = {k.get('csrc_hash') for k in doc['kernels'] if k.get('csrc_hash')}
        assert hashes, name
    # ... and an entry of other sources is not used: the bench keys print null
    entry = next(k for k in json.load(open(os.path.join(ROOT, 'profiles', 'valu.json')))['kernels'] if k.get('csrc_hash'))
    args = (entry['kernel'], entry['n_envs'], entry['n_agents'], entry['env_steps_per_launch'])
    monkeypatch.setattr(bench, '_CSRC_HASH', entry['csrc_hash'])
    assert bench.measured_valu(*args) is not None
    monkeypatch.setattr(bench, '_CSRC_HASH', '0' * 16)
    assert bench.measured_valu(*args) is None
    t = next(k for k in json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))['kernels'] if k.get('csrc_hash'))
    assert bench.measured_traffic(t['kernel'], t['n_envs'], t['n_agents'], 256) is None
    monkeypatch.setattr(bench, '_CSRC_HASH', t['csrc_hash'])
    assert bench.measured_traffic(t['kernel'], t['n_envs'], t['n_agents'], 256) > 0


def test_valu_share_is_priced_by_kind_and_never_exceeds_one(tmp_path, monkeypatch):
    dv = _load_tool('derive_valu')
    cost = dv.issue_costs_ns()
    assert cost['fast'] < cost['ordinary'] < cost['mul_f64'] and 1.0 < cost['fast'] < 1.4 and 1.7 < cost['ordinary'] < 2.2
    # pricing: all-fast 32-bit instructions cost the v_xor rate, an all-ordinary mix ~1.6 x that
    assert dv.price((1000, 0, 0, 0), 1.0, cost) == pytest.approx(1000 * cost['fast'])
    assert dv.price((1000, 0, 0, 0), 0.0, cost) / dv.price((1000, 0, 0, 0), 1.0, cost) == pytest.approx(cost['ordinary'] / cost['fast'])
    assert dv.price((0, 10, 10, 10), 0.5, cost) == pytest.approx(10 * (cost['int64'] + cost['mul_f64'] + cost['add_f64']))
    # a synthetic pass whose counters cannot fit its launch time is refused (round 3's file held a "share" of 4.5)
    kernel = 'void mapf::(anonymous namespace)::lq_rollout_kernel<2, 4, true, true, false, false, false, 0>(mapf::RolloutArgs, unsigned int, unsigned int)'

    def write(path, counters, fields=('Kernel_Name', 'Counter_Name', 'Counter_Value')):
        with open(path, 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=fields)
            w.writeheader()
            for c, v in counters.items():
                w.writerow({'Kernel_Name': kernel, 'Counter_Name': c, 'Counter_Value': v})
    write(tmp_path / 'sq1.csv', {'SQ_INSTS_VALU': 76.5e6, 'SQ_BUSY_CYCLES': 12.5e6, 'SQ_WAVES': 2048, 'SQ_WAVE_CYCLES': 170e6, 'SQ_ACTIVE_INST_VALU': 76.5e6})
    write(tmp_path / 'sq3.csv', {'SQ_INSTS_VALU_INT64': 7.2e6, 'SQ_INSTS_VALU_MUL_F64': 3.7e6, 'SQ_INSTS_VALU_ADD_F64': 0.5e6})

    def trace(ns):
        with open(tmp_path / 'trace.csv', 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
            w.writeheader()
            w.writerow({'Kernel_Name': kernel, 'Start_Timestamp': 1000, 'End_Timestamp': 1000 + ns})
    monkeypatch.setattr(dv, 'ROOT', str(tmp_path))              # valu.json is written under <ROOT>/profiles
    os.makedirs(tmp_path / 'profiles')
    for name in ('r04_valu_issue_cost.txt', 'r04_valu_issue_cost32.txt', 'valu_mix.json'):
        with open(os.path.join(ROOT, 'profiles', name)) as src, open(tmp_path / 'profiles' / name, 'w') as dst:
            dst.write(src.read())
    argv = ['derive_valu.py', 'label', '65536', '8', '256', str(tmp_path / 'sq1.csv'), str(tmp_path / 'trace.csv'), str(tmp_path / 'sq3.csv')]
    monkeypatch.setattr('sys.argv', argv)
    mix_hash = json.load(open(os.path.join(ROOT, 'profiles', 'valu_mix.json')))['csrc_hash']
    monkeypatch.setattr(bench, '_CSRC_HASH', mix_hash)
    trace(170000)                                               # 0.17 ms: plausible
    dv.main()
    entry = json.load(open(tmp_path / 'profiles' / 'valu.json'))['kernels'][0]
    assert 0.5 < entry['valu_share_of_launch_in_pass'] < 1.0 and entry['csrc_hash'] == mix_hash
    assert entry['waves_resident_per_simd'] < 2.5               # from the counters, not waves / 1024
    trace(60000)                                                # 0.06 ms: the same instructions cannot fit
    with pytest.raises(SystemExit) as err:
        dv.main()
    assert 'cannot be' in str(err.value)
