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


def _kernel_resources(asm_text):
    """{kernel symbol: (sgpr spills, vgpr spills, scratch bytes)} from the .amdgpu_metadata of a -S listing."""
    out = {}
    for block in asm_text.split('  - .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', block).group(1)
        out[name] = (int(re.search(r'\.sgpr_spill_count:\s+(\d+)', block).group(1)),
                     int(re.search(r'\.vgpr_spill_count:\s+(\d+)', block).group(1)),
                     int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', block).group(1)))
    return out


@pytest.fixture(scope='module')
def device_listings(tmp_path_factory):
    """hipcc -S listings (gfx950 device code + .amdgpu_metadata) of every translation unit that holds dispatchable kernels,
    compiled with the Makefile's flags: {listing name: text}."""
    tmp_path = tmp_path_factory.mktemp('listings')
    jobs = []
    flags = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-I' + os.path.join(ROOT, 'include'),
             '-S', '--cuda-device-only']
    for unit in ('mapf_lg_kernels', 'mapf_lg_rollout', 'mapf_transitions'):
        jobs.append((unit, [], tmp_path / (unit + '.s')))
    jobs.append(('mapf_lq_step', ['-mllvm', '-amdgpu-kernarg-preload-count=14'], tmp_path / 'mapf_lq_step.s'))   # (as the Makefile)
    for k in (8, 4, 2):                                           # packed-layout rollout: one object per (K, RECORD)
        for r in (1, 0):
            jobs.append(('mapf_lq_rollout', ['-DMAPF_LQ_K=%d' % k, '-DMAPF_LQ_RECORD=%d' % r], tmp_path / ('mapf_lq_k%d_r%d.s' % (k, r))))
    for g in range(4):
        jobs.append(('mapf_kernels', ['-DMAPF_GROUP=%d' % g], tmp_path / ('mapf_kernels_g%d.s' % g)))
    procs = [(out, subprocess.Popen(flags + extra + [os.path.join(CSRC, unit + '.hip'), '-o', str(out)],
                                    stderr=subprocess.DEVNULL)) for unit, extra, out in jobs]
    listings = {}
    for out, proc in procs:
        assert proc.wait() == 0, out
        listings[out.name] = out.read_text()
    return listings


def test_every_dispatchable_kernel_is_free_of_register_spills(device_listings):
    """No code object the launchers can reach may spill registers: a spilling thread-per-env rollout kernel is the
    one place a wrong result was ever observed on the GPU (compiler notes), and spills cost time.  Covers
    the lane-group and packed-layout families, the transition kernels AND the thread-per-env family (mapf_kernels.hip,
    compiled once per agent-count group like the Makefile does); of the latter's rollout kernels only those
    launch_rollout_g* dispatches (A <= kTpeRolloutMaxAgents) are held to it -- the others are never launched."""
    resources = {}
    for text in device_listings.values():
        resources.update(_kernel_resources(text))
    assert len(resources) >= 200
    max_tpe_rollout = int(re.search(r'#define MAPF_TPE_ROLLOUT_MAX\s+(\d+)', open(os.path.join(CSRC, 'mapf_kernels.hpp')).read()).group(1))
    checked = tpe_step = tpe_rollout = 0
    for name, (sgpr, vgpr, scratch) in resources.items():
        m = re.match(r'_ZN4mapf14rollout_kernelILi(\d+)E', name)
        if m and int(m.group(1)) > max_tpe_rollout:
            continue                                              # compiled, never dispatched
        tpe_rollout += bool(m)
        tpe_step += name.startswith('_ZN4mapf11step_kernelILi')
        assert (sgpr, vgpr, scratch) == (0, 0, 0), (name, sgpr, vgpr, scratch)
        checked += 1
    assert tpe_step == 32 and tpe_rollout == max_tpe_rollout and checked >= 130


def _kernel_metadata_blocks(asm_text):
    """{kernel symbol: its block of the .amdgpu_metadata kernel list}"""
    return {re.search(r'\.name:\s+(\S+)', block).group(1): block for block in asm_text.split('  - .agpr_count:')[1:]}


def test_packed_kernels_layout_assumptions_hold_in_the_compiled_objects(device_listings):
    """The packed kernels hard-code two facts about their own code objects; a compiler update that moves either must fail
    HERE (a build-time check on the CPU), not as a wrong address on the GPU:
      * lq_step_kernel<BIG> re-reads its StepArgs block from the kernarg segment at kStepArgsOffset (mapf_lq_step.hip):
        the by-value StepArgs argument must sit at exactly that offset, behind the 14 preloaded dwords;
      * lds_at() names LDS locations by ABSOLUTE byte address (mapf_lq.hpp): the kernel's table image must be the kernel's
        only LDS object, i.e. start at LDS address 0 -- static LDS of exactly sizeof(TableImage) for the plain step (one
        object of the whole fixed size), none at all for the kernels whose image is the dynamic segment."""
    src = open(os.path.join(CSRC, 'mapf_lq_step.hip')).read()
    m = re.search(r'constexpr uint32_t kStepArgsOffset = ([0-9 *+]+);', src)
    args_offset = eval(m.group(1))                                # "5 * 8 + 4 * 4": plain integer arithmetic
    text = device_listings['mapf_lq_step.s']
    blocks = _kernel_metadata_blocks(text)
    step = {n: b for n, b in blocks.items() if 'lq_step_kernel' in n}
    assert len(step) >= 60                                        # 9 plain (Q, K) x 4 + BIG forms
    n_big = 0
    for name, block in step.items():
        args = [(int(o), int(z), k) for o, z, k in re.findall(r'\.offset:\s+(\d+)\n\s+\.size:\s+(\d+)\n\s+\.value_kind:\s+(\w+)', block)]
        by_value = [a for a in args if a[2] == 'by_value']
        leading = [a for a in args if a[0] < args_offset]
        assert [a[0] for a in leading] == [0, 8, 16, 24, 32, 40, 44, 48, 52], (name, leading)   # five pointers, four dwords = 14 dwords
        block_arg = [a for a in by_value if a[1] > 64]
        assert len(block_arg) == 1 and block_arg[0][0] == args_offset, (name, block_arg, args_offset)
        lds = int(re.search(r'\.group_segment_fixed_size:\s+(\d+)', block).group(1))
        big = re.search(r'lq_step_kernelILi\d+ELi\d+ELb[01]ELb[01]ELi[123]E', name) is not None   # LDS-table forms: 16-byte rows / delta rows
        n_big += big
        assert lds == (0 if big else 1024), (name, lds)
        # the descriptor asks the command processor for the 14 leading dwords
        desc = text[text.index('.amdhsa_kernel ' + name):]
        desc = desc[:desc.index('.end_amdhsa_kernel')]
        assert re.search(r'\.amdhsa_user_sgpr_kernarg_preload_length\s+14\b', desc), name
    assert n_big >= 40                                            # 3 x 4 (eight agents per lane) + 4 x 4 (four) + 4 x 4 (delta rows) + 4 (delta rows + bitmaps)
    n_rollout = 0
    for listing, text in device_listings.items():
        if not listing.startswith('mapf_lq_k'):
            continue
        for name, block in _kernel_metadata_blocks(text).items():
            if 'lq_rollout_kernel' in name:
                assert int(re.search(r'\.group_segment_fixed_size:\s+(\d+)', block).group(1)) == 0, name
                n_rollout += 1
    assert n_rollout >= 100
    # ... and no trap instruction stands in for these checks in the shipped kernels any more
    for unit in ('mapf_lq_step.hip', 'mapf_lq_rollout.hip'):
        assert '__builtin_trap' not in open(os.path.join(CSRC, unit)).read(), unit


def test_packed_rollout_dispatch_never_plans_past_the_lds_or_launch_bounds():
    """mapf_debug_rollout_plan is the arithmetic try_launch_rollout_lq runs before every packed rollout launch (no device
    involved).  Swept over map sizes across every form's LDS boundary, agent counts, batch sizes, streamed / in-kernel policy,
    delta rows or not, and the MAPF_TUNE overrides the tests use: whatever form is planned fits the CU's 160 KB of LDS (the limit
    the launcher raises the kernel's dynamic segment to), fills whole blocks, and stays within its instance's launch bounds;
    and the forms end exactly where the next cell would not fit."""
    from gym_mapf_amd import _native
    import ctypes
    lib = _native.load()
    out = (ctypes.c_uint64 * 6)()
    LDS = 160 * 1024

    def plan(n_cells, A, E, streamed, delta, tune, n_cu=256, T=64):
        rc = lib.mapf_debug_rollout_plan(n_cells, A, E, T, streamed, delta, n_cu, tune, out)
        assert rc in (0, 1), (rc, lib.mapf_last_error())
        return rc, tuple(out)

    cells = sorted(set(list(range(2, 200, 7)) + list(range(600, 760, 3)) + list(range(800, 1800, 11)) + list(range(1650, 1720)) +
                       list(range(3000, 3400, 5)) + list(range(4000, 20500, 61)) + list(range(19700, 20300, 3)) + [683, 3278, 4097, 65535]))
    tunes = [None, b'k=8', b'k=4', b'k=2', b'bitmap_block=1024', b'bitmap_block=512', b'bitmap_pairs=0', b'bitmap_delta=0',
             b'bitmap_staycol=0', b'mv_lds_max_bytes=163840', b'mv_lds_max_bytes=0', b'quad_min_lanes=0,oct_min_lanes=0']
    seen_forms, n_packed = set(), 0
    for tune in tunes:
        for A in (2, 4, 8, 16, 32, 64, 128):
            for E in (64, 1000, 1024, 4096, 16384, 16448, 65536, 131072, 262144):
                for streamed in (1, 0):
                    for delta in (0, 1):
                        for V in cells:
                            rc, (K, Q, form, block, table, total) = plan(V, A, E, streamed, delta, tune)
                            if not rc:
                                continue
                            n_packed += 1
                            seen_forms.add(form)
                            ctx = (tune, A, E, streamed, delta, V, K, Q, form, block, table, total)
                            assert K in (2, 4, 8) and K * Q == A and Q in (1, 2, 4, 8, 16), ctx
                            assert block in (64, 128, 256, 512, 1024) and E % (block // Q) == 0, ctx
                            assert 1024 < table <= total <= LDS, ctx
                            assert (total > table) == (form in (2, 3, 4, 5)), ctx          # bitmaps behind the table
                            if form >= 2:
                                assert A == 32 and K == 4 and total == table + (block // 8) * ((((V + 31) // 32) * 4 + 15) & ~15), ctx
                            if form == 5:
                                assert delta, ctx
                            # launch bounds of the instances (mapf_lq_rollout.hip: 512 threads with eight agents per lane and for the
                            # in-kernel policy behind four 8-byte columns + bitmaps, 1024 otherwise)
                            assert block <= (512 if K == 8 or (form == 2 and not streamed) else 1024), ctx
    assert seen_forms == {0, 1, 2, 3, 4, 5} and n_packed > 50000, (seen_forms, n_packed)
    # the boundaries themselves (32 agents, default tuning), to the cell
    stride = lambda V: (((V + 31) // 32) * 4 + 15) & ~15                                   # noqa: E731  (bytes of one env's bitmap)
    for delta, streamed in ((1, 1), (0, 1), (0, 0)):
        forms = {V: plan(V, 32, 16384, streamed, delta, None) for V in range(3000, 4200)}
        last = max(V for V, (rc, _) in forms.items() if rc)
        assert 1024 + last * 40 <= LDS - 1024 < 1024 + (last + 1) * 40                     # the 8-byte rows' own limit (launcher's reserve kept)
        assert not any(rc for V, (rc, _) in forms.items() if V > last)
        five = [V for V, (rc, o) in forms.items() if rc and o[2] == 3]                     # five columns + 64 bitmaps
        four = [V for V, (rc, o) in forms.items() if rc and o[2] == 2]                     # four columns + 64 bitmaps
        if delta:
            assert not five and not four and all(o[2] == 5 for rc, o in forms.values() if rc)   # delta rows: 24 bytes a cell, fit to the end
        else:
            assert max(five) + 1 == min(four) and max(four) == last
            assert 1024 + max(five) * 40 + 64 * stride(max(five)) <= LDS < 1024 + (max(five) + 1) * 40 + 64 * stride(max(five) + 1)
            assert 1024 + last * 32 + 64 * stride(last) <= LDS
    # a malformed override is an error, not a default
    assert lib.mapf_debug_rollout_plan(683, 8, 65536, 64, 1, 0, 256, b'k=nine', out) < 0
    assert lib.mapf_debug_rollout_plan(683, 8, 65536, 64, 1, 0, 256, b'no_such_key=1', out) < 0
