"""ctypes binding of ``lib/libmapf_hip.so`` (C ABI: ``include/mapf_hip.h``).

This is the only door to the hot path.  There is deliberately no CPU fallback: if the
library is missing or no HIP device is usable the calls raise ``MapfNativeError``.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_double, c_int, c_int32, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

LIB_PATH = os.environ.get('MAPF_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libmapf_hip.so')

MAPF_ABI_VERSION = 5            # include/mapf_hip.h: the version this binding was written against (checked at load)
MAPF_OK, MAPF_EINVAL, MAPF_EHIP, MAPF_ENODEVICE, MAPF_EUNSUPPORTED = 0, -1, -2, -3, -4
MAPF_MAX_AGENTS = 128
MAPF_MAKESPAN, MAPF_SOC = 0, 1
MAPF_FLAG_DEVICE_PTRS, MAPF_FLAG_START_BROADCAST, MAPF_FLAG_GOAL_BROADCAST = 0x1, 0x2, 0x4
MAPF_FLAG_THREAD_PER_ENV, MAPF_FLAG_LANE_GROUP = 0x10, 0x20
MAPF_TPE_MAX_AGENTS = 16
MAPF_POLICY_RANDOM, MAPF_POLICY_GREEDY = 0, 1
MAPF_STEP_AUTO_RESET = 0x1
MAPF_KERNEL_STEP, MAPF_KERNEL_ROLLOUT, MAPF_KERNEL_TRANSITIONS = 0, 1, 2


class MapfNativeError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, 'libmapf_hip: %s (code %d)' % (message, code))
        self.code = code


class MapfDesc(Structure):
    _fields_ = [('struct_size', c_uint32), ('n_cells', c_uint32), ('n_agents', c_uint32), ('criteria', c_uint32),
                ('n_envs', c_uint64), ('env_id_offset', c_uint64), ('seed', c_uint64),
                ('nbr', c_void_p), ('start', c_void_p), ('goal', c_void_p),
                ('fail_prob', c_double), ('r_clash', c_double), ('r_goal', c_double), ('r_living', c_double),
                ('device', c_int32), ('flags', c_uint32), ('stream', c_void_p)]


class MapfRolloutIO(Structure):
    _fields_ = [('struct_size', c_uint32), ('n_steps', c_uint32), ('step_flags', c_uint32), ('accumulate', c_uint32),
                ('actions', c_void_p), ('out_returns', c_void_p), ('out_episodes', c_void_p),
                ('out_collisions', c_void_p), ('rec_local', c_void_p), ('rec_reward', c_void_p),
                ('rec_done', c_void_p), ('rec_collision', c_void_p), ('rec_prob', c_void_p)]


# every symbol include/mapf_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'mapf_create': (c_int, [POINTER(MapfDesc), POINTER(c_void_p)]),
    'mapf_destroy': (c_int, [c_void_p]),
    'mapf_reset': (c_int, [c_void_p, c_void_p]),
    'mapf_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32]),
    'mapf_rollout': (c_int, [c_void_p, POINTER(MapfRolloutIO)]),
    'mapf_fill_random_actions': (c_int, [c_void_p, c_void_p, c_uint64, c_uint32]),
    'mapf_set_policy': (c_int, [c_void_p, c_int, c_void_p]),
    'mapf_transitions': (c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    'mapf_transitions_window': (c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_uint64, c_uint32, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    'mapf_transitions_compact': (c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_uint64, c_uint32, c_uint64, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mapf_transition_rewards': (c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mapf_query_terminal': (c_int, [c_void_p, c_void_p]),
    'mapf_get_state': (c_int, [c_void_p, c_void_p, POINTER(c_uint64)]),
    'mapf_set_state': (c_int, [c_void_p, c_void_p, c_uint64]),
    'mapf_state_view': (c_int, [c_void_p, POINTER(c_void_p)]),
    'mapf_invalidate_state': (c_int, [c_void_p]),
    'mapf_graph_begin': (c_int, [c_void_p]),
    'mapf_graph_end': (c_int, [c_void_p, POINTER(c_void_p)]),
    'mapf_graph_launch': (c_int, [c_void_p, c_void_p, c_uint32]),
    'mapf_graph_steps': (c_int, [c_void_p, POINTER(c_uint64)]),
    'mapf_graph_destroy': (c_int, [c_void_p, c_void_p]),
    'mapf_sync': (c_int, [c_void_p]),
    'mapf_timer_begin': (c_int, [c_void_p]),
    'mapf_timer_end': (c_int, [c_void_p, POINTER(c_double)]),
    'mapf_get_stream': (c_int, [c_void_p, POINTER(c_void_p)]),
    'mapf_last_kernel': (c_char_p, [c_void_p, c_int]),
    'mapf_device_count': (c_int, [POINTER(c_int)]),
    'mapf_last_error': (c_char_p, []),
    'mapf_version': (c_char_p, []),
    'mapf_abi_version': (c_int, []),
    'mapf_debug_rollout_plan': (c_int, [c_uint32, c_int, c_uint64, c_uint32, c_int, c_int, c_int, c_char_p, POINTER(c_uint64)]),
}

_lib = None


def load():
    """Load the shared object once and type its entry points; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MapfNativeError(MAPF_ENODEVICE,
                                  '%s is missing -- build it with `python __graft_entry__.py build` '
                                  '(make -C gym-mapf_amd/csrc); there is no CPU fallback' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        # the ABI number covers the random streams' counter layouts too: a library of another version (MAPF_HIP_LIB pointing
        # at an old variant build) would run, and silently draw numbers the oracle does not
        abi = getattr(lib, 'mapf_abi_version', None)
        found = abi() if abi is not None else None
        if found != MAPF_ABI_VERSION:
            raise MapfNativeError(MAPF_EUNSUPPORTED, '%s has ABI %s, this binding needs ABI %d -- rebuild it (make -C gym-mapf_amd/csrc)'
                                  % (LIB_PATH, 'older than 5 (no mapf_abi_version)' if found is None else found, MAPF_ABI_VERSION))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _lib = lib
    return _lib


def check(rc):
    if rc != MAPF_OK:
        raise MapfNativeError(rc, load().mapf_last_error().decode('utf-8', 'replace'))


def device_count():
    n = c_int(0)
    rc = load().mapf_device_count(byref(n))
    return n.value if rc == MAPF_OK else 0


def version():
    return load().mapf_version().decode()
