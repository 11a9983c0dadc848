"""Known-answer tests for the Philox4x32-10 definition used by oracle and kernel.

Vectors: Random123 (Salmon et al., SC'11) kat_vectors for philox4x32 / 10 rounds.
"""
import numpy as np

import philox

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_scalar_kat():
    for ctr, key, out in KAT:
        assert philox.philox4x32_10(ctr, key) == out


def test_philox_numpy_matches_scalar():
    for ctr, key, out in KAT:
        got = philox.philox4x32_10_np(*[np.array([c]) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == out


def test_uniform_layout():
    ids = np.array([0, 1, 65535, (1 << 32) + 5, (1 << 40) + 123456789], np.uint64)
    for t in (0, 7, (1 << 33) + 9):
        u = philox.slip_uniforms_np(42, ids, t, 7)
        assert u.dtype == np.float64 and (u >= 0).all() and (u < 1).all()
        for j, e in enumerate(ids):
            for a in range(7):
                assert u[j, a] == philox.slip_uniform(42, int(e), t, a)
    # 53-bit grid: u * 2**53 is an integer
    assert np.all(np.mod(u * 2.0 ** 53, 1.0) == 0)


def test_action_stream_range_and_determinism():
    a = philox.random_actions_np(42, np.arange(4096), 3, 8)
    assert a.dtype == np.uint8 and a.max() == 4 and a.min() == 0
    assert np.array_equal(a, philox.random_actions_np(42, np.arange(4096), 3, 8))
    assert not np.array_equal(a, philox.random_actions_np(42, np.arange(4096), 4, 8))
    cnt = np.bincount(a.ravel(), minlength=5) / a.size
    assert np.all(np.abs(cnt - 0.2) < 0.01)
