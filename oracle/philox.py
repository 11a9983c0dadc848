"""Philox4x32-10 counter-based RNG and the (seed, env, step, agent) -> uniform map.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may use it (as the checker, never as the thing shipped).

The reference (gym_mapf/envs/mapf_env.py:139, :255) draws one
``RandomState.rand()`` per agent per non-terminal step.  north_star replaces
that stream by a counter-based generator so that thousands of envs can draw
independently on the GPU; this file is the CPU definition of that generator.
The HIP kernel (gym-mapf_amd/csrc/mapf_kernels.hip: philox4x32_10 /
slip_uniform53) must reproduce these words bit for bit.

Algorithm: Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3"
(SC'11), Philox-4x32 with 10 rounds; pinned by the Random123 known-answer
vectors in tests/test_philox.py.

Stream layout (shared with include/mapf_hip.h):

    key   = (seed & 0xffffffff, seed >> 32)
    h     = t >> 1                                  # one Philox call serves TWO consecutive steps of an agent quad
    quad  = agent >> 2
    slot  = 4 * (t & 1) + (agent & 3)               # 0..7: which of the call's eight half-words
    ctr   = (env_id & 0xffffffff,
             env_id >> 32,
             h & 0xffffffff,
             ((h >> 32) & 0xffff) | (quad << 16) | (rslot << 23) | (refine << 31))
    W     = philox4x32_10(ctr with rslot = 0, refine = 0, key)
    hi16  = (W[slot >> 1] >> (16 * (slot & 1))) & 0xffff          # word 2 * (t & 1) + ((agent >> 1) & 1): low half = the
                                                                  # even agent of the pair, high half = the odd one
    R     = philox4x32_10(ctr with rslot = slot, refine = 1, key)
    lo37  = ((R[0] & 0x1f) << 32) | R[1]
    u     = (hi16 * 2**37 + lo37) / 2**53            # 53-bit, in [0, 1) -- the resolution of RandomState.rand()

A 53-bit uniform is thus split over two counters.  The kernel compares hi16 against the top 16 bits of its
thresholds and evaluates the refine = 1 call only in the (probability ~2^-15 per agent-step) case where those
bits tie, so the value compared is exactly the u above while the common path costs ONE Philox call per four agents
per two steps: a single `step` of a lane that owns four agents needs one call (two of its words), a fused rollout
needs one call per lane per two steps.  (Rounds 1-3 used counter = (env, t >> 2, agent PAIR): the same cost per
rollout step, but a single step of a four-agent lane then needed two calls and used one word of each.)  The CPU
oracle simply computes both calls every time.

``t`` is the handle-global step index (number of ``step`` calls so far), so a terminal-state step simply
leaves its counters unused -- equivalent to the reference's "no draw on terminal steps" because nothing
downstream depends on how many words an env has consumed.
"""
import numpy as np

PHILOX_M0 = 0xD2511F53
PHILOX_M1 = 0xCD9E8D57
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
MASK32 = 0xFFFFFFFF

ACTION_STREAM_KEY_OFFSET = 1  # actions use key = seed + 1


def philox4x32_10(ctr, key):
    """Scalar Philox4x32-10.  ctr: 4 ints, key: 2 ints -> tuple of 4 uint32."""
    c0, c1, c2, c3 = (int(x) & MASK32 for x in ctr)
    k0, k1 = (int(x) & MASK32 for x in key)
    for r in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> 32, p0 & MASK32
        hi1, lo1 = p1 >> 32, p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK32, lo1, (hi0 ^ c3 ^ k1) & MASK32, lo0
        k0 = (k0 + PHILOX_W0) & MASK32
        k1 = (k1 + PHILOX_W1) & MASK32
    return c0, c1, c2, c3


def philox4x32_10_np(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 on uint64-held 32-bit lanes (numpy arrays)."""
    c0 = np.asarray(c0, dtype=np.uint64) & np.uint64(MASK32)
    c1 = np.asarray(c1, dtype=np.uint64) & np.uint64(MASK32)
    c2 = np.asarray(c2, dtype=np.uint64) & np.uint64(MASK32)
    c3 = np.asarray(c3, dtype=np.uint64) & np.uint64(MASK32)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint64(int(k0) & MASK32)
    k1 = np.uint64(int(k1) & MASK32)
    m32 = np.uint64(MASK32)
    s32 = np.uint64(32)
    for r in range(10):
        p0 = np.uint64(PHILOX_M0) * c0
        p1 = np.uint64(PHILOX_M1) * c2
        hi0, lo0 = p0 >> s32, p0 & m32
        hi1, lo1 = p1 >> s32, p1 & m32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & m32, lo1, (hi0 ^ c3 ^ k1) & m32, lo0
        k0 = (k0 + np.uint64(PHILOX_W0)) & m32
        k1 = (k1 + np.uint64(PHILOX_W1)) & m32
    return c0, c1, c2, c3


def _ctr_words(env_id, t, agent, refine):
    h = t >> 1
    slot = 4 * (t & 1) + (agent & 3)
    c0 = env_id & MASK32
    c1 = (env_id >> 32) & MASK32
    c2 = h & MASK32
    c3 = ((h >> 32) & 0xFFFF) | (((agent >> 2) & 0x1F) << 16)
    if refine:
        c3 |= (slot << 23) | (1 << 31)
    return c0, c1, c2, c3


def slip_uniform(seed, env_id, t, agent):
    """The 53-bit uniform the slip model of (env_id, step t, agent) consumes."""
    key = (seed & MASK32, (seed >> 32) & MASK32)
    t, agent = int(t), int(agent)
    slot = 4 * (t & 1) + (agent & 3)
    w = philox4x32_10(_ctr_words(int(env_id), t, agent, 0), key)
    r = philox4x32_10(_ctr_words(int(env_id), t, agent, 1), key)
    hi16 = (w[slot >> 1] >> (16 * (slot & 1))) & 0xFFFF
    lo37 = ((r[0] & 0x1F) << 32) | r[1]
    return ((hi16 << 37) | lo37) / 9007199254740992.0


def slip_uniforms_np(seed, env_ids, t, n_agents):
    """u[E, A] float64 for global env ids ``env_ids`` at step ``t``."""
    env_ids = np.asarray(env_ids, dtype=np.uint64).reshape(-1, 1)
    E = env_ids.shape[0]
    agents = np.arange(n_agents, dtype=np.uint64).reshape(1, -1)
    quads = agents >> np.uint64(2)
    t = int(t)
    h = t >> 1
    slot = np.uint64(4 * (t & 1)) + (agents & np.uint64(3))                    # [1, A]
    c0 = env_ids & np.uint64(MASK32)
    c1 = env_ids >> np.uint64(32)
    c2 = np.uint64(h & MASK32)
    c3 = np.uint64((h >> 32) & 0xFFFF) | (quads << np.uint64(16))
    k0, k1 = seed & MASK32, (seed >> 32) & MASK32
    w = philox4x32_10_np(c0, c1, c2, c3, k0, k1)
    word = np.choose(np.broadcast_to((slot >> np.uint64(1)).astype(np.int64), (E, n_agents)), w)
    hi16 = (word >> (np.uint64(16) * (slot & np.uint64(1)))) & np.uint64(0xFFFF)
    r = philox4x32_10_np(c0, c1, c2, c3 | (slot << np.uint64(23)) | np.uint64(1 << 31), k0, k1)
    lo37 = ((r[0] & np.uint64(0x1F)) << np.uint64(32)) | r[1]
    mant = (hi16 << np.uint64(37)) | lo37
    return mant.astype(np.float64) / 9007199254740992.0


def random_actions_np(seed, env_ids, t, n_agents):
    """Synthetic policy used by bench/rollout: actions u8[E, A] over 0..4.

    Stream (this build's definition, ABI 5): key = seed + 1; ONE Philox call serves an agent QUAD for FOUR
    consecutive steps -- m = t >> 2, ctr = (env_lo, env_hi, m_lo, (m_hi & 0xffff) | ((agent >> 2) << 16)); word
    t & 3 of the call belongs to step t, its byte agent & 3 to the agent: action = (byte * 5) >> 8.  (A byte has 256
    values: STAY is drawn with probability 52/256, every move with 51/256 -- uniform to within 1/256; a fused rollout
    pays one policy call per lane per four steps and one multiply + shift per action.  ABI <= 4 spent one call per
    quad per STEP and a whole 32-bit word per action, which made the in-kernel policy dearer than loading actions.)
    """
    seed = (int(seed) + ACTION_STREAM_KEY_OFFSET) & 0xFFFFFFFFFFFFFFFF
    env_ids = np.asarray(env_ids, dtype=np.uint64).reshape(-1, 1)
    agents = np.arange(n_agents, dtype=np.uint64).reshape(1, -1)
    quads = agents >> np.uint64(2)
    t = int(t)
    m = t >> 2
    c0 = env_ids & np.uint64(MASK32)
    c1 = env_ids >> np.uint64(32)
    c2 = np.uint64(m & MASK32)
    c3 = np.uint64((m >> 32) & 0xFFFF) | (quads << np.uint64(16))
    w = philox4x32_10_np(c0, c1, c2, c3, seed & MASK32, (seed >> 32) & MASK32)
    word = w[t & 3]
    byte = (word >> (np.uint64(8) * (agents & np.uint64(3)))) & np.uint64(0xFF)
    return ((byte * np.uint64(5)) >> np.uint64(8)).astype(np.uint8)


def random_action(seed, env_id, t, agent):
    """Scalar form of ``random_actions_np`` (one action)."""
    key = (int(seed) + ACTION_STREAM_KEY_OFFSET) & 0xFFFFFFFFFFFFFFFF
    env_id, t, agent = int(env_id), int(t), int(agent)
    m = t >> 2
    w = philox4x32_10((env_id & MASK32, (env_id >> 32) & MASK32, m & MASK32, ((m >> 32) & 0xFFFF) | ((agent >> 2) << 16)),
                      (key & MASK32, (key >> 32) & MASK32))
    return (((w[t & 3] >> (8 * (agent & 3))) & 0xFF) * 5) >> 8
