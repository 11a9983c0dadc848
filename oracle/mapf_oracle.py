"""CPU restatement of gym-mapf's ``MapfEnv.step()`` hot path -- the parity oracle.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module, and only as the
checker / the timed CPU baseline -- never as the product path.  The product
(``gym-mapf_amd/``) does not import anything from ``oracle/`` and raises when
its HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` ran the unmodified
reference (``/root/reference/gym_mapf``) in the build container and recorded
its outputs in ``tests/golden/*.npz|json``; ``tests/test_oracle_golden.py``
checks this restatement against every one of them bit for bit, and
``tests/test_reference_cases.py`` re-expresses the reference's own 25 unit
tests.  One boundary is UNPINNED: the seeded MT19937 stream that
``gym.utils.seeding.np_random(42)`` would produce lives in the third-party
``gym==0.13.0`` package (requirements.txt:7), which is absent here; the oracle
never relies on it -- every uniform is injected by the caller.

Everything below is written from the behaviour documented in SURVEY.md 3.3;
each function cites the reference lines it restates (paths relative to
/root/reference/).
"""
from itertools import product

# gym_mapf/envs/__init__.py:26 -- ACTIONS = [STAY, UP, RIGHT, DOWN, LEFT]
STAY, UP, RIGHT, DOWN, LEFT = 0, 1, 2, 3, 4
N_ACTIONS = 5
ACTION_NAMES = ('STAY', 'UP', 'RIGHT', 'DOWN', 'LEFT')

# gym_mapf/envs/__init__.py:19-25 -- POSSIBILITIES[a] = (slip "right", slip "left")
SLIP_RIGHT = (STAY, RIGHT, DOWN, LEFT, UP)
SLIP_LEFT = (STAY, LEFT, UP, RIGHT, DOWN)

MAKESPAN, SOC = 0, 1


# --------------------------------------------------------------------------- grid
def free_cells_column_major(lines):
    """Column-major list of free (row, col) cells.

    gym_mapf/envs/grid.py:37-40 iterates columns outermost; mapf_env.py:142
    keeps the cells that are EmptyCell.  Lines are stripped (grid.py:20).
    """
    rows = [ln.strip() for ln in lines]
    for ln in rows:
        for ch in ln:
            if ch not in '.@':
                raise KeyError(ch)  # grid.py:21 CHAR_TO_CELL lookup
    n_rows, n_cols = len(rows), len(rows[0])
    return rows, [(r, c) for c in range(n_cols) for r in range(n_rows) if rows[r][c] == '.']


def move_cell(rows, loc, action):
    """One-cell move with border clamp and obstacle bounce.

    gym_mapf/envs/mapf_env.py:43-75: the target is clamped to the map edge
    (max(0, .) / min(len-1, .)) and the agent stays put if the clamped target
    is an obstacle.
    """
    r, c = loc
    if action == UP:
        t = (max(0, r - 1), c)
    elif action == DOWN:
        t = (min(len(rows) - 1, r + 1), c)
    elif action == RIGHT:
        t = (r, min(len(rows[0]) - 1, c + 1))
    elif action == LEFT:
        t = (r, max(0, c - 1))
    else:
        return loc
    return loc if rows[t[0]][t[1]] == '@' else t


def neighbour_table(lines):
    """nbr[v][a] = local id reached from free cell v by noise-free action a."""
    rows, cells = free_cells_column_major(lines)
    index = {loc: i for i, loc in enumerate(cells)}
    return [[index[move_cell(rows, loc, a)] for a in range(N_ACTIONS)] for loc in cells]


# ---------------------------------------------------------------- joint-int codecs
def encode_mixed_radix(digits, base):
    """gym_mapf/envs/__init__.py:70-79 with equal bases: sum d_i * base**i."""
    total, mul = 0, 1
    for d in digits:
        total += d * mul
        mul *= base
    return total


def decode_mixed_radix(x, base, n):
    """gym_mapf/envs/__init__.py:50-67: n digits, least significant first."""
    out = []
    for _ in range(n):
        out.append(x % base)
        x //= base
    return tuple(out)


# ------------------------------------------------------------------ slip model
def slip_distribution(nbr_row, action, fail_prob):
    """Merged single-agent movement list [(next_cell, prob), ...].

    gym_mapf/envs/mapf_env.py:163-184: candidates are (1-rf-lf, a), (rf,
    right-of-a), (lf, left-of-a) with rf = lf = fail_prob/2 (:131-132); entries
    with p <= 0 are dropped (:172); entries that reach an already-listed cell
    are merged with ``old + new`` in first-seen order (:177-182).
    """
    rf = fail_prob / 2
    lf = fail_prob / 2
    cand = ((1 - rf - lf, action), (rf, SLIP_RIGHT[action]), (lf, SLIP_LEFT[action]))
    cells, probs = [], []
    for p, a in cand:
        if not p > 0:
            continue
        nxt = nbr_row[a]
        if nxt in cells:
            k = cells.index(nxt)
            probs[k] = probs[k] + p
        else:
            cells.append(nxt)
            probs.append(p)
    return list(zip(cells, probs))


def categorical_index(probs, u):
    """``(cumsum(probs) > u).argmax()`` -- gym==0.13.0
    gym/envs/toy_text/discrete.py categorical_sample (third-party, absent from
    the tree; call site mapf_env.py:255).  numpy's cumsum is a left-to-right
    float64 running sum; argmax of an all-False mask is 0.
    """
    run = 0.0
    for i, p in enumerate(probs):
        run = p if i == 0 else run + p
        if run > u:
            return i
    return 0


class OracleEnv:
    """Scalar restatement of ``MapfEnv`` restricted to the step/reset path.

    Constructor mirrors gym_mapf/envs/mapf_env.py:116-161.  Randomness is
    injected: ``step(actions, uniforms)`` takes the A uniforms the reference
    would have drawn from ``self.np_random.rand()`` in agent order (:253-257).
    """

    def __init__(self, lines, n_agents, starts, goals, fail_prob,
                 r_clash, r_goal, r_living, criteria=MAKESPAN):
        self.rows, self.cells = free_cells_column_major(lines)
        self.index = {loc: i for i, loc in enumerate(self.cells)}
        self.V = len(self.cells)
        self.A = n_agents
        self.nbr = [[self.index[move_cell(self.rows, loc, a)] for a in range(N_ACTIONS)]
                    for loc in self.cells]
        self.fail_prob = fail_prob
        self.r_clash, self.r_goal, self.r_living = r_clash, r_goal, r_living
        self.criteria = criteria
        self.start = tuple(self.index[tuple(l)] for l in starts)   # KeyError on obstacle (:143,:369)
        self.goal = tuple(self.index[tuple(l)] for l in goals)
        if len(self.start) != n_agents or len(self.goal) != n_agents:
            raise AssertionError('locations number differs from n_agents')  # :366-367
        self._dist = {}
        self.local = self.start

    # -- MAPF_POLICY_GREEDY of include/mapf_hip.h (no reference counterpart): per agent the action whose intended
    #    target is closest (Manhattan, over (row, col) locations) to its goal, first minimum in ACTIONS order wins
    def greedy_actions(self):
        acts = []
        for cell, goal in zip(self.local, self.goal):
            gr, gc = self.cells[goal]
            dist = [abs(self.cells[self.nbr[cell][a]][0] - gr) + abs(self.cells[self.nbr[cell][a]][1] - gc)
                    for a in range(N_ACTIONS)]
            acts.append(dist.index(min(dist)))
        return acts

    # -- reference reset(): mapf_env.py:290-293 (no reseed)
    def reset(self):
        self.local = self.start
        return self.local

    @property
    def s(self):
        return encode_mixed_radix(self.local, self.V)

    def is_terminal(self, local):
        """mapf_env.py:210-223: any shared cell, or every agent on its goal."""
        if len(set(local)) != len(local):
            return True
        return all(l == g for l, g in zip(local, self.goal))

    def distribution(self, cell, action):
        key = (cell, action)
        d = self._dist.get(key)
        if d is None:
            d = self._dist[key] = slip_distribution(self.nbr[cell], action, self.fail_prob)
        return d

    def living_reward(self, prev, actions):
        """mapf_env.py:436-446."""
        if self.criteria == MAKESPAN:
            return self.r_living
        stayed = sum(1 for i in range(self.A) if prev[i] == self.goal[i] and actions[i] == STAY)
        return (self.A - stayed) * self.r_living

    def is_collision(self, prev, nxt):
        """mapf_env.py:378-389: swap or shared target for any pair."""
        for i in range(self.A):
            for j in range(i + 1, self.A):
                if prev[i] == nxt[j] and prev[j] == nxt[i]:
                    return True
                if nxt[i] == nxt[j]:
                    return True
        return False

    def transition_reward(self, prev, actions, nxt):
        """mapf_env.py:225-235: collision is tested before goal."""
        living = self.living_reward(prev, actions)
        if self.is_collision(prev, nxt):
            return self.r_clash + living, True, True
        if all(n == g for n, g in zip(nxt, self.goal)):
            return self.r_goal + living, True, False
        return living, False, False

    def step(self, actions, uniforms):
        """mapf_env.py:237-266.

        Returns (next_local, reward, done, collision, prob, was_terminal).  On
        a terminal-state step the reference returns ``(s, 0, True, {"prob":
        0})`` with no collision key and draws nothing (:239-240).
        """
        prev = self.local
        if self.is_terminal(prev):
            return prev, 0, True, False, 0, True
        nxt = []
        prob = 1
        for i in range(self.A):
            dist = self.distribution(prev[i], actions[i])
            k = categorical_index([p for _, p in dist], uniforms[i])
            nxt.append(dist[k][0])
            prob *= dist[k][1]
        nxt = tuple(nxt)
        reward, done, collision = self.transition_reward(prev, actions, nxt)
        self.local = nxt
        return nxt, reward, done, collision, prob, False

    def transitions(self, local, actions):
        """All branches of env.P[s][a]: mapf_env.py:448-478 (same order)."""
        if self.is_terminal(local):
            return [((1.0, False), tuple(local), 0, True)]
        dists = [self.distribution(local[i], actions[i]) for i in range(self.A)]
        out = []
        for comb in product(*dists):
            prob = comb[0][1]
            for _, p in comb[1:]:
                prob = prob * p
            nxt = tuple(c for c, _ in comb)
            reward, done, collision = self.transition_reward(tuple(local), actions, nxt)
            out.append(((prob, collision), nxt, reward, done))
        return out
