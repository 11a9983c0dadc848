"""MapfGrid: the obstacle map, plus the tables the HIP path is built from.

Public behaviour follows the reference's ``gym_mapf/envs/grid.py`` (:1-46): ``grid[r, c]`` /
``grid[(r, c)]`` give ``EmptyCell`` / ``ObstacleCell``, ``grid[r]`` a row, ``len(grid)`` the
row count, iteration is COLUMN-major over all (row, col), ``==`` compares cell contents.
On top of that the grid exposes what the device needs: a uint8 obstacle array, the
column-major numbering of free cells (the reference's ``valid_locations``,
mapf_env.py:142) and the per-cell neighbour table that folds ``execute_action``
(mapf_env.py:43-94) into a lookup.
"""
import numpy as np


class ObstacleCell:
    pass


class EmptyCell:
    pass


CHAR_TO_CELL = {'.': EmptyCell, '@': ObstacleCell}
_CELL_OF_FLAG = (EmptyCell, ObstacleCell)


class MapfGrid:
    def __init__(self, map_lines):
        rows = []
        for raw in map_lines:
            text = raw.strip()                      # CRLF maps (Berlin_1_256) end in '\r\n'
            for ch in text:
                CHAR_TO_CELL[ch]                    # KeyError on anything but '.' / '@'
            rows.append([1 if ch == '@' else 0 for ch in text])
        self._rows = rows
        self.max_row = len(rows) - 1
        self.max_col = len(rows[0]) - 1
        self._tables = None

    # ------------------------------------------------------------ reference surface
    def __getitem__(self, *args):
        key = args[0]
        if type(key) == int:
            return [_CELL_OF_FLAG[f] for f in self._rows[key]]
        node = self._rows
        for k in key:
            node = node[k]
        if isinstance(node, list):
            return [_CELL_OF_FLAG[f] for f in node]
        return _CELL_OF_FLAG[node]

    def __iter__(self):
        n_rows = len(self._rows)
        for c in range(len(self._rows[0])):
            for r in range(n_rows):
                yield (r, c)

    def __len__(self):
        return len(self._rows)

    def __eq__(self, other):
        return self._rows == other._rows

    # ----------------------------------------------------------------- device tables
    @property
    def obstacles(self):
        """uint8[H, W], 1 where the cell is '@' (rows must be rectangular)."""
        return np.asarray(self._rows, dtype=np.uint8)

    def is_free(self, loc):
        return self._rows[loc[0]][loc[1]] == 0

    def tables(self):
        """(valid_locations list, loc_to_int dict, nbr uint16[V, 5]) -- built once.

        ``nbr[v, a]`` is the local id reached from free cell v by the noise-free action a
        (0 STAY, 1 UP, 2 RIGHT, 3 DOWN, 4 LEFT): clamp at the border, stay put when the
        clamped target is an obstacle (reference mapf_env.py:43-75).
        """
        if self._tables is None:
            obst = self.obstacles
            H, W = obst.shape
            free_cm = np.argwhere(obst.T == 0)                 # column-major: (col, row) pairs
            rr, cc = free_cm[:, 1], free_cm[:, 0]
            V = rr.shape[0]
            if V > 65536:
                raise ValueError('more than 65536 free cells: local ids do not fit uint16')
            ident = np.full((H, W), -1, dtype=np.int64)
            ident[rr, cc] = np.arange(V)
            nbr = np.empty((V, 5), dtype=np.int64)
            nbr[:, 0] = np.arange(V)
            for a, (dr, dc) in ((1, (-1, 0)), (2, (0, 1)), (3, (1, 0)), (4, (0, -1))):
                tr = np.clip(rr + dr, 0, H - 1)
                tc = np.clip(cc + dc, 0, W - 1)
                tgt = ident[tr, tc]
                nbr[:, a] = np.where(tgt < 0, nbr[:, 0], tgt)
            valid = [(int(r), int(c)) for r, c in zip(rr, cc)]
            self._tables = (valid, {loc: i for i, loc in enumerate(valid)}, nbr.astype(np.uint16))
        return self._tables
