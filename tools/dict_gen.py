#!/usr/bin/env python3
"""CylinderTag dictionary generator -- restatement of /root/reference/CylinderTag_generator.m:9-180,288-334 (select, dfs, inverse,
testConflict) for SURVEY.md 8(f) rank 4.  Offline host tooling: it produces `.marker` files the detector consumes
(CylinderTag::load_from_file, CylinderTag.cpp:16-41); nothing on the detection path imports it.

A dictionary is `tag_number` cyclic rows of `tag_col` codes in 0..63 (code = 8*left_id + right_id, both ids in 0..7 with the
"long" flag id >= 4 equal on both sides: generator.m:17, 80).  Every cyclic window of `feature_size` consecutive codes, read as
a base-64 number, must be unique over the whole dictionary AND over its mirror image (row reversed, every code replaced by
(7 - c%8)*8 + (7 - c//8): what a strip looks like upside down) -- testConflict, generator.m:288-334.

The reference searches with MATLAB's global rand stream and a 20 s wall-clock limit, so its output is not reproducible; this
restatement keeps the search (depth-first, most-constrained-successor ordering with random tie breaks, generator.m:62-180)
but takes a seed and a node budget.  The pinned parts are the predicates: the reference's own CTag_2f12c.marker satisfies
`legal_code` and `test_conflict` (tests/test_dictgen_cpu.py).

usage: python tools/dict_gen.py <tag_col> <feature_size> <tag_number> <out.marker> [seed]
"""
import sys

import numpy as np


def legal_code(c):
    """generator.m:17,80: both halves short (ids 0..3) or both long (ids 4..7)."""
    return not ((c % 8 <= 3 and c // 8 >= 4) or (c % 8 >= 4 and c // 8 <= 3))


def invert_code(c):
    return (7 - c % 8) * 8 + (7 - c // 8)


def window_value(codes):
    """codes[0] is the least significant base-64 digit (generator.m:84-87, 296-298); 0-based value."""
    v = 0
    for k, c in enumerate(codes):
        v += int(c) * 64 ** k
    return v


def inverse_value(v, fs):
    """generator.m:182-196: the window as seen upside down (digits inverted and reversed)."""
    digits = [(v // 64 ** j) % 64 for j in range(fs)]
    inv = [invert_code(d) for d in digits]
    return sum(inv[i] * 64 ** (fs - 1 - i) for i in range(fs))


def test_conflict(code, fs):
    """generator.m:288-334 -> True when every window of the dictionary and of its mirror image is unique."""
    code = np.asarray(code, dtype=np.int64)
    seen = set()
    n, m = code.shape
    mirror = np.array([[invert_code(int(c)) for c in row[::-1]] for row in code], dtype=np.int64)
    for mat in (code, mirror):
        for i in range(n):
            for j in range(m):
                v = window_value([mat[i, (j + k) % m] for k in range(fs)])
                if v in seen:
                    return False
                seen.add(v)
    return True


class Generator:
    def __init__(self, tag_col, fs, seed=0, node_budget=200000):
        self.col, self.fs = tag_col, fs
        self.rng = np.random.RandomState(seed)
        self.used = np.zeros(64 ** fs, dtype=bool)
        self.legal = [c for c in range(64) if legal_code(c)]
        # select(), generator.m:9-32: windows with an illegal digit or equal to their own mirror are never available
        for v in range(64 ** fs):
            digits = [(v // 64 ** j) % 64 for j in range(fs)]
            if not all(legal_code(d) for d in digits) or v == inverse_value(v, fs):
                self.used[v] = True
        self.budget = node_budget
        self.rows = []

    def _free(self, v):
        return not self.used[v] and not self.used[inverse_value(v, self.fs)]

    def _mark(self, vs, flag):
        for v in vs:
            self.used[v] = flag
            self.used[inverse_value(v, self.fs)] = flag

    def _dfs(self, row):
        fs, col = self.fs, self.col
        self.budget -= 1
        if self.budget < 0:
            return None
        if len(row) == col:
            return row
        if len(row) < col - 1:
            tail = row[len(row) - fs + 1:]
            wait = [c for c in self.legal if self._free(window_value(tail + [c]))]
            if not wait:
                return None
            # most onward options first, ties in random order (generator.m:104-137)
            score = []
            for c in wait:
                nxt = (tail + [c])[1:]
                score.append(sum(1 for d in self.legal if self._free(window_value(nxt + [d]))))
            order = sorted(range(len(wait)), key=lambda i: (-score[i], self.rng.rand()))
            for i in order:
                if score[i] == 0:
                    break
                v = window_value(tail + [wait[i]])
                self._mark([v], True)
                got = self._dfs(row + [wait[i]])
                if got is not None:
                    return got
                self._mark([v], False)
            return None
        # last column: the fs windows that wrap around the cyclic row must all be free and distinct (generator.m:150-178)
        for _ in range(100):
            c = self.legal[self.rng.randint(len(self.legal))]
            full = row + [c]
            cyc = [window_value([full[(j + k) % col] for k in range(fs)]) for j in range(col - fs, col)]
            inv = [inverse_value(v, fs) for v in cyc]
            if len(set(cyc)) == len(cyc) and all(self._free(v) for v in cyc) and not set(cyc) & set(inv):
                self._mark(cyc, True)
                return full
        return None

    def generate(self, tag_number):
        fs = self.fs
        attempts = 0
        while len(self.rows) < tag_number and self.budget > 0 and attempts < 50 * tag_number:
            attempts += 1
            free = np.flatnonzero(~self.used)
            if free.size == 0:
                break
            v0 = int(free[self.rng.randint(free.size)])
            if not self._free(v0):
                continue
            self._mark([v0], True)
            start = [(v0 // 64 ** j) % 64 for j in range(fs)]
            marked_before = self.used.copy()
            row = self._dfs(start)
            if row is None:
                self.used = marked_before
                self._mark([v0], False)
                continue
            self.rows.append(row)
        return np.array(self.rows, dtype=np.int32).reshape(-1, self.col)


def write_marker(path, code, fs):
    """the .marker text format CylinderTag::load_from_file reads: rows cols feature_size, then the codes."""
    code = np.asarray(code)
    with open(path, "w") as f:
        f.write("%d %d %d\n" % (code.shape[0], code.shape[1], fs))
        for row in code:
            f.write(" ".join(str(int(c)) for c in row) + "\n")


if __name__ == "__main__":
    col, fs, num, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    code = Generator(col, fs, seed).generate(num)
    print("rows generated: %d of %d; unique windows: %s" % (len(code), num, test_conflict(code, fs) if len(code) else None))
    write_marker(out, code, fs)
