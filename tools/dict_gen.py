#!/usr/bin/env python3
"""CylinderTag dictionary generator and strip writer -- restatement of /root/reference/CylinderTag_generator.m: select :9-32,
CylinderTagGenerator :34-59, dfs :61-191, inverse :193-206, plot_tag :208-219, draw :221-245, testConflict :247-286, for
SURVEY.md 8(f) rank 4.  Offline host tooling: it produces `.marker` files the detector consumes
(CylinderTag::load_from_file, CylinderTag.cpp:16-41); nothing on the detection path imports it.

A dictionary is `tag_number` cyclic rows of `tag_col` codes in 0..63 (code = 8*left_id + right_id, both ids in 0..7 with the
"long" flag id >= 4 equal on both sides: generator.m:17, 96).  Every cyclic window of `feature_size` consecutive codes, read as
a base-64 number, must be unique over the whole dictionary AND over its mirror image (row reversed, every code replaced by
(7 - c%8)*8 + (7 - c//8): what a strip looks like upside down) -- testConflict, generator.m:247-286.

The reference searches with MATLAB's global rand stream and a 20 s wall-clock limit, so its output is not reproducible; this
restatement keeps the search (depth-first, most-constrained-successor ordering with random tie breaks, generator.m:61-191)
but takes a seed and a node budget.  The pinned parts are the predicates: the reference's own CTag_2f12c.marker satisfies
`legal_code` and `test_conflict` (tests/test_dictgen_cpu.py).

The strip writer (plot_tag / draw) turns dictionary rows into the printable strips: write_strip_bmp().

usage: python tools/dict_gen.py <tag_col> <feature_size> <tag_number> <out.marker> [seed]
       python tools/dict_gen.py strips <in.marker> <out_dir> [tag_length] [margin]     (one cyN.bmp per row, generator.m:217)
"""
import sys

import numpy as np


def legal_code(c):
    """generator.m:17 (select) and :96, :114, :164 (dfs): both halves short (ids 0..3) or both long (ids 4..7)."""
    return not ((c % 8 <= 3 and c // 8 >= 4) or (c % 8 >= 4 and c // 8 <= 3))


def invert_code(c):
    return (7 - c % 8) * 8 + (7 - c // 8)


def window_value(codes):
    """codes[0] is the least significant base-64 digit (generator.m:251-254 in testConflict, :99-103 in dfs); 0-based value."""
    v = 0
    for k, c in enumerate(codes):
        v += int(c) * 64 ** k
    return v


def inverse_value(v, fs):
    """inverse, generator.m:193-206: the window as seen upside down (digits inverted and reversed)."""
    digits = [(v // 64 ** j) % 64 for j in range(fs)]
    inv = [invert_code(d) for d in digits]
    return sum(inv[i] * 64 ** (fs - 1 - i) for i in range(fs))


def test_conflict(code, fs):
    """testConflict, generator.m:247-286 -> True when every window of the dictionary and of its mirror image is unique."""
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
            # most onward options first, ties in random order (dfs, generator.m:111-139)
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
        # last column: the fs windows that wrap around the cyclic row must all be free and distinct (dfs, generator.m:161-189)
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


# ---------------------------------------------------------------------------------------------------------------
# printable strips: plot_tag (generator.m:208-219) + draw (:221-245)
# ---------------------------------------------------------------------------------------------------------------
DECODER = [(1.47, 0), (1.54, 0), (1.61, 0), (1.68, 0), (1.68, 1), (1.61, 1), (1.54, 1), (1.47, 1)]  # generator.m:223
WHITE_RATIO = 0.2                                                                                 # generator.m:224


def block_pos(half_id, tag_length):
    """draw, generator.m:227-234: centre of the white gap on one edge of a column -- the root of
    -p^2 + L p + (w/2 + w^2/4 - 0.2 cr) L^2 = 0 inside (0, L (1 - w)); the larger one for the "long" ids 4..7."""
    cr, use_max = DECODER[half_id]
    r = np.roots([-1.0, tag_length, (WHITE_RATIO / 2 + WHITE_RATIO ** 2 / 4 - 0.2 * cr) * tag_length ** 2])
    r = r[(r > 0) & (r < tag_length * (1 - WHITE_RATIO))]
    return float(r.max() if use_max else r.min())


def render_strip(row, tag_length=1200, ratio=15, margin=0):
    """plot_tag + draw for one dictionary row -> uint8 image (255 = paper, 0 = ink).  The strip is tag_length tall and
    1.5 * tag_length / ratio * len(row) wide (generator.m:212); column j holds two black quads between x = 1.5 L/ratio j and
    x + L/ratio, above and below the white gap of height 0.2 L whose centre runs from block_pos(left id) to block_pos(right id)
    (generator.m:243-244).  A pixel is inked when its centre lies inside a polygon ('SmoothEdges', false).  `margin` adds white
    paper around the strip (the reference writes none; a print has it, and detect() needs it: the outer threshold tiles of a
    frame are background)."""
    L = float(tag_length)
    h, w = int(tag_length), int(1.5 * tag_length / ratio * len(row))
    img = np.full((h, w), 255, np.uint8)
    yy = (np.arange(h) + 0.5)[:, None]
    for j, code in enumerate(row):
        pl, pr = block_pos(int(code) // 8, L), block_pos(int(code) % 8, L)
        x0, x1 = L / ratio * 1.5 * j, L / ratio * 1.5 * j + L / ratio
        xs = np.arange(int(np.floor(x0)), min(int(np.ceil(x1)) + 1, w))
        xc = xs + 0.5
        inside = (xc >= x0) & (xc <= x1)
        gap = pl + (pr - pl) * (xc - x0) / (x1 - x0)          # gap centre along the column
        ink = ((yy <= gap - L * WHITE_RATIO / 2) | (yy >= gap + L * WHITE_RATIO / 2)) & inside
        img[:, xs] = np.where(ink, 0, img[:, xs])
    if margin:
        out = np.full((h + 2 * margin[0], w + 2 * margin[1]) if isinstance(margin, tuple) else (h + 2 * margin, w + 2 * margin), 255, np.uint8)
        my, mx = margin if isinstance(margin, tuple) else (margin, margin)
        out[my:my + h, mx:mx + w] = img
        img = out
    return img


def write_bmp_gray(path, img):
    """8-bit palettised (gray ramp) bottom-up BMP, the format of the reference's test.bmp."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    pitch = (w + 3) & ~3
    rows = np.zeros((h, pitch), np.uint8)
    rows[:, :w] = img[::-1]
    hdr = b"BM" + (54 + 1024 + pitch * h).to_bytes(4, "little") + bytes(4) + (54 + 1024).to_bytes(4, "little")
    hdr += (40).to_bytes(4, "little") + w.to_bytes(4, "little") + h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (8).to_bytes(2, "little")
    hdr += bytes(4) + (pitch * h).to_bytes(4, "little") + bytes(16)
    pal = b"".join(bytes([i, i, i, 0]) for i in range(256))
    with open(path, "wb") as f:
        f.write(hdr + pal + rows.tobytes())


def write_strip_bmp(path, row, tag_length=1200, ratio=15, margin=0):
    write_bmp_gray(path, render_strip(row, tag_length, ratio, margin))


def read_marker(path):
    t = open(path).read().split()
    n, c, fs = int(t[0]), int(t[1]), int(t[2])
    return np.array([int(x) for x in t[3:3 + n * c]], np.int32).reshape(n, c), fs


if __name__ == "__main__":
    if sys.argv[1] == "strips":
        import os
        code, _ = read_marker(sys.argv[2])
        os.makedirs(sys.argv[3], exist_ok=True)
        tl = int(sys.argv[4]) if len(sys.argv) > 4 else 1200
        mg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
        for i, row in enumerate(code):
            write_strip_bmp(os.path.join(sys.argv[3], "cy%d.bmp" % (i + 1)), row, tl, 15, mg)  # generator.m:217
        print("wrote %d strips to %s" % (len(code), sys.argv[3]))
        sys.exit(0)
    col, fs, num, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    code = Generator(col, fs, seed).generate(num)
    print("rows generated: %d of %d; unique windows: %s" % (len(code), num, test_conflict(code, fs) if len(code) else None))
    write_marker(out, code, fs)
