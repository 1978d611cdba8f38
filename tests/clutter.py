"""Cluttered frames for the any-frame pass (tests/test_clutter_gpu.py, bench.py --clutter): content the reference processes without
complaint -- it keeps every component of 30 px .. 1 % of the half-size frame and walks them all (corner_detector.cpp:81-107,171-405)
-- but that needs more than the batch workspace's pools (cylindertag_amd/csrc/ctag_api.hip: make_caps).  Deterministic (seeded)."""
import numpy as np


def _stamp(img, y0, x0, shape, s, val):
    """One dark blob of about s x s full-resolution pixels: plus, T, L, U or a disc.  (At 19-21 pixels the T, L and U come out of
    edgeExtraction as non-quads -- their shoelace area misses the pixel count by more than threshold_RAC -- while a disc passes as a quad.)"""
    a = max(s // 3, 4)
    if shape == 0:      # plus
        img[y0 + (s - a) // 2:y0 + (s + a) // 2, x0:x0 + s] = val
        img[y0:y0 + s, x0 + (s - a) // 2:x0 + (s + a) // 2] = val
    elif shape == 1:    # T
        img[y0:y0 + a, x0:x0 + s] = val
        img[y0:y0 + s, x0 + (s - a) // 2:x0 + (s + a) // 2] = val
    elif shape == 2:    # L
        img[y0:y0 + s, x0:x0 + a] = val
        img[y0 + s - a:y0 + s, x0:x0 + s] = val
    elif shape == 3:    # U
        img[y0:y0 + s, x0:x0 + a] = val
        img[y0:y0 + s, x0 + s - a:x0 + s] = val
        img[y0 + s - a:y0 + s, x0:x0 + s] = val
    else:               # disc
        yy, xx = np.mgrid[0:s, 0:s]
        m = (yy - (s - 1) / 2.0) ** 2 + (xx - (s - 1) / 2.0) ** 2 <= (s / 2.0) ** 2
        img[y0:y0 + s, x0:x0 + s][m] = val


def blob_field(base, pitch_x=26, pitch_y=24, size=(19, 21), seed=5, margin=3, shapes=(1, 2, 3)):
    """`base` (a marker frame or a plain bright one) with a lattice of small dark non-quad blobs wherever the neighbourhood is bright:
    thousands of components of >= 30 half-resolution pixels next to whatever the frame held."""
    img = base.copy()
    rows, cols = img.shape
    rng = np.random.RandomState(seed)
    n = 0
    for y0 in range(margin + 3, rows - size[1] - margin - 3, pitch_y):
        for x0 in range(margin + 3 + (y0 // pitch_y % 2) * 2, cols - size[1] - margin - 3, pitch_x):
            s = int(rng.randint(size[0], size[1] + 1))
            if base[max(y0 - margin, 0):y0 + s + margin, max(x0 - margin, 0):x0 + s + margin].min() > 120:
                _stamp(img, y0, x0, int(shapes[rng.randint(0, len(shapes))]), s, int(rng.randint(10, 40)))
                n += 1
    return img, n


def chevron_texture(base, pitch=9, band=84, arm=40, width=4, ink=20, margin=4):
    """Nested thin chevrons ('<<<<'), band after band, wherever `base` is bright: fewer components than the candidate pool holds,
    but each reserves the perimeter of a large bounding box in the edge-cluster pool -- a texture that needs several hundred thousand
    cluster points."""
    img = base.copy()
    rows, cols = img.shape
    n = 0
    for y0 in range(6, rows - band - 6, band + 10):
        for x0 in range(8, cols - arm - width - 8, pitch):
            if base[max(y0 - margin, 0):y0 + band + margin, max(x0 - margin, 0):x0 + arm + width + margin].min() <= 120:
                continue
            for k in range(band // 2):
                x = x0 + arm - (k * arm) // (band // 2)
                img[y0 + k, x:x + width] = ink
                img[y0 + band - 1 - k, x:x + width] = ink
            n += 1
    return img, n


def long_diagonal(rows=2160, cols=3840, width=7, level=200, ink=20):
    """One thin dark band from corner to corner of a 4K frame: the longest boundary a component can have (its silhouette has at
    most 2 (w + h) pixels: 6000 at half resolution), with an area inside the reference's 1 % limit."""
    img = np.full((rows, cols), level, np.uint8)
    for y in range(40, rows - 40):
        x = 40 + (y - 40) * (cols - 80 - width) // (rows - 80)
        img[y, x:x + width] = ink
    return img


def tile_border_specks(base, tile_w=320, tile_h=30, ink=0, margin=4):
    """Isolated dark specks -- one half-size pixel each, every other pixel -- along all four borders of every 320 x 30 label tile of the
    half-size image, wherever `base` is bright: the second labelling pass publishes a component that touches its tile's border, so a
    1080p frame asks for ~350 pool entries per tile, 54 tiles of them more than its component pool (13 824) holds.  The reference
    labels such a frame like any other (every speck is a 1..2-pixel component below the area filter)."""
    img = base.copy()
    rows, cols = img.shape
    hy, hx = np.mgrid[0:rows // 2, 0:cols // 2]
    on_row = ((hy % tile_h == 0) | (hy % tile_h == tile_h - 1)) & (hx % 2 == 0)
    on_col = ((hx % tile_w == 0) | (hx % tile_w == tile_w - 1)) & (hy % 2 == 0)
    speck = on_row | on_col
    bright = base >= 140
    k = 2 * margin + 1
    # a speck only where its whole neighbourhood is bright (keeps clear of the markers)
    from numpy.lib.stride_tricks import sliding_window_view
    pad = np.pad(bright, margin, mode="edge")
    ok = sliding_window_view(pad, (k, k)).all(axis=(2, 3))
    ok_half = ok[0::2, 0::2] & ok[1::2, 1::2]
    speck &= ok_half[:rows // 2, :cols // 2]
    full = np.repeat(np.repeat(speck, 2, axis=0), 2, axis=1)
    img[:full.shape[0], :full.shape[1]][full] = ink
    return img
