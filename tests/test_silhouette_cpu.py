"""The argument k_silhouette_mask (cylindertag_amd/csrc/k_quad.hip, round 6) rests on, checked on the CPU with nothing but numpy / scipy: the silhouette
first hits of corner_detector.cpp:184-232 -- per row and column of a component's bounding box, the first and last pixel OF THE COMPONENT -- follow from the
threshold mask's row runs and ONE label probe per run, because a maximal horizontal run of foreground lies in one component, and a run cut by the box's edge
belongs to another component (the box of ours would otherwise reach further).  The kernel's own bit tricks are restated here on Python integers: run starts
`m & ~(m << 1)`, a run's mask `((m + lowbit) ^ m) & m` (the carry of the addition runs through the run and stops behind it)."""
import numpy as np
import pytest
from scipy import ndimage


def _first_last_by_pixels(lab, l, x0, y0, x1, y1):
    box = lab[y0:y1 + 1, x0:x1 + 1] == l
    rows = [(int(np.argmax(r)), int(len(r) - 1 - np.argmax(r[::-1]))) if r.any() else None for r in box]
    cols = [(int(np.argmax(c)), int(len(c) - 1 - np.argmax(c[::-1]))) if c.any() else None for c in box.T]
    return rows, cols


def _first_last_by_runs(mask, lab, l, x0, y0, x1, y1):
    """A row of the box as one integer (bit b = column x0 + b); a probe per run start; the component's own bits; then the extents."""
    w, h = x1 - x0 + 1, y1 - y0 + 1
    comp, probes = [], 0
    for y in range(y0, y1 + 1):
        m = 0
        for b in range(w):
            if mask[y, x0 + b]:
                m |= 1 << b
        c, starts = 0, m & ~(m << 1)
        while starts:
            sb = starts & -starts
            starts ^= sb
            run = ((m + sb) ^ m) & m
            probes += 1
            if lab[y, x0 + sb.bit_length() - 1] == l:  # the run's first pixel answers for the whole run
                c |= run
        comp.append(c)
    rows = [((c & -c).bit_length() - 1, c.bit_length() - 1) if c else None for c in comp]
    cols = []
    for b in range(w):
        ys = [y for y in range(h) if (comp[y] >> b) & 1]
        cols.append((ys[0], ys[-1]) if ys else None)
    return rows, cols, probes


@pytest.mark.parametrize("seed", range(6))
def test_row_runs_and_one_probe_per_run_give_the_silhouette(seed):
    rng = np.random.RandomState(seed)
    h, w = 90, 140
    # blobs of several scales: smooth noise thresholded (large interlocking shapes, boxes that contain other components' pixels), speckle on top
    f = ndimage.gaussian_filter(rng.rand(h, w), 2.5 + seed % 3)
    mask = f > np.percentile(f, 55)
    mask ^= rng.rand(h, w) < 0.02
    lab, n = ndimage.label(mask, structure=np.ones((3, 3), int))  # 8-connectivity, as the reference's connectedComponents
    checked = pixels = probes_total = 0
    for l, sl in enumerate(ndimage.find_objects(lab), start=1):
        y0, y1, x0, x1 = sl[0].start, sl[0].stop - 1, sl[1].start, sl[1].stop - 1
        if (lab == l).sum() < 12:
            continue
        want = _first_last_by_pixels(lab, l, x0, y0, x1, y1)
        rows, cols, probes = _first_last_by_runs(mask, lab, l, x0, y0, x1, y1)
        assert (rows, cols) == want, "component %d of seed %d" % (l, seed)
        assert all(r is not None for r in rows) and all(c is not None for c in cols)  # every row and column of a component's box holds one of its pixels
        checked += 1
        pixels += (y1 - y0 + 1) * (x1 - x0 + 1)
        probes_total += probes
    assert checked >= 5 and probes_total * 4 < pixels  # the point of it: far fewer probes than labels in the boxes
