"""Deterministic input sequences shared by the tests and the golden generator."""
import numpy as np


def avi_substitute(img, n=64, rows=1080, cols=1920, seed=2):
    """Stand-in for the reference's missing test.avi: frame k is test.bmp shifted by
    an integer (dx, dy) in 0..7, cropped to rows x cols, with a fixed-point brightness gain in [0.9, 1.1]."""
    rng = np.random.RandomState(seed)
    out = np.empty((n, rows, cols), np.uint8)
    y_off = (img.shape[0] - rows) // 2 - 4
    for k in range(n):
        dx, dy = int(rng.randint(0, 8)), int(rng.randint(0, 8))
        gain = int(rng.randint(230, 283))  # /256
        crop = img[y_off + dy:y_off + dy + rows, :].astype(np.int32)
        crop = np.roll(crop, -dx, axis=1)
        out[k] = np.clip((crop * gain + 128) >> 8, 0, 255).astype(np.uint8)
    return out
