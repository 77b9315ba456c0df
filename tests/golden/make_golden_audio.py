#!/usr/bin/env python3
"""Writes tests/golden/resample_*.npz: vectors for the per-row cubic resample of the reference's audio pre-processing,
/root/reference/utils.py:337-343, produced by running `scipy.interpolate.interp1d(kind='cubic', fill_value="extrapolate")`
ITSELF (scipy is importable in the build image; librosa is not, so the MFCC values fed to it here are synthetic rows of
MFCC-like range). Each fixture: rows (R, T) float32 inputs, B, out (R, B) float64 = what the reference's loop computes.

    python tests/golden/make_golden_audio.py
"""
import os

import numpy as np
from scipy.interpolate import interp1d

HERE = os.path.dirname(os.path.abspath(__file__))


def reference_rows(rows, b):
    out = []
    for f_idx in range(rows.shape[0]):                       # utils.py:336-343, verbatim call
        interpolator = interp1d(np.arange(rows.shape[1]), rows[f_idx, :], kind='cubic', fill_value="extrapolate")
        out.append(interpolator(np.linspace(0, rows.shape[1] - 1, b)))
    return np.array(out)


def main():
    rng = np.random.default_rng(12344321)
    cases = {"t44_b30": (30, 44, 30),        # one second at 22 050 Hz: T = 1 + 22050 // 512 = 44 -> B = 30 (skip_frames = 30)
             "t87_b60": (30, 87, 60),        # two seconds (skip_frames = 60, main.py:311)
             "t4_b30": (30, 4, 30),          # the shortest segment a cubic spline accepts
             "t5_b7": (7, 5, 7),
             "t30_b30": (30, 30, 30),        # T == B: the identity up to rounding
             "t23_b30": (30, 23, 30)}        # a clipped last slot: fewer frames than bins (upsampling)
    for name, (r, t, b) in cases.items():
        rows = (rng.standard_normal((r, t)) * 20.0).astype(np.float32)
        rows[0] -= 300.0                                      # coefficient 0 of an MFCC sits far from the others
        out = reference_rows(rows, b)
        np.savez(os.path.join(HERE, f"resample_{name}.npz"), rows=rows, b=np.array([b]), out=out)
        print(name, rows.shape, "->", out.shape, out.dtype)


if __name__ == "__main__":
    main()
