"""Reader/checker for the tests/golden/*.npz fixtures (written FROM THE REFERENCE by make_golden.py)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


class Golden:
    """A tests/golden/*.npz fixture written by tests/golden/make_golden.py FROM THE REFERENCE."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.n = int(self.z["meta|n"][0]); self.h = int(self.z["meta|h"][0])
        self.steps = int(self.z["meta|steps"][0]); self.audio = bool(self.z["meta|audio"][0])
        self.drop = bool(self.z["meta|drop"][0])

    def keys(self, prefix):
        return sorted({k.split("|")[0] for k in self.z.files if k.startswith(prefix)})

    def check(self, key, t, rtol, atol=0.0, what=""):
        """Compare tensor `t` (torch, any device/layout, logical torch-native shape) with the stored summary.
        Returns the error measure used (for reporting)."""
        import torch
        from cvml_goalnet_amd import synth
        import zlib
        a = t.detach().to("cpu", torch.float64).reshape(-1).numpy()
        shape = tuple(int(x) for x in self.z[key + "|shape"])
        assert int(np.prod(shape)) == a.size, f"{key}: numel {a.size} vs golden shape {shape}"
        if key + "|full" in self.z.files:
            g = self.z[key + "|full"].astype(np.float64)
            scale = max(np.abs(g).max(), 1e-30)
            err = np.abs(a - g).max()
            assert err <= atol + rtol * scale, f"{self.name}:{key}{what}: max abs err {err:.3e} (scale {scale:.3e})"
            return err / scale
        stats = self.z[key + "|stats"]
        idx = synth.sample_indices(a.size, 16, zlib.crc32(key.encode()) & 0xFFFF)
        g = self.z[key + "|samples"].astype(np.float64)
        scale = max(stats[2], 1e-30)
        err = np.abs(a[idx] - g).max()
        assert err <= atol + rtol * scale, f"{self.name}:{key}{what}: sample err {err:.3e} (scale {scale:.3e})"
        # sum of squares: relative; plain sum: relative to sqrt(numel * sumsq) (it can cancel)
        ss = (a * a).sum()
        assert abs(ss - stats[1]) <= 4 * rtol * max(stats[1], 1e-30) + atol, f"{self.name}:{key}{what}: sumsq {ss} vs {stats[1]}"
        s = a.sum()
        tol_s = 4 * rtol * np.sqrt(a.size * max(stats[1], 1e-30)) + atol * a.size
        assert abs(s - stats[0]) <= tol_s, f"{self.name}:{key}{what}: sum {s} vs {stats[0]} (tol {tol_s})"
        amax = np.abs(a).max()
        assert abs(amax - stats[2]) <= rtol * scale + atol, f"{self.name}:{key}{what}: absmax {amax} vs {stats[2]}"
        return err / scale


GOLDEN_CASES_SMALL = ["avm_a1_n10_h40_p0", "avm_a1_n10_h40_mask3", "avm_a0_n10_h40_mask", "avm_a1_n1_h40_p0",
                      "avm_a1_n16_h40_mask", "avm_a0_n7_h52_p0"]
GOLDEN_CASES_BIG = ["avm_a1_n2_h224_p0"]


