"""Shader clock while the 16-bit conv3 forward / the fp32 conv3 forward run back to back for a few seconds: rocm-smi samples from a
thread of this process (read-only). Answers: is the main loop of gemm_bf16_256.hip (1.26-1.6 us per K-tile = 2048 MFMA cycles per SIMD)
MFMA-bound at the clock the chip sustains?"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvml_goalnet_amd import AVM, ops  # noqa: E402

dev = torch.device("cuda", 0)
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            pw = re.findall(r"Power \(W\): ([\d.]+)", out)
            samples.append((time.time(), sclk[:1], pw[:1]))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), repr(e), None))
        time.sleep(0.2)


n, hh, ww, cin, cout = 1024, 72, 72, 256, 512
x = torch.relu(torch.randn(n, hh, ww, cin, device=dev))
sc = torch.ones(cin, device=dev); sh = torch.zeros(cin, device=dev)
w = torch.randn(cout * 9 * cin, device=dev) * 0.05
b = torch.randn(cout, device=dev)
_, xp = ops.padded_bf16_alloc(n, hh, ww, cin, dev)
ops.to_bf16_padded(x, sc, sh, xp, n, hh, ww, cin)
wb = ops.cast_bf16(w, torch.empty(w.shape, dtype=torch.bfloat16, device=dev))
y16 = torch.empty(n, hh, ww, cout, dtype=torch.bfloat16, device=dev)
y32 = torch.empty(n, hh, ww, cout, device=dev)
t = threading.Thread(target=sampler); t.start()
for label, fn, reps in (("idle", lambda: time.sleep(0.05), 30),
                        ("bf16 conv3 forward", lambda: ops.conv3x3_fwd_bf16p_o16(xp, wb, b, True, y16, n, hh, ww, cin, cout), 300),
                        ("fp32 conv3 forward", lambda: ops.conv3x3_fwd(x, sc, sh, w, b, True, y32, n, hh, ww, cin, cout), 40)):
    torch.cuda.synchronize(); t0 = time.time(); i0 = len(samples)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    s = samples[i0 + 2:]
    print(f"{label}: {ms:.3f} ms per launch over {time.time() - t0:.1f} s; sclk samples {[a for _, a, _ in s][:12]} power {[p for _, _, p in s][:12]}")
stop = True; t.join()
