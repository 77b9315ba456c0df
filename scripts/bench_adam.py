"""Fused Adam over an arena of the bench's size (1.29 G parameters), with and without the 16-bit copy of a slice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvml_goalnet_amd import ops

dev = "cuda:0"
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1286754368
p = torch.randn(n, device=dev) * 0.01; g = torch.randn(n, device=dev) * 1e-3
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
step = torch.zeros(1, dtype=torch.int64, device=dev)
sh_n = 512 * 70 * 70 * 512
shadow = torch.empty(sh_n, dtype=torch.bfloat16, device=dev)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)

t, ta = timeit(lambda: ops.adam_step_dev(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, step, step_bias=1))
print(f"adam            : {t:7.3f} ms (avg {ta:.3f})  {n * 28 / t / 1e6:7.0f} GB/s")
t, ta = timeit(lambda: ops.adam_step_dev_shadow(p, g, m, v, 1e-3, 0.9, 0.999, 1e-8, step, shadow, 1024 * 1024, step_bias=1))
print(f"adam + 16-bit copy of {sh_n / 1e9:.2f} G weights: {t:7.3f} ms (avg {ta:.3f})  {(n * 28 + sh_n * 2) / t / 1e6:7.0f} GB/s")
