"""Per-tile time line of the phased conv forward kernel from the in-kernel stamps of a -DGN_STAMPS build (GOALNET_LIB_PATH).
Stamps (100 MHz real-time counter), per tile: 0 tile start, 1 bias staged, 2 first operands landed, 3 main loop done, 4 next tile's prologue
issued + MFMAs drained, 5 stores issued. The launch is persistent: tile L + G follows tile L on the same CU (G = blocks launched)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cvml_goalnet_amd import ops, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
o16 = len(sys.argv) > 2 and sys.argv[2] == "o16"
h = w = 72; cin, cout = 256, 512
dev = "cuda:0"
x = torch.randn(n, h, w, cin, device=dev)
wb = (torch.randn(cout, 3, 3, cin, device=dev) * 0.05).to(torch.bfloat16); b = torch.randn(cout, device=dev)
y = torch.empty(n, h, w, cout, device=dev, dtype=torch.bfloat16 if o16 else torch.float32)
bx, xp = ops.padded_bf16_alloc(n, h, w, cin, dev); ops.to_bf16_padded(x, None, None, xp, n, h, w, cin)
fn = (lambda: ops.conv3x3_fwd_bf16p_o16(xp, wb, b, True, y, n, h, w, cin, cout)) if o16 else (lambda: ops.conv3x3_fwd_bf16p(xp, wb, b, True, y, n, h, w, cin, cout))
for _ in range(3): fn()
torch.cuda.synchronize()
nb = min(65536, n * h * w // 256 * (cout // 256))
buf = np.zeros(6 * nb, dtype=np.uint64)
lib = _lib.load()
rc = lib.goalnet_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb)
assert rc == 0, rc
t = buf.reshape(nb, 6).astype(np.float64) * 0.01      # microseconds
t0 = t[:, 0].min()
d = np.diff(t, axis=1)
names = ["tile start -> bias staged", "first operands land", "main loop", "next prologue issue + drain", "epilogue (stores issued)"]
print(f"{nb} blocks; kernel span {(t[:, 5].max() - t0):.1f} us; per-block total {(t[:,5]-t[:,0]).mean():.2f} us")
for i, nm in enumerate(names):
    print(f"  {nm:28s} mean {d[:, i].mean():7.2f} us   p10 {np.percentile(d[:, i], 10):7.2f}   p90 {np.percentile(d[:, i], 90):7.2f}")
G = int(os.environ.get("GOALNET_PERSISTENT", "256"))
if G > 1 and nb > G:
    per = t[G:, 0] - t[:-G, 0]
    gap = t[G:, 0] - t[:-G, 5]
    print(f"  tile period on a CU          mean {per.mean():7.2f} us   p10 {np.percentile(per, 10):7.2f}   p90 {np.percentile(per, 90):7.2f}")
    print(f"  stores issued -> next start  mean {gap.mean():7.2f} us")
# gap between a block's end and the start of the next block on the same CU: sort by start time, estimate from density
starts = np.sort(t[:, 0]); ends = np.sort(t[:, 5])
print(f"  blocks in flight (mean): {((t[:,5]-t[:,0]).sum() / (t[:,5].max() - t0)):.1f}")
