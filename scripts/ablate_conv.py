"""conv3 forward (bf16, padded input) alone at the bench shape: used for A/B builds (GOALNET_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvml_goalnet_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = w = 72; cin = int(sys.argv[2]) if len(sys.argv) > 2 else 256; cout = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = "cuda:0"
torch.manual_seed(0)
x = torch.randn(n, h, w, cin, device=dev)
wb = (torch.randn(cout, 3, 3, cin, device=dev) * 0.05).to(torch.bfloat16); b = torch.randn(cout, device=dev)
y = torch.empty(n, h, w, cout, device=dev)
bx, xp = ops.padded_bf16_alloc(n, h, w, cin, dev); ops.to_bf16_padded(x, None, None, xp, n, h, w, cin)
fl = 2.0 * n * h * w * 9 * cin * cout
fn = lambda: ops.conv3x3_fwd_bf16p(xp, wb, b, True, y, n, h, w, cin, cout)
for _ in range(3): fn()
torch.cuda.synchronize()
ts = []
for _ in range(6):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
t = min(ts)
print(f"cin={cin} cout={cout} ktiles={9 * cin // 64} tiles/CU={n * h * w // 256 * (cout // 256) / 256:.1f} " f"{os.path.basename(os.environ.get('GOALNET_LIB_PATH', 'default')):>32s}: {t:7.3f} ms  {fl / t / 1e9:7.1f} TF/s  (avg {fl / (sum(ts) / len(ts)) / 1e9:.1f})")
