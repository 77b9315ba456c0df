"""fp32 conv weight gradient alone at the bench shape (fewer frames): used for A/B builds (GOALNET_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvml_goalnet_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda:0"
torch.manual_seed(0)
for (h, w, cin, cout) in ((72, 72, 256, 512), (74, 74, 64, 256)):
    x = torch.randn(n, h, w, cin, device=dev); sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.1
    dy = torch.randn(n, h, w, cout, device=dev); dw = torch.empty(cout, 3, 3, cin, device=dev)
    fl = 2.0 * n * h * w * 9 * cin * cout
    fn = lambda: ops.conv3x3_wgrad(x, sc, sh, dy, dw, n, h, w, cin, cout)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = min(ts)
    print(f"wgrad {h}x{w} {cin}->{cout} {os.path.basename(os.environ.get('GOALNET_LIB_PATH', 'default')):>16s}: {t:7.3f} ms  {fl / t / 1e9:7.1f} TF/s")
