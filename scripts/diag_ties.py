"""Diagnostic: do HIP and the fp32 oracle pick different max-pool argmax positions (near-ties)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.nn.functional as F
from cvml_goalnet_amd import AVM, ops, synth
from oracle import avm_ref

n, h = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 40
params = synth.make_params(h, h)
m = AVM(True, device="cuda:0")
sd = {k: torch.from_numpy(v) for k, v in params.items()}; sd.update(avm_ref.init_buffers()); m.load_state_dict(sd)
vis = torch.from_numpy(synth.make_visual(n, h, h)); aud = torch.from_numpy(synth.make_audio(n))
out, ctx = m.forward_device(aud.cuda(), vis.cuda(), save=True)
for dt in (torch.float32, torch.float64):
    p = {k: torch.from_numpy(v).to(dt) for k, v in params.items()}
    inter = {}
    masks = [torch.from_numpy(x).to(dt) for x in synth.make_drop_masks(n)]
    avm_ref.forward(p, avm_ref.init_buffers(dt), aud.to(dt), vis.to(dt), masks, True, inter)
    for i in (1, 2, 3):
        y = inter[f"visbl.relu{i}"]
        pooled, pidx = F.max_pool2d(y, 3, 1, 0, return_indices=True)
        wc = y.shape[3]; hp, wp = pooled.shape[2], pooled.shape[3]
        ih, iw = pidx // wc, pidx % wc
        tap = ((ih - torch.arange(hp).view(1,1,hp,1)) * 3 + (iw - torch.arange(wp).view(1,1,1,wp))).to(torch.uint8)
        mine = ops.idx_to_nhwc(ctx[f"idx{i}"], *ctx[f"idx{i}"].shape).cpu().permute(0, 3, 1, 2)
        diff = (mine != tap)
        # top-2 gap per window
        u = F.unfold(y.reshape(-1, 1, y.shape[2], y.shape[3]), 3).transpose(1, 2)   # (N*C, L, 9)
        top2 = u.topk(2, dim=2).values
        gap = (top2[..., 0] - top2[..., 1]).reshape(pooled.shape)
        pos = pooled > 0
        print(f"{dt} block{i}: argmax mismatches {int(diff.sum())} of {diff.numel()}; "
              f"mismatches with max>0: {int((diff & pos).sum())}; min gap among max>0 windows {gap[pos].min().item():.3e}; "
              f"gaps at mismatching positive windows: {gap[diff & pos][:8].tolist()}")
