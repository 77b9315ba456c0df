"""Run many fused train steps at the bench shape and print the loss trajectory (sanity: finite, decreasing on a fixed batch)."""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvml_goalnet_amd import AVM, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
torch.manual_seed(1234)
model = AVM(audio_included=True, device=dev, seed=synth.BASE_SEED, precision=prec)
aud, vis, lab = bench.make_inputs(1024, 224, 224, dev, synth.BASE_SEED)
losses = []
t0 = time.time()
for s in range(steps):
    loss, pred = model.train_step(aud, vis, lab)
    losses.append(loss)
torch.cuda.synchronize()
dt = time.time() - t0
ls = [float(l.item()) for l in losses]
print(json.dumps({"steps": steps, "precision": prec, "s_per_step": dt / steps, "loss_first": ls[:3], "loss_last": ls[-3:],
                  "all_finite": all(l == l and abs(l) < 1e9 for l in ls), "pred_range": [float(pred.min()), float(pred.max())]}))
