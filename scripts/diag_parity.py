"""Logit parity at 224x224, 16 frames, vs the CPU oracle: fp32 mode, bf16 mode with linear5 on the fp32 weight-streaming
path (default for <= 16 rows) and with linear5 forced onto the bf16 MFMA path."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvml_goalnet_amd import AVM, synth
from oracle import avm_ref

dev = "cuda:0"
n, h = 16, 224
params = synth.make_params(h, h, 30, True)
sd = {k: torch.from_numpy(v) for k, v in params.items()}
sd.update(avm_ref.init_buffers())
vis = torch.from_numpy(synth.make_visual(n, h, h)); aud = torch.from_numpy(synth.make_audio(n))
masks = [torch.from_numpy(m) for m in synth.make_drop_masks(n, step=0)]
inter = {}
with torch.no_grad():
    avm_ref.forward({k: v.clone() for k, v in sd.items() if v.is_floating_point() and "running" not in k},
                    avm_ref.init_buffers(), aud, vis, masks, True, inter)
ref = inter["logit"].view(-1)
out = {}
for name, prec, force in (("fp32", "fp32", None), ("bf16_lin5_fp32stream", "bf16", None), ("bf16_lin5_bf16", "bf16", "1")):
    if force:
        os.environ["GOALNET_FORCE_BF5"] = force
    else:
        os.environ.pop("GOALNET_FORCE_BF5", None)
    m = AVM(audio_included=True, device=dev, precision=prec)
    m.load_state_dict(sd)
    with torch.no_grad():
        m.forward_device(aud.to(dev), vis.to(dev), save=False)
    d = (m.last_logit.cpu() - ref).abs()
    out[name] = {"mae": d.mean().item(), "max": d.max().item()}
    del m
    torch.cuda.empty_cache()
print(json.dumps(out))
