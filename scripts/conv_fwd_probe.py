"""conv2 + conv3 forward of the bench workload, alone: the program `bench.py` puts under `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` (separate passes) to read the dominant kernel's HBM traffic live (roofline.traffic), and that
scripts/profile_bench.sh-style manual passes can use as well.

    python scripts/conv_fwd_probe.py --dtype f32|bf16|fp16 [--frames 1024] [--hw 224] [--reps 2]

Same entry points, operand layouts and sizes as AVM.forward_device (cvml_goalnet_amd/avm.py): fp32 = goalnet_conv3x3_fwd on the pooled
activation with the BatchNorm affine folded into the load; 16-bit = goalnet_conv3x3_fwd_bf16p_o16 on the zero-padded 16-bit operand.
Values are random: traffic does not depend on them.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvml_goalnet_amd import AVM, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = a.frames
    (_, _), (hp1, wp1), (hp2, wp2), _ = AVM._sizes(a.hw, a.hw)
    h16 = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(a.dtype)
    for (hh, ww, cin, cout) in ((hp1, wp1, 64, 256), (hp2, wp2, 256, 512)):
        x = torch.rand(n, hh, ww, cin, device=dev)
        sc = torch.rand(cin, device=dev) + 0.5
        sh = torch.rand(cin, device=dev) - 0.5
        w = (torch.rand(cout * 9 * cin, device=dev) - 0.5) * 0.05
        b = torch.rand(cout, device=dev)
        if h16 is None:
            y = torch.empty(n, hh, ww, cout, device=dev)
            for _ in range(a.reps):
                ops.conv3x3_fwd(x, sc, sh, w, b, True, y, n, hh, ww, cin, cout)
        else:
            _, xp = ops.padded_bf16_alloc(n, hh, ww, cin, dev, dtype=h16)
            ops.to_bf16_padded(x, sc, sh, xp, n, hh, ww, cin)
            wb = ops.cast_bf16(w, torch.empty(w.shape, dtype=h16, device=dev))
            o16 = ops.conv3x3_fwd_bf16p_o16_ok(n, hh, ww, cin, cout)
            y = torch.empty(n, hh, ww, cout, dtype=h16 if o16 else torch.float32, device=dev)
            for _ in range(a.reps):
                (ops.conv3x3_fwd_bf16p_o16 if o16 else ops.conv3x3_fwd_bf16p)(xp, wb, b, True, y, n, hh, ww, cin, cout)
        torch.cuda.synchronize()
        del x, y
    print("ok")


if __name__ == "__main__":
    main()
