"""How does the 16-bit MFMA round its fp32 accumulation? One dot product whose exact value is 1 + 0.75 ulp (and 1 + 0.25 ulp,
and their negatives), through goalnet_linear_fwd_bf16 (v_mfma_f32_32x32x16_bf16) and goalnet_linear_fwd (v_mfma_f32_32x32x2_f32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvml_goalnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
K, J, M = 64, 64, 4
ulp = 2.0 ** -23
for frac, a2, b2 in ((0.75, 3 * 2.0 ** -13, 2.0 ** -12), (0.25, 2.0 ** -13, 2.0 ** -12), (0.5, 2.0 ** -12, 2.0 ** -12)):
    for sign in (1.0, -1.0):
        for pos in (1, 17, 40):                                  # the small term in the same MFMA as the 1, in the next one, in a later one
            x = torch.zeros(M, K, device=dev)
            w = torch.zeros(J, K, device=dev)
            x[:, 0] = sign
            w[:, 0] = 1.0
            x[:, pos] = sign * a2
            w[:, pos] = b2
            y16 = torch.empty(M, J, device=dev)
            ops.linear_fwd_bf16(x.to(torch.bfloat16), w.to(torch.bfloat16), torch.zeros(J, device=dev), y16)
            y32 = torch.empty(M, J, device=dev)
            ops.linear_fwd(x, w, torch.zeros(J, device=dev), y32)
            exact = sign * (1.0 + frac * ulp)
            print(f"exact {sign:+.0f}(1 + {frac} ulp), small term at k={pos:2d}: bf16 MFMA -> {sign:+.0f}(1 + {(abs(y16[0, 0].item()) - 1) / ulp:.2f} ulp)   "
                  f"fp32 MFMA -> {sign:+.0f}(1 + {(abs(y32[0, 0].item()) - 1) / ulp:.2f} ulp)")
