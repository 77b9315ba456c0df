"""Prints a hash of the 16-bit / fp32 outputs of the 256^2 phased tile (conv forward with bias + ReLU, data-gradient form
without) on whole and ragged shapes: run under two builds (GOALNET_LIB_PATH) the lines must be identical, because the
operand order of the MFMA changes neither the products nor the order of the K sum."""
import hashlib
import os
import sys

import torch

os.environ.setdefault("GOALNET_BF16_TILE", "256")          # every 16-bit contraction on the 256^2 phased tile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvml_goalnet_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1)
for h16 in (torch.bfloat16, torch.float16):
    # the last shape has THREE column tiles and > 256 tiles: a walking block meets a different tile column (another bias slice) every step
    for (n, hh, ww, cin, cout) in ((16, 56, 56, 256, 512), (3, 37, 29, 256, 512), (16, 112, 112, 64, 256), (5, 56, 56, 512, 256), (8, 72, 72, 64, 768)):
        x = torch.randn(n, hh, ww, cin, device=dev)
        sc = torch.ones(cin, device=dev)
        sh = torch.zeros(cin, device=dev)
        w = torch.randn(cout * 9 * cin, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        _, xp = ops.padded_bf16_alloc(n, hh, ww, cin, dev, dtype=h16)
        ops.to_bf16_padded(x, sc, sh, xp, n, hh, ww, cin)
        wb = ops.cast_bf16(w, torch.empty(w.shape, dtype=h16, device=dev))
        for relu, bias in ((True, b), (False, None)):
            out = []
            if ops.conv3x3_fwd_bf16p_o16_ok(n, hh, ww, cin, cout):
                y = torch.full((n, hh, ww, cout), 7.0, dtype=h16, device=dev)
                ops.conv3x3_fwd_bf16p_o16(xp, wb, bias, relu, y, n, hh, ww, cin, cout)
                out.append(y.view(torch.int16))
            y32 = torch.full((n, hh, ww, cout), 7.0, device=dev)
            ops.conv3x3_fwd_bf16p(xp, wb, bias, relu, y32, n, hh, ww, cin, cout)
            out.append(y32.view(torch.int32))
            torch.cuda.synchronize()
            hs = [hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:12] for o in out]
            print(str(h16)[6:], n, hh, ww, cin, cout, "relu" if relu else "raw", *hs, float(y32.abs().max()))


def h(*ts):
    torch.cuda.synchronize()
    return [hashlib.sha1(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12] for t in ts]


for h16 in (torch.bfloat16, torch.float16):
    # weight gradient (ROLE 2), whole and ragged pixel counts
    for (n, hh, ww, cin, cout) in ((8, 56, 56, 256, 512), (3, 37, 29, 64, 256)):
        x = torch.randn(n, hh, ww, cin, device=dev)
        dy = torch.randn(n, hh, ww, cout, device=dev)
        _, xp = ops.padded_bf16_alloc(n, hh, ww, cin, dev, dtype=h16)
        _, dyp = ops.padded_bf16_alloc(n, hh, ww, cout, dev, dtype=h16)
        ops.to_bf16_padded(x, torch.ones(cin, device=dev), torch.zeros(cin, device=dev), xp, n, hh, ww, cin)
        ops.to_bf16_padded(dy, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), dyp, n, hh, ww, cout)
        dw = torch.full((cout * 9 * cin,), 7.0, device=dev)
        ops.conv3x3_wgrad_bf16(xp, dyp, dw, n, hh, ww, cin, cout)
        print(str(h16)[6:], "wgrad", n, hh, ww, cin, cout, *h(dw), float(dw.abs().max()))
    # linear5's three contractions (ROLE 3, 4, 5)
    for (M, K, J) in ((512, 16384, 512), (300, 8256, 512)):
        x = torch.randn(M, K, device=dev).to(h16)
        w = (torch.randn(J, K, device=dev) * 0.02).to(h16)
        dy = torch.randn(M, J, device=dev).to(h16)
        b = torch.randn(J, device=dev)
        y = torch.full((M, J), 7.0, device=dev)
        ops.linear_fwd_bf16(x, w, b, y, relu=True)
        dx = torch.full((M, K), 7.0, device=dev)
        ops.linear_bwd_dx_bf16(dy, w, dx)
        out = [y, dx]
        if ops.linear_bwd_dx_bf16_o16_ok(M, K, J):
            dx16 = torch.full((M, K), 7.0, device=dev, dtype=h16)
            ops.linear_bwd_dx_bf16_o16(dy, w, dx16)
            out.append(dx16)
        dw = torch.full((J, K), 7.0, device=dev)
        ops.linear_bwd_dw_bf16(dy, x, dw)
        out.append(dw)
        print(str(h16)[6:], "linear", M, K, J, *h(*out), float(dw.abs().max()), float(dx.abs().max()))


# split operands (parts = 3: bf16 triples; parts = 2: scaled fp16 pairs, bias added after the unscaling from the per-tile LDS copy): three
# column tiles, > 256 tiles, bias + ReLU
for parts in (3, 2):
    dt = torch.bfloat16 if parts == 3 else torch.float16
    n, hh, ww, cin, cout = 8, 72, 72, 64, 768
    x = torch.randn(n, hh, ww, cin, device=dev)
    w = torch.randn(cout * 9 * cin, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    ax = aw = osc = None
    if parts == 2:
        ax = ops.absmax(x, torch.zeros(1, dtype=torch.int32, device=dev), n * hh * ww, cin)
        aw = ops.absmax(w, torch.zeros(1, dtype=torch.int32, device=dev), cout * 9, cin)
        osc = ops.split_scales(ax, aw)
    _, xp = ops.padded_bf16_alloc(n, hh, ww, parts * cin, dev, dtype=dt)
    ops.split_padded(parts, x, None, None, xp, n, hh, ww, cin, amax=ax)
    wsp = ops.split_rows(parts, w, torch.empty(cout * 9 * parts * cin, dtype=dt, device=dev), cout * 9, cin, amax=aw)
    y = torch.full((n, hh, ww, cout), 7.0, device=dev)
    ops.conv3x3_fwd_split(parts, xp, wsp, b, True, y, n, hh, ww, cin, cout, oscale=osc)
    print("split", parts, n, hh, ww, cin, cout, *h(y), float(y.abs().max()))
