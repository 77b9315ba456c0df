"""ORACLE — test infrastructure only. Never imported by the product path.

CPU restatement (torch CPU ops, fp32 or fp64) of the reference's hot path, written from the spec in
SURVEY.md §8(a); it follows, and cites, these reference lines:

    VisBl.forward   /root/reference/utils.py:172-195   (layers utils.py:151-170)
    AudBl.forward   /root/reference/utils.py:214-227   (layers utils.py:203-211)
    AVM.forward     /root/reference/utils.py:260-272   (fusion utils.py:242-258)
    loss            /root/reference/main.py:68, 191    nn.MSELoss() on (n,1) vs (n,)  -> (n,n) broadcast
    train step      /root/reference/main.py:187-193    zero_grad, forward, loss, backward, Adam.step
    optimizer       /root/reference/main.py:70         optim.Adam(lr=1e-3), torch defaults

Who may use it (task rule ③): tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — as
the checker or as the timed CPU baseline, never as the thing shipped.

Pinning: the reference holds no tests, golden vectors or fixtures for this path, and
its arithmetic lives in PyTorch's ATen CPU kernels (unpinned by the repo; the report names torch
2.1.0, this image has 2.10.0). The restatement is pinned instead against outputs of the reference
itself, imported in the build container by tests/golden/make_golden.py (`utils.AVM` instance,
deterministic weights/inputs from cvml_goalnet_amd/synth.py); the resulting vectors are committed
under tests/golden/*.npz and checked by tests/test_oracle_golden.py.

The model is always in train mode (the reference never calls .eval()): BatchNorm
uses batch statistics and updates running stats on every forward; dropout is live. Dropout masks are
explicit inputs here (multipliers 0 or 1/(1-p)), `None` meaning p = 0.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.LazyBatchNorm2d default, utils.py:154
BN_MOMENTUM = 0.1
ADAM_LR = 1e-3      # main.py:51
ADAM_BETAS = (0.9, 0.999)
ADAM_EPS = 1e-8


def init_buffers(dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """BatchNorm buffers as a fresh reference instance holds them."""
    b = {}
    for i, c in ((1, 64), (2, 256), (3, 512)):
        b[f"visbl.bnorm{i}.running_mean"] = torch.zeros(c, dtype=dtype)
        b[f"visbl.bnorm{i}.running_var"] = torch.ones(c, dtype=dtype)
        b[f"visbl.bnorm{i}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return b


class _ForcedMaxPool(torch.autograd.Function):
    """MaxPool2d(3, 1) whose argmax positions are GIVEN (taps 0..8 = kh*3+kw inside each window).

    Max-pool routing is discontinuous: when the two largest values of a window are 1 ulp apart, two correct
    fp32 implementations of the preceding convolution can pick different positions, and every gradient upstream
    then differs at O(1) in a few elements. Parity tests therefore compare backward passes under the SAME routing
    decisions (the device's), and check separately that every disagreement with the natural argmax is such a
    near-tie (tests/test_gpu_avm.py)."""

    @staticmethod
    def forward(ctx, x, taps):
        n, c, h, w = x.shape
        hp, wp = h - 2, w - 2
        t = taps.long()
        flat = (torch.arange(hp).view(1, 1, hp, 1) + t // 3) * w + (torch.arange(wp).view(1, 1, 1, wp) + t % 3)
        ctx.save_for_backward(flat)
        ctx.shape = x.shape
        return x.reshape(n, c, h * w).gather(2, flat.reshape(n, c, -1)).reshape(n, c, hp, wp)

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        gi = torch.zeros(n, c, h * w, dtype=g.dtype).scatter_add_(2, flat.reshape(n, c, -1), g.reshape(n, c, -1))
        return gi.reshape(n, c, h, w), None


def natural_taps(y: torch.Tensor):
    """ATen's own argmax of MaxPool2d(3,1) on y (N,C,H,W) as taps 0..8, plus the gap between the largest and the
    second largest value of every window."""
    pooled, pidx = F.max_pool2d(y, 3, 1, 0, return_indices=True)
    w = y.shape[3]
    hp, wp = pooled.shape[2], pooled.shape[3]
    ih, iw = pidx // w, pidx % w
    taps = ((ih - torch.arange(hp).view(1, 1, hp, 1)) * 3 + (iw - torch.arange(wp).view(1, 1, 1, wp))).to(torch.uint8)
    u = F.unfold(y.reshape(-1, 1, y.shape[2], y.shape[3]), 3).transpose(1, 2)
    top2 = u.topk(2, dim=2).values
    gap = (top2[..., 0] - top2[..., 1]).reshape(pooled.shape)
    return taps, gap, pooled


def _vis_block(x, p, b, i, stride, pad, inter, taps=None, gate=None):
    """conv -> ReLU -> MaxPool(3,1) -> train-mode BatchNorm  (utils.py:174-187).
    taps / gate (tests only): the max-pool's argmax positions and the ReLU's gate AT those positions are given. ReLU is as
    discontinuous as the max-pool's routing: a window maximum within rounding error of zero is passed by one correct fp32
    convolution (y = +1e-9: the gradient flows) and blocked by another (y = -1e-9), and that one element then shows up at
    O(1) in the bias / weight gradients of the layer. Parity tests compare backward passes under the device's decisions and
    check separately that every decision that differs from the oracle's own is such a near-zero (tests/test_gpu_bench_shapes.py)."""
    pre = f"visbl.conv{i}"
    x = F.conv2d(x, p[pre + ".weight"], p[pre + ".bias"], stride=stride, padding=pad)
    if inter is not None:
        inter[pre] = x                                        # pre-activation
    if taps is not None and gate is not None:
        if inter is not None:
            inter[f"visbl.relu{i}"] = F.relu(x)
        x = _ForcedMaxPool.apply(x, taps) * gate.to(x.dtype)
    else:
        x = F.relu(x)
        if inter is not None:
            inter[f"visbl.relu{i}"] = x
        x = F.max_pool2d(x, kernel_size=3, stride=1, padding=0) if taps is None else _ForcedMaxPool.apply(x, taps)
    if inter is not None:
        inter[f"visbl.maxpool{i}"] = x
    bn = f"visbl.bnorm{i}"
    x = F.batch_norm(x, b[bn + ".running_mean"], b[bn + ".running_var"], p[bn + ".weight"], p[bn + ".bias"],
                     training=True, momentum=BN_MOMENTUM, eps=BN_EPS)
    b[bn + ".num_batches_tracked"] += 1
    if inter is not None:
        inter[bn] = x
    return x


def forward(p: Dict[str, torch.Tensor], b: Dict[str, torch.Tensor], audio, visual,
            drop_masks: Optional[List[torch.Tensor]] = None, audio_included: bool = True,
            inter: Optional[dict] = None, pool_taps: Optional[dict] = None, head: str = "regression",
            relu_gates: Optional[dict] = None) -> torch.Tensor:
    """AVM.forward(audio_input, visual_input) -> (N,1) in (1,5).  utils.py:260-272.

    `b` (BN running stats) is updated in place, as the reference's train-mode forward does even under
    no_grad. `drop_masks`: [visbl.drop5, fusion.2, fusion.5, fusion.8, fusion.11] multipliers or None.
    `inter`: optional dict that receives named intermediate activations.
    `pool_taps`: optional {1,2,3 -> uint8 (N,C,Hp,Wp)} argmax positions to force in the three max-pools
    (tests only, see _ForcedMaxPool); None = the reference's own behaviour. `relu_gates`: optional {1,2,3 -> bool (N,C,Hp,Wp)}, with
    pool_taps: whether the ReLU passes at each forced argmax position (tests only, see _vis_block); optional keys "visbl.linear5",
    "fusion.0" / ".3" / ".6" / ".9" -> bool (N, width): the gate of that layer's ReLU (tests of the 16-bit modes, whose 4e-3
    activation noise flips ~0.5 % of a ReLU network's gates: the gradient's relative error is then ~sqrt(0.5 %), whatever the kernels do).
    """
    pt = pool_taps or {}
    rg = relu_gates or {}
    dm = drop_masks if drop_masks is not None else [None] * 5

    def drop(x, m):
        return x if m is None else x * m

    # VisBl, utils.py:172-195
    x = _vis_block(visual, p, b, 1, 3, 3, inter, pt.get(1), rg.get(1))
    x = _vis_block(x, p, b, 2, 1, 1, inter, pt.get(2), rg.get(2))
    x = _vis_block(x, p, b, 3, 1, 1, inter, pt.get(3), rg.get(3))
    def relu_g(z, key):
        """ReLU, or (tests) the given gate at this layer: z * gate — see _vis_block on why gates are forced"""
        g = rg.get(key)
        return F.relu(z) if g is None else z * g.to(z.dtype)

    x = torch.flatten(x, 1)                                   # NCHW flatten: c*H*W + h*W + w
    x = relu_g(F.linear(x, p["visbl.linear5.weight"], p["visbl.linear5.bias"]), "visbl.linear5")
    v = drop(x, dm[0])
    if inter is not None:
        inter["visbl.drop5"] = v

    if audio_included:
        # AudBl, utils.py:214-227
        a = F.relu(F.conv1d(audio, p["audbl.conv1.weight"], p["audbl.conv1.bias"], stride=2, padding=1))
        a = F.relu(F.conv1d(a, p["audbl.conv2.weight"], p["audbl.conv2.bias"], stride=2, padding=1))
        a = torch.flatten(a, 1)
        a = F.relu(F.linear(a, p["audbl.linear3.weight"], p["audbl.linear3.bias"]))
        if inter is not None:
            inter["audbl.relu3"] = a
        x = torch.cat((a, v), dim=-1)                         # audio FIRST, utils.py:266
    else:
        x = v

    # fusion, utils.py:242-258
    for li, k in enumerate((0, 3, 6, 9)):
        x = relu_g(F.linear(x, p[f"fusion.{k}.weight"], p[f"fusion.{k}.bias"]), f"fusion.{k}")
        x = drop(x, dm[1 + li])
        if inter is not None:
            inter[f"fusion.{k + 2}"] = x
    z = F.linear(x, p["fusion.12.weight"], p["fusion.12.bias"])
    if inter is not None:
        inter["logit"] = z                                    # pre-sigmoid / pre-softmax
    if head == "classifier":
        # EXTENSION — the reference's commented-out variant: nn.Softmax(dim = 1) in place of nn.Sigmoid (utils.py:257) on a
        # Linear(128 -> C); `output = 4 * output + 1` (utils.py:270) is live code and applies to either head
        return 4 * torch.softmax(z, dim=1) + 1
    return 4 * torch.sigmoid(z) + 1                           # utils.py:270


def mse_bcast(pred: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss()(pred(n,1), labels(n,)): broadcast to (n,n), mean over n^2 terms.  main.py:191."""
    d = pred - labels            # (n,1) - (n,) -> (n,n)
    return (d * d).mean()


def ce_loss(pred: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """EXTENSION (comment-only in the reference): nn.CrossEntropyLoss()(pred (n,C), (labels - 1).long()).  main.py:69, 96, 189."""
    return F.cross_entropy(pred, (labels - 1).long())


def adam_step(p: Dict[str, torch.Tensor], g: Dict[str, torch.Tensor], state: dict,
              lr=ADAM_LR, betas=ADAM_BETAS, eps=ADAM_EPS) -> None:
    """torch.optim.Adam defaults (no weight decay, no amsgrad), in place.  main.py:70, 193.
    Same operation order as torch's single-tensor path."""
    b1, b2 = betas
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for k, w in p.items():
        if k not in g or g[k] is None:
            continue
        m = state.setdefault("m." + k, torch.zeros_like(w))
        v = state.setdefault("v." + k, torch.zeros_like(w))
        m.lerp_(g[k], 1.0 - b1)
        v.mul_(b2).addcmul_(g[k], g[k], value=1.0 - b2)
        denom = (v.sqrt() / bc2_sqrt).add_(eps)
        w.addcdiv_(m, denom, value=-step_size)


def train_step(p, b, state, audio, visual, labels, drop_masks=None, audio_included=True, inter=None, pool_taps=None,
               head="regression", relu_gates=None):
    """One sub-batch train step, main.py:187-193. Returns (loss, pred, grads). `p` is updated in place."""
    leaf = {k: v.detach().requires_grad_(True) for k, v in p.items()}
    pred = forward(leaf, b, audio, visual, drop_masks, audio_included, inter, pool_taps, head, relu_gates)
    loss = ce_loss(pred, labels) if head == "classifier" else mse_bcast(pred, labels)
    names = list(leaf.keys())
    grads = torch.autograd.grad(loss, [leaf[k] for k in names], allow_unused=True)
    g = dict(zip(names, grads))
    with torch.no_grad():
        adam_step(p, g, state)
    return loss.detach(), pred.detach(), g


def macs_per_frame(h: int, w: int, bins: int = 30, audio_included: bool = True) -> dict:
    """Forward multiply-accumulates per frame, per layer group."""
    h1, w1 = (h + 3) // 3 + 1, (w + 3) // 3 + 1
    p1 = (h1 - 2, w1 - 2)
    p2 = (p1[0] - 2, p1[1] - 2)
    p3 = (p2[0] - 2, p2[1] - 2)
    l1 = (bins - 1) // 2 + 1
    l2 = (l1 - 1) // 2 + 1
    d = {
        "conv1": h1 * w1 * 64 * 27,
        "conv2": p1[0] * p1[1] * 256 * 576,
        "conv3": p2[0] * p2[1] * 512 * 2304,
        "linear5": 512 * 512 * p3[0] * p3[1],
        "audbl": (l1 * 64 * 90 + l2 * 128 * 192 + 128 * 128 * l2) if audio_included else 0,
        "fusion": (640 if audio_included else 512) * 512 + 512 * 512 + 512 * 256 + 256 * 128 + 128,
    }
    d["total"] = sum(d.values())
    return d
