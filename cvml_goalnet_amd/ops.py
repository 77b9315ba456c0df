"""Thin torch-tensor wrappers over the C ABI (include/goalnet_hip.h).

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and
torch.distributed. Every arithmetic operation of the hot path runs in libgoalnet_hip.so.
Tensors are fp32, on the GPU, in the device layouts named in the header (NHWC / OHWI).
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from ._lib import STAT_PARTS, check

F32 = torch.float32


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.GoalnetError("expected a GPU tensor")
        if t.dtype not in (F32, torch.float64, torch.uint8, torch.bfloat16, torch.float16, torch.int32):
            raise _lib.GoalnetError(f"unexpected dtype {t.dtype}")


def _req(cond, msg):
    """argument / shape checks raise RuntimeError (what ATen raises for a shape mismatch in the reference, SURVEY.md §8(b)
    "Ownership / errors"); unlike `assert` they survive `python -O`"""
    if not cond:
        raise RuntimeError(msg)


def _ld(t):
    """leading dimension (row stride) of a 2-D view with unit column stride"""
    _req(t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), "_ld: argument check failed: t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1)")
    return t.stride(0)


def lib():
    return _lib.load()


# ---------------------------------------------------------------------------------------------
def fill_uniform(dst, seed, tensor_id, lo, hi):
    _chk(dst)
    _req(dst.is_contiguous() and dst.dtype == F32, "fill_uniform: argument check failed: dst.is_contiguous() and dst.dtype == F32")
    check(lib().goalnet_fill_uniform(dst.data_ptr(), dst.numel(), seed, tensor_id, lo, hi, _s()), "fill_uniform")
    return dst


def dropout_mask(dst, seed, tensor_id, p):
    _chk(dst)
    _req(dst.is_contiguous() and dst.dtype == F32, "dropout_mask: argument check failed: dst.is_contiguous() and dst.dtype == F32")
    check(lib().goalnet_dropout_mask(dst.data_ptr(), dst.numel(), seed, tensor_id, p, _s()), "dropout_mask")
    return dst


def transpose_inner(src, dst, B, R, C):
    """[B][R][C] -> [B][C][R] on contiguous buffers"""
    _chk(src, dst)
    _req(src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel() == B * R * C, "transpose_inner: argument check failed: src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel() == B * R * C")
    check(lib().goalnet_transpose_inner(src.data_ptr(), dst.data_ptr(), B, R, C, _s()), "transpose_inner")
    return dst


def conv3x3_weight_flip(w, wt, cout, cin):
    _chk(w, wt)
    _req(w.numel() == wt.numel() == cout * cin * 9, "conv3x3_weight_flip: argument check failed: w.numel() == wt.numel() == cout * cin * 9")
    check(lib().goalnet_conv3x3_weight_flip(w.data_ptr(), wt.data_ptr(), cout, cin, _s()), "conv3x3_weight_flip")
    return wt


def conv3x3_weight_flip2(wa, wta, couta, cina, wb, wtb, coutb, cinb):
    _chk(wa, wta, wb, wtb)
    _req(wa.numel() == wta.numel() == couta * cina * 9 and wb.numel() == wtb.numel() == coutb * cinb * 9, "conv3x3_weight_flip2: sizes")
    check(lib().goalnet_conv3x3_weight_flip2(wa.data_ptr(), wta.data_ptr(), couta, cina, wb.data_ptr(), wtb.data_ptr(), coutb, cinb, _s()),
          "conv3x3_weight_flip2")


def conv1_fwd(x_nchw, w, b, y, N, H, W):
    _chk(x_nchw, w, b, y)
    _req(x_nchw.is_contiguous() and x_nchw.numel() == N * 3 * H * W, "conv1_fwd: argument check failed: x_nchw.is_contiguous() and x_nchw.numel() == N * 3 * H * W")
    Ho, Wo = (H + 3) // 3 + 1, (W + 3) // 3 + 1
    _req(y.numel() == N * Ho * Wo * 64 and w.numel() == 64 * 27 and b.numel() == 64, "conv1_fwd: argument check failed: y.numel() == N * Ho * Wo * 64 and w.numel() == 64 * 27 and b.numel() == 64")
    check(lib().goalnet_conv1_fwd(x_nchw.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, H, W, _s()), "conv1_fwd")
    return y


def conv1_wgrad(x_nchw, dy, dw, db, N, H, W):
    _chk(x_nchw, dy, dw, db)
    Ho, Wo = (H + 3) // 3 + 1, (W + 3) // 3 + 1
    _req(dy.numel() == N * Ho * Wo * 64 and dw.numel() == 64 * 27 and db.numel() == 64, "conv1_wgrad: argument check failed: dy.numel() == N * Ho * Wo * 64 and dw.numel() == 64 * 27 and db.numel() == 64")
    nbytes = lib().goalnet_conv1_wgrad_ws_bytes(N, H, W)
    ws = torch.empty(nbytes // 4, dtype=F32, device=dy.device)
    check(lib().goalnet_conv1_wgrad(x_nchw.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nbytes,
                                    N, H, W, _s()), "conv1_wgrad")


def stat_parts(units):
    """rows of a partial-sum buffer for `units` independent work items (frames): min(units, STAT_PARTS)"""
    return max(1, min(int(units), STAT_PARTS))


def _rows(partials, width):
    """the partial-sum kernels take the row count from the buffer the caller allocated: [nparts][width] doubles"""
    _req(partials.dtype == torch.float64 and partials.is_contiguous() and partials.numel() % width == 0, "_rows: argument check failed: partials.dtype == torch.float64 and partials.is_contiguous() and partials.numel() % width == 0")
    n = partials.numel() // width
    _req(1 <= n <= STAT_PARTS, "_rows: argument check failed: 1 <= n <= STAT_PARTS")
    return n


def idx_to_nhwc(idx, N, Hp, Wp, C):
    """argmax bytes as the kernels store them ([N][C/S][Hp][Wp][S], S = 32 or C; include/goalnet_hip.h) -> (N, Hp, Wp, C)"""
    S = 32 if C % 32 == 0 else C
    return idx.reshape(N, C // S, Hp, Wp, S).permute(0, 2, 3, 1, 4).reshape(N, Hp, Wp, C)


def pool_bnstats_fwd(y, p, idx, partials, N, Hc, Wc, C):
    _chk(y, p, idx, partials)
    _req(y.numel() == N * Hc * Wc * C and p.numel() == N * (Hc - 2) * (Wc - 2) * C, "pool_bnstats_fwd: argument check failed: y.numel() == N * Hc * Wc * C and p.numel() == N * (Hc - 2) * (Wc - 2) * C")
    _req(idx is None or (idx.dtype == torch.uint8 and idx.numel() == p.numel()), "pool_bnstats_fwd: argument check failed: idx is None or (idx.dtype == torch.uint8 and idx.numel() == p.numel())")
    if p.dtype in H16:                              # pooled activation stored in 16 bits (statistics of the stored values)
        check(lib().goalnet_pool_bnstats_fwd_p16(y.data_ptr(), int(y.dtype in H16), p.data_ptr(), _p(idx), partials.data_ptr(),
                                                 _rows(partials, 2 * C), N, Hc, Wc, C, _f16(y, p), _s()), "pool_bnstats_fwd_p16")
        return
    _req(y.dtype == F32, "pool_bnstats_fwd: argument check failed: y.dtype == F32")
    check(lib().goalnet_pool_bnstats_fwd(y.data_ptr(), p.data_ptr(), _p(idx), partials.data_ptr(), _rows(partials, 2 * C),
                                         N, Hc, Wc, C, _s()), "pool_bnstats_fwd")


def bn_finalize(partials, gamma, beta, rmean, rvar, momentum, eps, count, C, mean, invstd, scale, shift):
    _chk(partials, gamma, beta, rmean, rvar, mean, invstd, scale, shift)
    for t in (gamma, beta, mean, invstd, scale, shift):
        _req(t.numel() == C and t.is_contiguous(), "bn_finalize: argument check failed: t.numel() == C and t.is_contiguous()")
    check(lib().goalnet_bn_finalize(partials.data_ptr(), _rows(partials, 2 * C), gamma.data_ptr(), beta.data_ptr(), _p(rmean), _p(rvar), momentum, eps,
                                    count, C, mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), _s()),
          "bn_finalize")


def bn_bwd_reduce(dz, p, mean, invstd, partials, npix, C):
    """dz: fp32, or bf16 as the *_o16 GEMMs write it; p: fp32, or bf16 as pool_bnstats_fwd stores it into a bf16 tensor"""
    _chk(dz, p, mean, invstd, partials)
    _req(dz.numel() == p.numel() == npix * C, "bn_bwd_reduce: argument check failed: dz.numel() == p.numel() == npix * C")
    if dz.dtype in H16 or p.dtype in H16:
        check(lib().goalnet_bn_bwd_reduce_t(dz.data_ptr(), int(dz.dtype in H16), p.data_ptr(), int(p.dtype in H16),
                                            mean.data_ptr(), invstd.data_ptr(), partials.data_ptr(), _rows(partials, 2 * C), npix, C,
                                            _f16(dz, p), _s()), "bn_bwd_reduce_t")
        return
    check(lib().goalnet_bn_bwd_reduce(dz.data_ptr(), p.data_ptr(), mean.data_ptr(), invstd.data_ptr(), partials.data_ptr(),
                                      _rows(partials, 2 * C), npix, C, _s()), "bn_bwd_reduce")


def bn_bwd_finalize(partials, gamma, mean, invstd, count, C, dgamma, dbeta, coef3):
    _chk(partials, gamma, mean, invstd, dgamma, dbeta, coef3)
    _req(coef3.numel() == 3 * C and dgamma.numel() == C and dbeta.numel() == C, "bn_bwd_finalize: argument check failed: coef3.numel() == 3 * C and dgamma.numel() == C and dbeta.numel() == C")
    check(lib().goalnet_bn_bwd_finalize(partials.data_ptr(), _rows(partials, 2 * C), gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(), count, C,
                                        dgamma.data_ptr(), dbeta.data_ptr(), coef3.data_ptr(), _s()), "bn_bwd_finalize")


def bnpool_bwd(dz, p, idx, coef3, dy, dbias_partials, N, Hc, Wc, C):
    _chk(dz, p, idx, coef3, dy, dbias_partials)
    npool = N * (Hc - 2) * (Wc - 2) * C
    _req(dz.numel() == p.numel() == idx.numel() == npool and dy.numel() == N * Hc * Wc * C, "bnpool_bwd: argument check failed: dz.numel() == p.numel() == idx.numel() == npool and dy.numel() == N * Hc * Wc * C")
    if dz.dtype in H16 or p.dtype in H16:
        check(lib().goalnet_bnpool_bwd_bf16p_t(dz.data_ptr(), int(dz.dtype in H16), p.data_ptr(), int(p.dtype in H16),
                                               idx.data_ptr(), coef3.data_ptr(), dy.data_ptr(), None, dbias_partials.data_ptr(),
                                               _rows(dbias_partials, C), N, Hc, Wc, C, _f16(dz, p), _s()), "bnpool_bwd_bf16p_t")
        return
    check(lib().goalnet_bnpool_bwd(dz.data_ptr(), p.data_ptr(), idx.data_ptr(), coef3.data_ptr(), dy.data_ptr(),
                                   dbias_partials.data_ptr(), _rows(dbias_partials, C), N, Hc, Wc, C, _s()), "bnpool_bwd")


def bnpool_bwd_bf16p(dz, p, idx, coef3, dy, dypad, dbias_partials, N, Hc, Wc, C):
    _chk(dz, p, idx, coef3, dy, dypad, dbias_partials)
    _req(dypad.dtype in H16 and dypad.numel() >= N * (Hc + 2) * (Wc + 2) * C, "bnpool_bwd_bf16p: argument check failed: dypad.dtype in H16 and dypad.numel() >= N * (Hc + 2) * (Wc + 2) * C")
    if dz.dtype in H16 or p.dtype in H16:
        check(lib().goalnet_bnpool_bwd_bf16p_t(dz.data_ptr(), int(dz.dtype in H16), p.data_ptr(), int(p.dtype in H16),
                                               idx.data_ptr(), coef3.data_ptr(), _p(dy), dypad.data_ptr(), dbias_partials.data_ptr(),
                                               _rows(dbias_partials, C), N, Hc, Wc, C, _f16(dz, p, dypad), _s()), "bnpool_bwd_bf16p_t")
        return
    check(lib().goalnet_bnpool_bwd_bf16p(dz.data_ptr(), p.data_ptr(), idx.data_ptr(), coef3.data_ptr(), _p(dy),
                                         dypad.data_ptr(), dbias_partials.data_ptr(), _rows(dbias_partials, C), N, Hc, Wc, C,
                                         _f16(dypad), _s()), "bnpool_bwd_bf16p")


SMALL_BN_ELEMS = 1 << 20       # conv outputs of up to this many elements (C <= 512) take the small-shape pool / BatchNorm kernels


def pool_bn_fwd_small(y, p, idx, gamma, beta, rmean, rvar, momentum, eps, st, N, Hc, Wc, C):
    """pool_bnstats_fwd + bn_finalize in one launch (small fp32 shapes); st = (4, C): mean, invstd, scale, shift"""
    _chk(y, p, idx, gamma, beta, rmean, rvar, st)
    _req(y.dtype == F32 and p.dtype == F32 and y.numel() == N * Hc * Wc * C and p.numel() == N * (Hc - 2) * (Wc - 2) * C,
         "pool_bn_fwd_small: fp32 y (N,Hc,Wc,C) and p (N,Hc-2,Wc-2,C)")
    _req(idx is None or (idx.dtype == torch.uint8 and idx.numel() == p.numel()), "pool_bn_fwd_small: idx must be uint8, one per pooled element")
    _req(st.numel() == 4 * C and st.is_contiguous(), "pool_bn_fwd_small: st must be (4, C)")
    nbytes = lib().goalnet_bn_small_ws_bytes(C)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=y.device)
    ctr = _tile_counters(("pool_bn_fwd", N, Hc, Wc, C), y.device)
    check(lib().goalnet_pool_bn_fwd_small(y.data_ptr(), p.data_ptr(), _p(idx), gamma.data_ptr(), beta.data_ptr(), _p(rmean), _p(rvar),
                                          momentum, eps, st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(),
                                          ws.data_ptr(), nbytes, ctr.data_ptr(), N, Hc, Wc, C, _s()), "pool_bn_fwd_small")


def bn_bwd_reduce_small(dz, p, mean, invstd, gamma, dgamma, dbeta, coef3, N, Hc, Wc, C):
    """bn_bwd_reduce + bn_bwd_finalize in one launch (small fp32 shapes)"""
    _chk(dz, p, mean, invstd, gamma, dgamma, dbeta, coef3)
    npool = N * (Hc - 2) * (Wc - 2) * C
    _req(dz.dtype == F32 and p.dtype == F32 and dz.numel() == p.numel() == npool, "bn_bwd_reduce_small: fp32 dz / p of the pooled shape")
    _req(dgamma.numel() == C and dbeta.numel() == C and coef3.numel() == 3 * C, "bn_bwd_reduce_small: dgamma, dbeta (C), coef3 (3 C)")
    nbytes = lib().goalnet_bn_small_ws_bytes(C)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=dz.device)
    ctr = _tile_counters(("bn_bwd_reduce", N, Hc, Wc, C), dz.device)
    check(lib().goalnet_bn_bwd_reduce_small(dz.data_ptr(), p.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), dgamma.data_ptr(),
                                            dbeta.data_ptr(), coef3.data_ptr(), ws.data_ptr(), nbytes, ctr.data_ptr(), N, Hc, Wc, C, _s()),
          "bn_bwd_reduce_small")


def bnpool_bwd_small(dz, p, idx, coef3, dy, dbias, N, Hc, Wc, C):
    """bnpool_bwd + the conv bias gradient in one launch (small fp32 shapes; a direct 9-window gather)"""
    _chk(dz, p, idx, coef3, dy, dbias)
    npool = N * (Hc - 2) * (Wc - 2) * C
    _req(dz.dtype == F32 and p.dtype == F32 and dz.numel() == p.numel() == idx.numel() == npool and dy.numel() == N * Hc * Wc * C and dbias.numel() == C,
         "bnpool_bwd_small: fp32 dz / p / idx of the pooled shape, dy of the conv shape, dbias (C)")
    nbytes = lib().goalnet_bn_small_ws_bytes(C)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=dz.device)
    ctr = _tile_counters(("bnpool_bwd", N, Hc, Wc, C), dz.device)
    check(lib().goalnet_bnpool_bwd_small(dz.data_ptr(), p.data_ptr(), idx.data_ptr(), coef3.data_ptr(), dy.data_ptr(), dbias.data_ptr(),
                                         ws.data_ptr(), nbytes, ctr.data_ptr(), N, Hc, Wc, C, _s()), "bnpool_bwd_small")


def partials_sum(partials, nparts, stride, C, out):
    _chk(partials, out)
    _req(partials.dtype == torch.float64 and out.numel() == C, "partials_sum: argument check failed: partials.dtype == torch.float64 and out.numel() == C")
    check(lib().goalnet_partials_sum(partials.data_ptr(), nparts, stride, C, out.data_ptr(), _s()), "partials_sum")


def partials_sum2(pa, Ca, oa, pb, Cb, ob):
    """partials_sum of two [rows][C] arrays in one launch"""
    _chk(pa, oa, pb, ob)
    _req(oa.numel() == Ca and ob.numel() == Cb, "partials_sum2: out sizes")
    check(lib().goalnet_partials_sum2(pa.data_ptr(), _rows(pa, Ca), Ca, oa.data_ptr(), pb.data_ptr(), _rows(pb, Cb), Cb, ob.data_ptr(), _s()),
          "partials_sum2")


def partials_sum_f64(partials, C, out):
    """out[c] (double) = sum over the rows of partials[rows][C]"""
    _chk(partials, out)
    _req(partials.dtype == torch.float64 and out.dtype == torch.float64 and out.numel() == C, "partials_sum_f64: argument check failed: partials.dtype == torch.float64 and out.dtype == torch.float64 and out.numel() == C")
    check(lib().goalnet_partials_sum_f64(partials.data_ptr(), _rows(partials, C), C, C, out.data_ptr(), _s()), "partials_sum_f64")


N_TILE_CTR = 1024
# The fused split-K reduction (the last block of a tile sums its slabs) is built and tested, and OFF by default: one block summing
# 13 slabs x 64 KB is a chain of ~50 dependent memory round trips (measured on the 10-frame step: +450 us per step against the
# separate reduce launch, which spreads the same bytes over every CU and costs 1.6 us of launch gap inside a graph)
FUSED_SPLITK = os.environ.get("GOALNET_FUSED_SPLITK", "0") == "1"
_TILE_CTR = {}
_WGRAD_CODES = {}


def _tile_counters(key, device):
    """zeroed int32 ticket counters lent to the fused split-K reduction of one call site (include/goalnet_hip.h: zero on entry,
    zero again on exit). One tensor per (call site, shape): calls that may overlap on two streams — a weight gradient on the
    side stream under a data gradient on the main one — never share counters."""
    k = (key, device.index)
    t = _TILE_CTR.get(k)
    if t is None:
        t = _TILE_CTR[k] = torch.zeros(N_TILE_CTR, dtype=torch.int32, device=device)
    return t


def conv3x3_fwd(x, scale, shift, w, bias, relu, y, N, H, W, Cin, Cout):
    _chk(x, scale, shift, w, bias, y)
    _req(x.numel() == N * H * W * Cin and y.numel() == N * H * W * Cout and w.numel() == Cout * 9 * Cin, "conv3x3_fwd: argument check failed: x.numel() == N * H * W * Cin and y.numel() == N * H * W * Cout and w.numel() == Cout * 9 * Cin")
    _req(scale is None or (scale.numel() == Cin and shift.numel() == Cin), "conv3x3_fwd: argument check failed: scale is None or (scale.numel() == Cin and shift.numel() == Cin)")
    _req(bias is None or bias.numel() == Cout, "conv3x3_fwd: argument check failed: bias is None or bias.numel() == Cout")
    nbytes = lib().goalnet_conv3x3_fwd_ws_bytes(N, H, W, Cin, Cout)       # > 0 only for small N (split-K slabs)
    ws = torch.empty(nbytes // 4, dtype=F32, device=x.device) if nbytes else None
    ctr = _tile_counters(("conv3x3_fwd", N, H, W, Cin, Cout, scale is None), x.device) if (nbytes and FUSED_SPLITK) else None
    check(lib().goalnet_conv3x3_fwd(x.data_ptr(), _p(scale), _p(shift), w.data_ptr(), _p(bias), int(relu), y.data_ptr(),
                                    N, H, W, Cin, Cout, _p(ws), nbytes, _p(ctr), N_TILE_CTR if ctr is not None else 0, _s()), "conv3x3_fwd")
    return y


def conv3x3_wgrad(x, scale, shift, dy, dw, N, H, W, Cin, Cout):
    _chk(x, scale, shift, dy, dw)
    _req(x.numel() == N * H * W * Cin and dy.numel() == N * H * W * Cout and dw.numel() == Cout * 9 * Cin, "conv3x3_wgrad: argument check failed: x.numel() == N * H * W * Cin and dy.numel() == N * H * W * Cout and dw.numel() == Cout * 9 * Cin")
    nbytes = lib().goalnet_conv3x3_wgrad_ws_bytes(N, H, W, Cin, Cout)
    ws = torch.empty(nbytes // 4, dtype=F32, device=x.device)
    # the border-code table depends on (N, H, W) only: built once per shape and device, not once per call
    ck = (N, H, W, x.device.index)
    codes = _WGRAD_CODES.get(ck)
    if codes is None:
        codes = torch.empty(lib().goalnet_conv3x3_wgrad_codes_bytes(N, H, W) // 4, dtype=torch.int32, device=x.device)
        check(lib().goalnet_conv3x3_wgrad_codes(codes.data_ptr(), N, H, W, _s()), "conv3x3_wgrad_codes")
        if len(_WGRAD_CODES) < 64 and N * H * W <= (1 << 22):
            _WGRAD_CODES[ck] = codes
    ctr = _tile_counters(("conv3x3_wgrad", N, H, W, Cin, Cout), x.device) if FUSED_SPLITK else None
    check(lib().goalnet_conv3x3_wgrad(x.data_ptr(), _p(scale), _p(shift), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes,
                                      codes.data_ptr(), _p(ctr), N_TILE_CTR if ctr is not None else 0, N, H, W, Cin, Cout, _s()), "conv3x3_wgrad")
    return dw


BF16 = torch.bfloat16
F16 = torch.float16
H16 = (BF16, F16)          # the 16-bit storage formats of the reduced-precision engine; the C ABI's `f16` flag = (dtype == float16)


def _f16(*ts):
    """the `f16` flag of a call: 1 when its 16-bit tensors are torch.float16, 0 for bfloat16 (they must agree)"""
    kinds = {t.dtype for t in ts if t is not None and t.dtype in H16}
    _req(len(kinds) <= 1, "_f16: 16-bit operands of one call must share a format")
    return int(F16 in kinds)


def cast_bf16(x, y):
    _chk(x, y)
    _req(x.dtype == F32 and y.dtype in H16 and x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel(), "cast_bf16: argument check failed: x.dtype == F32 and y.dtype in H16 and x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()")
    check(lib().goalnet_cast_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _f16(y), _s()), "cast_bf16")
    return y


def cast_f32(x, y):
    _chk(x, y)
    _req(x.dtype in H16 and y.dtype == F32 and x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel(), "cast_f32: argument check failed: x.dtype in H16 and y.dtype == F32 and x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()")
    check(lib().goalnet_cast_f32(x.data_ptr(), y.data_ptr(), x.numel(), _f16(x), _s()), "cast_f32")
    return y


def bn_apply_bf16(x, scale, shift, y, C):
    _chk(x, scale, shift, y)
    _req(x.dtype in (F32,) + H16 and y.dtype in H16 and x.numel() == y.numel() and scale.numel() == C, "bn_apply_bf16: argument check failed: x.dtype in (F32,) + H16 and y.dtype in H16 and x.numel() == y.numel() and scale.numel() == C")
    if x.dtype in H16:
        check(lib().goalnet_bn_apply_bf16_p16(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), x.numel(), C, _f16(x, y), _s()), "bn_apply_bf16_p16")
        return y
    check(lib().goalnet_bn_apply_bf16(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), x.numel(), C, _f16(y), _s()), "bn_apply_bf16")
    return y


def conv3x3_fwd_bf16(x, w, bias, relu, y, N, H, W, Cin, Cout):
    _chk(x, w, bias, y)
    _req(x.dtype in H16 and w.dtype == x.dtype and y.dtype == F32, "conv3x3_fwd_bf16: argument check failed: x.dtype in H16 and w.dtype == x.dtype and y.dtype == F32")
    _req(x.numel() == N * H * W * Cin and y.numel() == N * H * W * Cout and w.numel() == Cout * 9 * Cin, "conv3x3_fwd_bf16: argument check failed: x.numel() == N * H * W * Cin and y.numel() == N * H * W * Cout and w.numel() == Cout * 9 * Cin")
    check(lib().goalnet_conv3x3_fwd_bf16(x.data_ptr(), w.data_ptr(), _p(bias), int(relu), y.data_ptr(), N, H, W, Cin, Cout, _f16(x, w), _s()),
          "conv3x3_fwd_bf16")
    return y


def padded_bf16_alloc(N, H, W, C, device, dtype=torch.bfloat16):
    """Zeroed 16-bit buffer for a padded [N][H+2][W+2][C] tensor with its guard bands; returns (buffer, view at padded pixel 0)."""
    import ctypes as _ct
    tot, off = _ct.c_int64(), _ct.c_int64()
    check(lib().goalnet_bf16_padded_layout(N, H, W, C, _ct.byref(tot), _ct.byref(off)), "bf16_padded_layout")
    buf = torch.zeros(tot.value, dtype=dtype, device=device)
    return buf, buf[off.value:]


def to_bf16_padded(x, scale, shift, ypad, N, H, W, C):
    _chk(x, scale, shift, ypad)
    _req(x.dtype in (F32,) + H16 and ypad.dtype in H16 and x.numel() == N * H * W * C and ypad.numel() >= N * (H + 2) * (W + 2) * C, "to_bf16_padded: argument check failed: x.dtype in (F32,) + H16 and ypad.dtype in H16 and x.numel() == N * H * W * C and ypad.numel() >= N * (H + 2) * (W + 2) * C")
    if x.dtype in H16:
        check(lib().goalnet_to_bf16_padded_p16(x.data_ptr(), _p(scale), _p(shift), ypad.data_ptr(), N, H, W, C, _f16(x, ypad), _s()), "to_bf16_padded_p16")
        return ypad
    check(lib().goalnet_to_bf16_padded(x.data_ptr(), _p(scale), _p(shift), ypad.data_ptr(), N, H, W, C, _f16(ypad), _s()), "to_bf16_padded")
    return ypad


def conv3x3_fwd_bf16p(xpad, w, bias, relu, y, N, H, W, Cin, Cout):
    _chk(xpad, w, bias, y)
    _req(xpad.dtype in H16 and w.dtype == xpad.dtype and y.dtype == F32, "conv3x3_fwd_bf16p: argument check failed: xpad.dtype in H16 and w.dtype == xpad.dtype and y.dtype == F32")
    _req(y.numel() == N * H * W * Cout and w.numel() == Cout * 9 * Cin, "conv3x3_fwd_bf16p: argument check failed: y.numel() == N * H * W * Cout and w.numel() == Cout * 9 * Cin")
    nbytes = lib().goalnet_conv3x3_fwd_bf16p_ws_bytes(N, H, W, Cin, Cout)
    ws = torch.empty(nbytes // 4, dtype=F32, device=y.device) if nbytes else None
    check(lib().goalnet_conv3x3_fwd_bf16p(xpad.data_ptr(), w.data_ptr(), _p(bias), int(relu), y.data_ptr(), N, H, W, Cin, Cout,
                                          _p(ws), nbytes, _f16(xpad, w), _s()), "conv3x3_fwd_bf16p")
    return y


def conv3x3_fwd_bf16p_o16_ok(N, H, W, Cin, Cout) -> bool:
    return bool(lib().goalnet_conv3x3_fwd_bf16p_o16_ok(N, H, W, Cin, Cout))


def conv3x3_fwd_bf16p_o16(xpad, w, bias, relu, y, N, H, W, Cin, Cout):
    """bf16 result (bias None, relu False: the data gradient); only for dims conv3x3_fwd_bf16p_o16_ok accepts"""
    _chk(xpad, w, bias, y)
    _req(xpad.dtype in H16 and w.dtype == xpad.dtype and y.dtype == xpad.dtype, "conv3x3_fwd_bf16p_o16: argument check failed: xpad.dtype in H16 and w.dtype == xpad.dtype and y.dtype == xpad.dtype")
    _req(y.numel() == N * H * W * Cout and w.numel() == Cout * 9 * Cin, "conv3x3_fwd_bf16p_o16: argument check failed: y.numel() == N * H * W * Cout and w.numel() == Cout * 9 * Cin")
    check(lib().goalnet_conv3x3_fwd_bf16p_o16(xpad.data_ptr(), w.data_ptr(), _p(bias), int(relu), y.data_ptr(), N, H, W, Cin, Cout,
                                              _f16(xpad, w, y), _s()),
          "conv3x3_fwd_bf16p_o16")
    return y


def conv3x3_wgrad_bf16(xpad, dypad, dw, N, H, W, Cin, Cout):
    _chk(xpad, dypad, dw)
    _req(xpad.dtype in H16 and dypad.dtype == xpad.dtype and dw.dtype == F32 and dw.numel() == Cout * 9 * Cin, "conv3x3_wgrad_bf16: argument check failed: xpad.dtype in H16 and dypad.dtype == xpad.dtype and dw.dtype == F32 and dw.numel() == Cout * 9 * Cin")
    nbytes = lib().goalnet_conv3x3_wgrad_bf16_ws_bytes(N, H, W, Cin, Cout)
    ws = torch.empty(nbytes // 4, dtype=F32, device=dw.device)
    check(lib().goalnet_conv3x3_wgrad_bf16(xpad.data_ptr(), dypad.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes,
                                           N, H, W, Cin, Cout, _f16(xpad, dypad), _s()), "conv3x3_wgrad_bf16")
    return dw


# ---- precision = "bf16x6" / "fp16x3" (csrc/split3.hip): fp32 operands as bf16 triples (parts = 3, six partial products per product) or
# as fp16 pairs of the power-of-two-scaled value (parts = 2, three partial products; `amax` = the tensor's magnitude word, `oscale` =
# the two operands' combined scale and its inverse for the GEMM epilogue)
def _part_dtype(parts):
    _req(parts in (2, 3), "parts must be 3 (bf16 triples) or 2 (scaled fp16 pairs)")
    return torch.bfloat16 if parts == 3 else torch.float16


def absmax(x, amax, rows, C, scale=None, shift=None, bnC=0):
    """atomic max of the bit pattern of |x * scale + shift| over a (rows, C) fp32 matrix into the int32 word `amax` (zeroed by the caller)"""
    _chk(x, amax, scale, shift)
    _req(x.dtype == F32 and amax.dtype == torch.int32 and amax.numel() == 1 and x.numel() >= rows * C, "absmax: sizes / dtypes")
    ldx = _ld(x) if x.dim() == 2 else C
    check(lib().goalnet_absmax(x.data_ptr(), ldx, _p(scale), _p(shift), bnC, rows, C, amax.data_ptr(), _s()), "absmax")
    return amax


def split_scales(amax_a, amax_b, oscale=None):
    """oscale = the exponent (int32 word on the device) the GEMM epilogue adds to undo the two operands' power-of-two scales"""
    oscale = torch.empty(1, dtype=torch.int32, device=amax_a.device) if oscale is None else oscale
    _chk(amax_a, amax_b, oscale)
    check(lib().goalnet_split_scales(amax_a.data_ptr(), amax_b.data_ptr(), oscale.data_ptr(), _s()), "split_scales")
    return oscale


def split_padded(parts, x, scale, shift, ypads, N, H, W, C, amax=None):
    """x fp32 (N,H,W,C) [* scale + shift per channel] -> interior of the zero-padded (N,H+2,W+2,parts C) 16-bit view `ypads`"""
    _chk(x, scale, shift, ypads, amax)
    _req(x.dtype == F32 and ypads.dtype == _part_dtype(parts) and x.numel() == N * H * W * C and ypads.numel() >= N * (H + 2) * (W + 2) * parts * C
         and (parts == 3 or amax is not None), "split_padded: x fp32 (N,H,W,C), ypads 16-bit (N,H+2,W+2,parts C); parts = 2 needs amax")
    check(lib().goalnet_split_padded(parts, x.data_ptr(), _p(scale), _p(shift), _p(amax), ypads.data_ptr(), N, H, W, C, _s()), "split_padded")
    return ypads


def split_rows(parts, x, ys, rows, C, scale=None, shift=None, bnC=0, amax=None):
    """x fp32 (rows, C) (row stride = x's leading dim when 2-D) [* scale + shift with channel = column % bnC] -> ys 16-bit [rows][parts C]"""
    _chk(x, ys, scale, shift, amax)
    _req(x.dtype == F32 and ys.dtype == _part_dtype(parts) and ys.numel() == parts * rows * C and (parts == 3 or amax is not None),
         "split_rows: sizes / dtypes; parts = 2 needs amax")
    ldx = _ld(x) if x.dim() == 2 else C
    _req(x.dim() == 2 and tuple(x.shape) == (rows, C) or x.numel() == rows * C, "split_rows: x must hold rows x C values")
    check(lib().goalnet_split_rows(parts, x.data_ptr(), ldx, _p(scale), _p(shift), bnC, _p(amax), ys.data_ptr(), rows, C, _s()), "split_rows")
    return ys


def linear_split_ok(parts, M, K, J) -> bool:
    return bool(lib().goalnet_linear_split_ok(parts, M, K, J))


def conv3x3_fwd_split(parts, xpads, ws, bias, relu, y, N, H, W, Cin, Cout, oscale=None):
    _chk(xpads, ws, bias, y, oscale)
    dt = _part_dtype(parts)
    _req(xpads.dtype == dt and ws.dtype == dt and y.dtype == F32 and ws.numel() == Cout * 9 * parts * Cin and y.numel() == N * H * W * Cout
         and xpads.numel() >= N * (H + 2) * (W + 2) * parts * Cin and (parts == 3 or oscale is not None), "conv3x3_fwd_split: sizes / dtypes")
    check(lib().goalnet_conv3x3_fwd_split(parts, xpads.data_ptr(), ws.data_ptr(), _p(bias), int(relu), y.data_ptr(), N, H, W, Cin, Cout, _p(oscale), _s()),
          "conv3x3_fwd_split")
    return y


def conv3x3_wgrad_split(parts, xpads, dypads, dw, N, H, W, Cin, Cout, oscale=None):
    _chk(xpads, dypads, dw, oscale)
    dt = _part_dtype(parts)
    _req(xpads.dtype == dt and dypads.dtype == dt and dw.dtype == F32 and dw.numel() == Cout * 9 * Cin and (parts == 3 or oscale is not None),
         "conv3x3_wgrad_split: sizes / dtypes")
    nbytes = lib().goalnet_conv3x3_wgrad_split_ws_bytes(parts, N, H, W, Cin, Cout)
    ws = torch.empty(nbytes // 4, dtype=F32, device=dw.device)
    check(lib().goalnet_conv3x3_wgrad_split(parts, xpads.data_ptr(), dypads.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes, N, H, W, Cin, Cout,
                                            _p(oscale), _s()), "conv3x3_wgrad_split")
    return dw


def linear_fwd_split(parts, xs, wsp, bias, y, M, K, J, *, relu=False, dropmask=None, mult_out=None, oscale=None):
    """y (M,J view, fp32) from the split operands xs [M][parts K], wsp [J][parts K]"""
    _chk(xs, wsp, bias, y, dropmask, mult_out, oscale)
    dt = _part_dtype(parts)
    _req(xs.dtype == dt and wsp.dtype == dt and xs.numel() == parts * M * K and wsp.numel() == parts * J * K and tuple(y.shape) == (M, J)
         and (parts == 3 or oscale is not None), "linear_fwd_split: sizes / dtypes")
    nbytes = lib().goalnet_linear_fwd_split_ws_bytes(parts, M, K, J)
    _req(nbytes > 0, "linear_fwd_split: dims not served (linear_split_ok)")
    ws = torch.empty(nbytes // 4, dtype=F32, device=y.device)
    check(lib().goalnet_linear_fwd_split(parts, xs.data_ptr(), wsp.data_ptr(), _p(bias), int(relu), _p(dropmask), 0 if dropmask is None else _ld(dropmask),
                                         y.data_ptr(), _ld(y), _p(mult_out), 0 if mult_out is None else _ld(mult_out), M, K, J, ws.data_ptr(), nbytes,
                                         _p(oscale), _s()), "linear_fwd_split")
    return y


def linear_bwd_dx_split(parts, dys, wsp, dx, M, K, J, oscale=None):
    _chk(dys, wsp, dx, oscale)
    dt = _part_dtype(parts)
    _req(dys.dtype == dt and wsp.dtype == dt and dx.dtype == F32 and dys.numel() == parts * M * J and wsp.numel() == parts * J * K
         and tuple(dx.shape) == (M, K) and (parts == 3 or oscale is not None), "linear_bwd_dx_split: sizes / dtypes")
    check(lib().goalnet_linear_bwd_dx_split(parts, dys.data_ptr(), wsp.data_ptr(), dx.data_ptr(), _ld(dx), M, K, J, _p(oscale), _s()), "linear_bwd_dx_split")
    return dx


def linear_bwd_dw_split(parts, dys, xs, dw, M, K, J, oscale=None):
    _chk(dys, xs, dw, oscale)
    dt = _part_dtype(parts)
    _req(dys.dtype == dt and xs.dtype == dt and dw.dtype == F32 and dys.numel() == parts * M * J and xs.numel() == parts * M * K
         and dw.numel() == J * K and (parts == 3 or oscale is not None), "linear_bwd_dw_split: sizes / dtypes")
    check(lib().goalnet_linear_bwd_dw_split(parts, dys.data_ptr(), xs.data_ptr(), dw.data_ptr(), M, K, J, _p(oscale), _s()), "linear_bwd_dw_split")
    return dw


def linear_bwd_dx_bf16(dy, w, dx, mult=None):
    _chk(dy, w, dx, mult)
    _req(dy.dtype in H16 and w.dtype == dy.dtype and dx.dtype == F32, "linear_bwd_dx_bf16: argument check failed: dy.dtype in H16 and w.dtype == dy.dtype and dx.dtype == F32")
    M, J = dy.shape
    K = dx.shape[1]
    _req(w.numel() == J * K and dx.shape[0] == M, "linear_bwd_dx_bf16: argument check failed: w.numel() == J * K and dx.shape[0] == M")
    check(lib().goalnet_linear_bwd_dx_bf16(dy.data_ptr(), _ld(dy), w.data_ptr(), _p(mult), 0 if mult is None else _ld(mult),
                                           dx.data_ptr(), _ld(dx), M, K, J, _f16(dy, w), _s()), "linear_bwd_dx_bf16")
    return dx


def linear_bwd_dx_bf16_o16_ok(M, K, J) -> bool:
    return bool(lib().goalnet_linear_bwd_dx_bf16_o16_ok(M, K, J))


def linear_bwd_dx_bf16_o16(dy, w, dx):
    """bf16 result; only for dims linear_bwd_dx_bf16_o16_ok accepts"""
    _chk(dy, w, dx)
    _req(dy.dtype in H16 and w.dtype == dy.dtype and dx.dtype == dy.dtype, "linear_bwd_dx_bf16_o16: argument check failed: dy.dtype in H16 and w.dtype == dy.dtype and dx.dtype == dy.dtype")
    M, J = dy.shape
    K = dx.shape[1]
    _req(w.numel() == J * K and dx.shape[0] == M, "linear_bwd_dx_bf16_o16: argument check failed: w.numel() == J * K and dx.shape[0] == M")
    check(lib().goalnet_linear_bwd_dx_bf16_o16(dy.data_ptr(), _ld(dy), w.data_ptr(), dx.data_ptr(), _ld(dx), M, K, J, _f16(dy, w, dx), _s()),
          "linear_bwd_dx_bf16_o16")
    return dx


def linear_bwd_dw_bf16(dy, x, dw):
    _chk(dy, x, dw)
    _req(dy.dtype in H16 and x.dtype == dy.dtype and dw.dtype == F32, "linear_bwd_dw_bf16: argument check failed: dy.dtype in H16 and x.dtype == dy.dtype and dw.dtype == F32")
    M, J = dy.shape
    K = x.shape[1]
    _req(dw.numel() == J * K and x.shape[0] == M, "linear_bwd_dw_bf16: argument check failed: dw.numel() == J * K and x.shape[0] == M")
    check(lib().goalnet_linear_bwd_dw_bf16(dy.data_ptr(), _ld(dy), x.data_ptr(), _ld(x), dw.data_ptr(), M, K, J, _f16(dy, x), _s()), "linear_bwd_dw_bf16")
    return dw


def linear_fwd_bf16(x, w, bias, y, *, relu=False, dropmask=None, mult_out=None):
    _chk(x, w, bias, y, dropmask, mult_out)
    _req(x.dtype in H16 and w.dtype == x.dtype, "linear_fwd_bf16: argument check failed: x.dtype in H16 and w.dtype == x.dtype")
    M, K = x.shape
    J = y.shape[1]
    _req(w.numel() == J * K and y.shape[0] == M, "linear_fwd_bf16: argument check failed: w.numel() == J * K and y.shape[0] == M")
    nbytes = lib().goalnet_linear_fwd_bf16_ws_bytes(M, K, J)
    ws = torch.empty(max(nbytes // 4, 1), dtype=F32, device=x.device) if nbytes else None
    check(lib().goalnet_linear_fwd_bf16(x.data_ptr(), _ld(x), w.data_ptr(), _p(bias), int(relu), _p(dropmask),
                                        0 if dropmask is None else _ld(dropmask), y.data_ptr(), _ld(y), _p(mult_out),
                                        0 if mult_out is None else _ld(mult_out), M, K, J, _p(ws), nbytes, _f16(x, w), _s()), "linear_fwd_bf16")
    return y


def linear_fwd(x, w, bias, y, *, relu=False, scale=None, shift=None, bnC=0, dropmask=None, mult_out=None, K=None):
    """y (M,J view) = act(x (M,K view) @ w(J,K)^T + bias) * dropmask. x/y/dropmask/mult_out may be column
    slices of wider buffers (row stride = leading dim)."""
    _chk(x, w, bias, y, scale, shift, dropmask, mult_out)
    M = x.shape[0]
    K = x.shape[1] if K is None else K
    J = y.shape[1]
    _req(w.numel() == J * K and y.shape[0] == M, "linear_fwd: argument check failed: w.numel() == J * K and y.shape[0] == M")
    nbytes = lib().goalnet_linear_fwd_ws_bytes(M, K, J)
    ws = torch.empty(max(nbytes // 4, 1), dtype=F32, device=x.device) if nbytes else None
    check(lib().goalnet_linear_fwd(x.data_ptr(), _ld(x), _p(scale), _p(shift), bnC, w.data_ptr(), _p(bias), int(relu),
                                   _p(dropmask), 0 if dropmask is None else _ld(dropmask), y.data_ptr(), _ld(y),
                                   _p(mult_out), 0 if mult_out is None else _ld(mult_out), M, K, J, _p(ws), nbytes, _s()),
          "linear_fwd")
    return y


def linear_bwd_dx(dy, w, dx, mult=None):
    _chk(dy, w, dx, mult)
    M, J = dy.shape
    K = dx.shape[1]
    _req(w.numel() == J * K and dx.shape[0] == M, "linear_bwd_dx: argument check failed: w.numel() == J * K and dx.shape[0] == M")
    check(lib().goalnet_linear_bwd_dx(dy.data_ptr(), _ld(dy), w.data_ptr(), _p(mult), 0 if mult is None else _ld(mult),
                                      dx.data_ptr(), _ld(dx), M, K, J, _s()), "linear_bwd_dx")
    return dx


def linear_bwd_dw(dy, x, dw, *, scale=None, shift=None, bnC=0, db=None):
    """db (optional): bias gradient = column sums of dy, produced by the same call"""
    _chk(dy, x, dw, scale, shift, db)
    M, J = dy.shape
    K = x.shape[1]
    _req(dw.numel() == J * K and x.shape[0] == M and (db is None or db.numel() == J), "linear_bwd_dw: argument check failed: dw.numel() == J * K and x.shape[0] == M and (db is None or db.numel() == J)")
    check(lib().goalnet_linear_bwd_dw(dy.data_ptr(), _ld(dy), x.data_ptr(), _ld(x), _p(scale), _p(shift), bnC, dw.data_ptr(), _p(db),
                                      M, K, J, _s()), "linear_bwd_dw")
    return dw


def colsum(x, out):
    _chk(x, out)
    M, J = x.shape
    _req(out.numel() == J, "colsum: argument check failed: out.numel() == J")
    check(lib().goalnet_colsum(x.data_ptr(), _ld(x), M, J, out.data_ptr(), _s()), "colsum")
    return out


def mul(x, m, y):
    _chk(x, m, y)
    M, J = x.shape
    check(lib().goalnet_mul(x.data_ptr(), _ld(x), m.data_ptr(), _ld(m), y.data_ptr(), _ld(y), M, J, _s()), "mul")
    return y


def conv1d_fwd(x, w, b, y, relu, N, Cin, L, Cout, stride=2, pad=1):
    _chk(x, w, b, y)
    Lo = (L + 2 * pad - 3) // stride + 1
    _req(x.numel() == N * Cin * L and y.numel() == N * Cout * Lo and w.numel() == Cout * Cin * 3, "conv1d_fwd: argument check failed: x.numel() == N * Cin * L and y.numel() == N * Cout * Lo and w.numel() == Cout * Cin * 3")
    check(lib().goalnet_conv1d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), int(relu), y.data_ptr(), N, Cin, L, Cout, stride, pad,
                                   _s()), "conv1d_fwd")
    return y


def conv1d_bwd(x, dz, w, dx, dw, db, N, Cin, L, Cout, stride=2, pad=1):
    _chk(x, dz, w, dx, dw, db)
    Lo = (L + 2 * pad - 3) // stride + 1
    _req(x.numel() == N * Cin * L and dz.numel() == N * Cout * Lo, "conv1d_bwd: argument check failed: x.numel() == N * Cin * L and dz.numel() == N * Cout * Lo")
    nbytes = lib().goalnet_conv1d_bwd_ws_bytes(N, Cin, Cout)
    ws = torch.empty(nbytes // 8, dtype=torch.float64, device=x.device) if nbytes else None
    check(lib().goalnet_conv1d_bwd(x.data_ptr(), dz.data_ptr(), w.data_ptr(), _p(dx), dw.data_ptr(), db.data_ptr(), N, Cin, L, Cout,
                                   stride, pad, _p(ws), nbytes, _s()), "conv1d_bwd")


def conv1d_bwd_small(x, dz, y, w, dx, dw, db, N, Cin, L, Cout, stride=2, pad=1):
    """conv1d_bwd for N < 64 frames in one launch; y (the layer's ReLU output) gates dz when given"""
    _chk(x, dz, y, w, dx, dw, db)
    Lo = (L + 2 * pad - 3) // stride + 1
    _req(x.numel() == N * Cin * L and dz.numel() == N * Cout * Lo and (y is None or y.numel() == dz.numel()), "conv1d_bwd_small: x / dz / y shapes")
    _req(dw.numel() == Cout * Cin * 3 and db.numel() == Cout and (dx is None or dx.numel() == x.numel()), "conv1d_bwd_small: dw / db / dx shapes")
    check(lib().goalnet_conv1d_bwd_small(x.data_ptr(), dz.data_ptr(), _p(y), w.data_ptr(), _p(dx), dw.data_ptr(), db.data_ptr(),
                                         N, Cin, L, Cout, stride, pad, _s()), "conv1d_bwd_small")


def relu_bwd(dy, y, dz):
    _chk(dy, y, dz)
    _req(dy.is_contiguous() and y.is_contiguous() and dz.is_contiguous() and dy.numel() == y.numel() == dz.numel(), "relu_bwd: argument check failed: dy.is_contiguous() and y.is_contiguous() and dz.is_contiguous() and dy.numel() == y.numel() == dz.numel()")
    check(lib().goalnet_relu_bwd(dy.data_ptr(), y.data_ptr(), dz.data_ptr(), dy.numel(), _s()), "relu_bwd")
    return dz


MLP_WIDTHS = (512, 512, 256, 128)


def _ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[_p(t) for t in ts])


def mlp_fwd(cat, ws, bs, masks, hs, mults, logit, out, labels=None, loss=None, dout=None):
    """fusion.0 .. fusion.12 + Sigmoid + 4y+1 on <= 16 rows in one launch. cat (n, K0) may be a view with a row stride;
    ws / bs: 5 weights / biases; masks: 4 dropout multipliers or None; hs: 4 outputs; mults: 4 saved multipliers or None"""
    _chk(cat, logit, out, labels, loss, dout, *ws, *bs, *masks, *hs, *mults)
    n, K0 = cat.shape
    _req(labels is None or (labels.numel() == n and labels.dtype == F32 and labels.is_contiguous() and loss is not None and dout is not None
                            and dout.numel() == n), "mlp_fwd: labels (n) need loss (1) and dout (n)")
    _req(len(ws) == 5 and len(bs) == 5 and len(masks) == 4 and len(hs) == 4 and len(mults) == 4, "mlp_fwd: 5 layers, 4 hidden outputs")
    for l, wd in enumerate(MLP_WIDTHS):
        kin = K0 if l == 0 else MLP_WIDTHS[l - 1]
        _req(ws[l].numel() == wd * kin and bs[l].numel() == wd and hs[l].shape == (n, wd) and hs[l].is_contiguous(), f"mlp_fwd: layer {l} shapes")
        _req(masks[l] is None or masks[l].shape == (n, wd), f"mlp_fwd: mask {l} shape")
        _req(mults[l] is None or (mults[l].shape == (n, wd) and mults[l].is_contiguous()), f"mlp_fwd: mult {l} shape")
    _req(ws[4].numel() == 128 and bs[4].numel() == 1 and logit.numel() == n and out.numel() == n, "mlp_fwd: head shapes")
    ldm = (ctypes.c_int64 * 4)(*[0 if m is None else _ld(m) for m in masks])
    sync = _tile_counters(("mlp_fwd", n, K0), cat.device)
    check(lib().goalnet_mlp_fwd(cat.data_ptr(), _ld(cat), K0, _ptrs(ws), _ptrs(bs), _ptrs(masks), ldm, _ptrs(hs), _ptrs(mults),
                                logit.data_ptr(), out.data_ptr(), _p(labels), _p(loss), _p(dout), n, sync.data_ptr(), _s()), "mlp_fwd")


def mlp_bwd(dout, out, xs, ms, ws, dws, dbs, dcat, db5, voff):
    """backward of mlp_fwd in one launch. xs: cat (view), h1..h4; ms: mcat (view), mult1..4; dws / dbs: 5 gradient slots;
    dcat (n, K0) contiguous; db5 (K0 - voff) or None."""
    _chk(dout, out, dcat, db5, *xs, *ms, *ws, *dws, *dbs)
    n, K0 = xs[0].shape
    _req(len(xs) == 5 and len(ms) == 5 and len(ws) == 5 and len(dws) == 5 and len(dbs) == 5, "mlp_bwd: 5 layers")
    _req(dout.numel() == n and out.numel() == n and dout.is_contiguous() and out.is_contiguous(), "mlp_bwd: dout / out (n)")
    _req(dcat.shape == (n, K0) and dcat.is_contiguous() and (db5 is None or db5.numel() == K0 - voff), "mlp_bwd: dcat (n, K0), db5 (K0 - voff)")
    for l in range(1, 5):
        _req(xs[l].shape == (n, MLP_WIDTHS[l - 1]) and xs[l].is_contiguous() and (ms[l] is None or (ms[l].shape == xs[l].shape and ms[l].is_contiguous())),
             f"mlp_bwd: layer {l} input / multiplier shapes")
    for l in range(5):
        _req(dws[l].numel() == ws[l].numel() and dws[l].is_contiguous() and dbs[l].is_contiguous(), f"mlp_bwd: layer {l} gradient slots")
    nbytes = lib().goalnet_mlp_bwd_ws_bytes(n)
    wsb = torch.empty(nbytes // 4, dtype=F32, device=dout.device)
    sync = _tile_counters(("mlp_bwd", n, K0), dout.device)
    check(lib().goalnet_mlp_bwd(dout.data_ptr(), out.data_ptr(), _ptrs(xs), _ld(xs[0]), _ptrs(ms), 0 if ms[0] is None else _ld(ms[0]),
                                _ptrs(ws), _ptrs(dws), _ptrs(dbs), dcat.data_ptr(), _ld(dcat), _p(db5), voff, n, K0, wsb.data_ptr(), nbytes,
                                sync.data_ptr(), _s()), "mlp_bwd")


def mlp_sync_error(device, n, K0) -> bool:
    """True when a grid barrier of the fused MLP kernels timed out on this device (one host read-back; tests)"""
    bad = False
    for name in ("mlp_fwd", "mlp_bwd"):
        t = _TILE_CTR.get(((name, n, K0), device.index))
        bad = bad or (t is not None and bool(t[2].item() != 0))
    return bad


def head_fwd(h, w, b, logit, out):
    _chk(h, w, b, logit, out)
    N, K = h.shape
    check(lib().goalnet_head_fwd(h.data_ptr(), _ld(h), w.data_ptr(), b.data_ptr(), _p(logit), out.data_ptr(), N, K, _s()), "head_fwd")


def head_bwd(dout, out, h, w, mult, dh, dw, db):
    _chk(dout, out, h, w, mult, dh, dw, db)
    N, K = h.shape
    _req(dout.numel() == N and out.numel() == N and dout.is_contiguous() and out.is_contiguous(), "head_bwd: argument check failed: dout.numel() == N and out.numel() == N and dout.is_contiguous() and out.is_contiguous()")
    check(lib().goalnet_head_bwd(dout.data_ptr(), out.data_ptr(), h.data_ptr(), _ld(h), w.data_ptr(), _p(mult),
                                 0 if mult is None else _ld(mult), dh.data_ptr(), _ld(dh), dw.data_ptr(), db.data_ptr(), N, K, _s()),
          "head_bwd")


def cls_head_fwd(h, w, b, logits, scores):
    """classifier-head extension: scores (N, C) = 4 softmax(h w^T + b) + 1"""
    _chk(h, w, b, logits, scores)
    N, K = h.shape
    C = scores.shape[1]
    _req(w.numel() == C * K and b.numel() == C and scores.is_contiguous() and (logits is None or logits.is_contiguous()), "cls_head_fwd: argument check failed: w.numel() == C * K and b.numel() == C and scores.is_contiguous() and (logits is None or logits.is_contiguous())")
    check(lib().goalnet_cls_head_fwd(h.data_ptr(), _ld(h), w.data_ptr(), b.data_ptr(), _p(logits), scores.data_ptr(), N, K, C, _s()), "cls_head_fwd")


def cross_entropy(scores, labels, loss, dscores):
    """nn.CrossEntropyLoss()(scores, (labels - 1).long()) and its gradient wrt scores"""
    _chk(scores, labels, loss, dscores)
    N, C = scores.shape
    _req(labels.numel() == N and scores.is_contiguous() and (dscores is None or dscores.is_contiguous()), "cross_entropy: argument check failed: labels.numel() == N and scores.is_contiguous() and (dscores is None or dscores.is_contiguous())")
    check(lib().goalnet_cross_entropy(scores.data_ptr(), labels.data_ptr(), _p(loss), _p(dscores), N, C, _s()), "cross_entropy")


def cls_head_bwd(dscores, scores, h, w, mult, dh, dw, db):
    _chk(dscores, scores, h, w, mult, dh, dw, db)
    N, K = h.shape
    C = scores.shape[1]
    _req(dscores.is_contiguous() and scores.is_contiguous() and dw.numel() == C * K and db.numel() == C, "cls_head_bwd: argument check failed: dscores.is_contiguous() and scores.is_contiguous() and dw.numel() == C * K and db.numel() == C")
    check(lib().goalnet_cls_head_bwd(dscores.data_ptr(), scores.data_ptr(), h.data_ptr(), _ld(h), w.data_ptr(), _p(mult),
                                     0 if mult is None else _ld(mult), dh.data_ptr(), _ld(dh), dw.data_ptr(), db.data_ptr(), N, K, C, _s()),
          "cls_head_bwd")


def argmax_plus1(scores, classes):
    _chk(scores, classes)
    N, C = scores.shape
    _req(classes.numel() == N and scores.is_contiguous(), "argmax_plus1: argument check failed: classes.numel() == N and scores.is_contiguous()")
    check(lib().goalnet_argmax_plus1(scores.data_ptr(), classes.data_ptr(), N, C, _s()), "argmax_plus1")
    return classes


def mse_bcast(pred, labels, loss, dpred):
    _chk(pred, labels, loss, dpred)
    N = pred.numel()
    _req(labels.numel() == N, "mse_bcast: argument check failed: labels.numel() == N")
    check(lib().goalnet_mse_bcast(pred.data_ptr(), labels.data_ptr(), N, _p(loss), _p(dpred), _s()), "mse_bcast")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    _chk(p, g, m, v)
    n = p.numel()
    _req(g.numel() == n and m.numel() == n and v.numel() == n, "adam_step: argument check failed: g.numel() == n and m.numel() == n and v.numel() == n")
    check(lib().goalnet_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, beta1, beta2, eps, step, grad_scale,
                                  _s()), "adam_step")


# ---- device-resident step state (graph-capturable variants; include/goalnet_hip.h "device-resident step state") ----
I64 = torch.int64


def _ctr(t):
    _req(t.is_cuda and t.dtype == I64 and t.numel() == 1, "_ctr: argument check failed: t.is_cuda and t.dtype == I64 and t.numel() == 1")
    return t.data_ptr()


def counter_add(counter, delta):
    check(lib().goalnet_counter_add(_ctr(counter), int(delta), _s()), "counter_add")


def dropout_masks_dev(dst, n, widths, seed, tid_base, tid_stride, step, p, row_offset=0):
    """dst: flat fp32 buffer of n*sum(widths); returns the per-layer (n, width) views. row_offset: the masks are rows
    [row_offset, row_offset + n) of the (row_offset + n, width) masks the same stream yields (data-parallel shards)."""
    _chk(dst)
    total = n * sum(widths)
    _req(dst.is_contiguous() and dst.dtype == F32 and dst.numel() == total, "dropout_masks_dev: argument check failed: dst.is_contiguous() and dst.dtype == F32 and dst.numel() == total")
    arr = (ctypes.c_int * len(widths))(*widths)
    check(lib().goalnet_dropout_masks_dev(dst.data_ptr(), n, arr, len(widths), seed, tid_base, tid_stride, _ctr(step), p, int(row_offset), _s()),
          "dropout_masks_dev")
    out, off = [], 0
    for wdt in widths:
        out.append(dst[off:off + n * wdt].view(n, wdt))
        off += n * wdt
    return out


def counters_add4(counters, d0, d1, d2, d3):
    _req(counters.is_cuda and counters.dtype == I64 and counters.numel() == 4 and counters.is_contiguous(), "counters_add4: argument check failed: counters.is_cuda and counters.dtype == I64 and counters.numel() == 4 and counters.is_contiguous()")
    check(lib().goalnet_counters_add4(counters.data_ptr(), int(d0), int(d1), int(d2), int(d3), _s()), "counters_add4")


def counters_add4_guarded(counters, d0, d1, d2, d3, bad_step):
    """counters_add4 of a precision="fp16" step: the step count stays where it is when this step's Adam was skipped"""
    _req(counters.is_cuda and counters.dtype == I64 and counters.numel() == 4 and counters.is_contiguous(), "counters_add4_guarded: argument check failed: counters.is_cuda and counters.dtype == I64 and counters.numel() == 4 and counters.is_contiguous()")
    check(lib().goalnet_counters_add4_guarded(counters.data_ptr(), int(d0), int(d1), int(d2), int(d3), _ctr(bad_step), _s()), "counters_add4_guarded")


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, step_bias=0):
    """Adam with the 1-based step count = *step + step_bias"""
    _chk(p, g, m, v)
    n = p.numel()
    _req(g.numel() == n and m.numel() == n and v.numel() == n, "adam_step_dev: argument check failed: g.numel() == n and m.numel() == n and v.numel() == n")
    check(lib().goalnet_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, beta1, beta2, eps, _ctr(step),
                                      int(step_bias), grad_scale, _s()), "adam_step_dev")


def adam_step_dev_blocks(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, step_bias=0, max_blocks=128):
    """adam_step_dev as a background pass on at most max_blocks blocks"""
    _chk(p, g, m, v)
    n = p.numel()
    _req(g.numel() == n and m.numel() == n and v.numel() == n, "adam_step_dev_blocks: p, g, m, v must have the same size")
    check(lib().goalnet_adam_step_dev_blocks(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, beta1, beta2, eps, _ctr(step),
                                             int(step_bias), grad_scale, int(max_blocks), _s()), "adam_step_dev_blocks")


def adam_step_dev_shadow(p, g, m, v, lr, beta1, beta2, eps, step, shadow, shadow_begin, grad_scale=1.0, step_bias=0):
    """adam_step_dev that also writes bf16(p_new) of arena[shadow_begin : shadow_begin + shadow.numel()] into `shadow`"""
    _chk(p, g, m, v, shadow)
    n = p.numel()
    _req(g.numel() == n and m.numel() == n and v.numel() == n and shadow.dtype in H16 and shadow.is_contiguous(), "adam_step_dev_shadow: argument check failed: g.numel() == n and m.numel() == n and v.numel() == n and shadow.dtype in H16 and shadow.is_contiguous()")
    check(lib().goalnet_adam_step_dev_shadow(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, beta1, beta2, eps, _ctr(step),
                                             int(step_bias), grad_scale, shadow.data_ptr(), int(shadow_begin), shadow.numel(), _f16(shadow), _s()),
          "adam_step_dev_shadow")


def adam_step_dev_guarded(p, g, m, v, lr, beta1, beta2, eps, step, bad_step, shadow=None, shadow_begin=0, grad_scale=1.0, step_bias=0):
    """precision="fp16": adam_step_dev[_shadow] that leaves everything untouched when grad_finite_check stamped this step"""
    _chk(p, g, m, v, shadow)
    n = p.numel()
    _req(g.numel() == n and m.numel() == n and v.numel() == n and (shadow is None or (shadow.dtype in H16 and shadow.is_contiguous())), "adam_step_dev_guarded: argument check failed: g.numel() == n and m.numel() == n and v.numel() == n and (shadow is None or (shadow.dtype in H16 and shadow.is_contiguous()))")
    check(lib().goalnet_adam_step_dev_guarded(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, beta1, beta2, eps, _ctr(step),
                                              int(step_bias), grad_scale, _p(shadow), int(shadow_begin), 0 if shadow is None else shadow.numel(),
                                              0 if shadow is None else _f16(shadow), _ctr(bad_step), _s()), "adam_step_dev_guarded")


def scale_(x, s):
    _chk(x)
    _req(x.dtype == F32 and x.is_contiguous(), "scale_: argument check failed: x.dtype == F32 and x.is_contiguous()")
    check(lib().goalnet_scale(x.data_ptr(), x.numel(), float(s), _s()), "scale")
    return x


def grad_finite_check(g, step, bad_step, skipped, step_bias=1):
    _chk(g)
    _req(g.dtype == F32 and g.is_contiguous(), "grad_finite_check: argument check failed: g.dtype == F32 and g.is_contiguous()")
    check(lib().goalnet_grad_finite_check(g.data_ptr(), g.numel(), _ctr(step), int(step_bias), _ctr(bad_step), _ctr(skipped), _s()), "grad_finite_check")


def rows_copy_batch(segments):
    """segments: up to 4 of (table, block, nrows, cursor, cursor_bias, gather); rows [cursor + bias, +nrows) of `table`
    are read into (gather) or written from (scatter) `block`. One launch."""
    arr = (_lib.RowCopy * len(segments))()
    for k, (table, block, nrows, cursor, bias, gather) in enumerate(segments):
        _chk(table, block)
        _req(table.is_contiguous() and block.is_contiguous() and table.dtype == block.dtype, "rows_copy_batch: argument check failed: table.is_contiguous() and block.is_contiguous() and table.dtype == block.dtype")
        row_bytes = table[0].numel() * table.element_size() if table.dim() > 1 else table.element_size()
        _req(block.numel() * block.element_size() == row_bytes * nrows, "rows_copy_batch: argument check failed: block.numel() * block.element_size() == row_bytes * nrows")
        src, dst = (table, block) if gather else (block, table)
        arr[k] = _lib.RowCopy(src.data_ptr(), dst.data_ptr(), row_bytes, nrows, 1 if gather else 0, _ctr(cursor), int(bias))
    check(lib().goalnet_rows_copy_batch(arr, len(segments), _s()), "rows_copy_batch")


def rows_scatter_tick(segments, counters, d0, d1, d2, d3, bad_step=None):
    """rows_copy_batch(segments) followed by counters_add4[_guarded] in ONE launch (the segments are small)"""
    arr = (_lib.RowCopy * len(segments))()
    for k, (table, block, nrows, cursor, bias, gather) in enumerate(segments):
        _chk(table, block)
        _req(table.is_contiguous() and block.is_contiguous() and table.dtype == block.dtype, "rows_scatter_tick: contiguous tensors of one dtype")
        row_bytes = table[0].numel() * table.element_size() if table.dim() > 1 else table.element_size()
        _req(block.numel() * block.element_size() == row_bytes * nrows, "rows_scatter_tick: block size")
        src, dst = (table, block) if gather else (block, table)
        arr[k] = _lib.RowCopy(src.data_ptr(), dst.data_ptr(), row_bytes, nrows, 1 if gather else 0, _ctr(cursor), int(bias))
    _req(counters.is_cuda and counters.dtype == I64 and counters.numel() == 4 and counters.is_contiguous(), "rows_scatter_tick: int64[4] counters")
    check(lib().goalnet_rows_scatter_tick(arr, len(segments), counters.data_ptr(), int(d0), int(d1), int(d2), int(d3),
                                          0 if bad_step is None else _ctr(bad_step), _s()), "rows_scatter_tick")


def rows_gather(table, block, nrows, cursor):
    """block[0:nrows] = table[cursor : cursor + nrows]; rows are the leading dimension of contiguous tensors."""
    _chk(table, block)
    _req(table.is_contiguous() and block.is_contiguous() and table.dtype == block.dtype, "rows_gather: argument check failed: table.is_contiguous() and block.is_contiguous() and table.dtype == block.dtype")
    row_bytes = table[0].numel() * table.element_size() if table.dim() > 1 else table.element_size()
    _req(block.numel() * block.element_size() == row_bytes * nrows, "rows_gather: argument check failed: block.numel() * block.element_size() == row_bytes * nrows")
    check(lib().goalnet_rows_gather(table.data_ptr(), block.data_ptr(), row_bytes, nrows, _ctr(cursor), _s()), "rows_gather")


def rows_scatter(block, table, nrows, cursor):
    """table[cursor : cursor + nrows] = block[0:nrows]"""
    _chk(table, block)
    _req(table.is_contiguous() and block.is_contiguous() and table.dtype == block.dtype, "rows_scatter: argument check failed: table.is_contiguous() and block.is_contiguous() and table.dtype == block.dtype")
    row_bytes = table[0].numel() * table.element_size() if table.dim() > 1 else table.element_size()
    _req(block.numel() * block.element_size() == row_bytes * nrows, "rows_scatter: argument check failed: block.numel() * block.element_size() == row_bytes * nrows")
    check(lib().goalnet_rows_scatter(block.data_ptr(), table.data_ptr(), row_bytes, nrows, _ctr(cursor), _s()), "rows_scatter")
