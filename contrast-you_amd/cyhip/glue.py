"""The glue of the reference's second backbone (`UNet2`, contrastyou/arch/unet2.py) on the HIP kernels of
csrc/cy_unet2.hip: K x K / 1 x 1 / transposed convolutions as im2col + strided GEMM + col2im, channel LayerNorm,
the linear attention and the bottleneck's softmax attention.  f32, NHWC; torch supplies memory, views and the
autograd graph -- every arithmetic launch below is a hand-written kernel (the weight re-layouts of a few KB and the
[N, heads, 32] closed-form term of the k-softmax backward are the only torch arithmetic left).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib, ops
from ._lib import MatLayout

__all__ = ["gemm", "LinearRowsFn", "Conv2dFn", "ConvTranspose2dFn", "ChanLayerNormFn", "LinearAttentionFn",
           "AttentionFn", "rows_view"]


def _f32c(t: Tensor) -> Tensor:
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def rows_view(x: Tensor) -> Tensor:
    """[N, C, H, W] (any memory format, any float dtype) -> f32 [N, H, W, C] contiguous (a view when it already is)"""
    x = ops.to_nhwc(x if x.dtype == torch.float32 else x.float())
    return x.permute(0, 2, 3, 1)


def _as_nchw(rows: Tensor) -> Tensor:
    """f32 [N, H, W, C] contiguous -> logical [N, C, H, W] over the same (channels_last) memory"""
    return rows.permute(0, 3, 1, 2)


def _auto_ksplit(M: int, N: int, K: int, nbatch: int) -> int:
    tiles = ((M + 63) // 64) * ((N + 63) // 64) * nbatch
    if tiles >= 256 or K < 2048:
        return 1
    return max(1, min(64, (512 + tiles - 1) // tiles, K // 512, 65535 // max(nbatch, 1)))


def gemm(A: Tuple[Tensor, int], la: MatLayout, B: Tuple[Tensor, int], lb: MatLayout, Cm: Tuple[Tensor, int],
         lc: MatLayout, M: int, N: int, K: int, *, bias: Optional[Tensor] = None, nb1: int = 1, nb2: int = 1,
         alpha: float = 1.0, accumulate: bool = False, ksplit: Optional[int] = None) -> None:
    """C = alpha * A * B (+ bias) (+ C) through cy_gemm_strided; operands are (f32 tensor, element offset) pairs"""
    (a, ao), (b, bo), (c, co) = A, B, Cm
    ops.require_gpu(a, b, c)
    assert a.dtype == b.dtype == c.dtype == torch.float32
    nbatch = nb1 * nb2
    ks = _auto_ksplit(M, N, K, nbatch) if ksplit is None else ksplit
    nbytes = _lib.load().cy_gemm_strided_ws_bytes(M, N, nbatch, ks)
    ws = ops._ws(nbytes, c.device) if nbytes else None
    _lib.call("cy_gemm_strided", a.data_ptr() + 4 * ao, C.byref(la), b.data_ptr() + 4 * bo, C.byref(lb),
              c.data_ptr() + 4 * co, C.byref(lc), ops._ptr(bias), M, N, K, nb1, nb2, float(alpha), int(accumulate), ks,
              ops._ptr(ws), nbytes, ops._stream())


def _colsum(x2d: Tensor) -> Tensor:
    M, N = x2d.shape
    out = ops._f32(N, x2d.device)
    nbytes = _lib.load().cy_colsum_ws_bytes(M, N)
    ws = ops._ws(nbytes, x2d.device)
    _lib.call("cy_colsum", x2d.data_ptr(), out.data_ptr(), M, N, 0, ws.data_ptr(), nbytes, ops._stream())
    return out


def _rowmajor(cols: int) -> MatLayout:
    return MatLayout(cols, 1, 0, 0)


def _transposed(cols: int) -> MatLayout:
    """layout of X^T for a row-major X with `cols` columns"""
    return MatLayout(1, cols, 0, 0)


def _linear_fwd(x2d: Tensor, w2d: Tensor, bias: Optional[Tensor]) -> Tensor:
    M, I = x2d.shape
    O = w2d.shape[0]
    y = ops._f32(M * O, x2d.device).view(M, O)
    gemm((x2d, 0), _rowmajor(I), (w2d, 0), _transposed(I), (y, 0), _rowmajor(O), M, O, I, bias=bias)
    return y


def _linear_bwd(x2d: Tensor, w2d: Tensor, dy: Tensor, need_dx: bool, need_dw: bool, need_db: bool):
    M, I = x2d.shape
    O = w2d.shape[0]
    dx = dw = db = None
    if need_dx:
        dx = ops._f32(M * I, dy.device).view(M, I)
        gemm((dy, 0), _rowmajor(O), (w2d, 0), _rowmajor(I), (dx, 0), _rowmajor(I), M, I, O)
    if need_dw:
        dw = ops._f32(O * I, dy.device).view(O, I)
        gemm((dy, 0), _transposed(O), (x2d, 0), _rowmajor(I), (dw, 0), _rowmajor(I), O, I, M)
    if need_db:
        db = _colsum(dy)
    return dx, dw, db


class LinearRowsFn(torch.autograd.Function):
    """y[m] = W x[m] + b over the rows of a [M, I] matrix: a 1x1 convolution on an NHWC map"""

    @staticmethod
    def forward(ctx, x2d: Tensor, w2d: Tensor, b: Optional[Tensor]):
        x2d, w = _f32c(x2d), _f32c(w2d)
        ctx.save_for_backward(x2d, w)
        ctx.has_bias = b is not None
        return _linear_fwd(x2d, w, None if b is None else _f32c(b))

    @staticmethod
    def backward(ctx, dy: Tensor):
        x2d, w = ctx.saved_tensors
        n = ctx.needs_input_grad
        dx, dw, db = _linear_bwd(x2d, w, _f32c(dy), n[0], n[1], ctx.has_bias and n[2])
        return dx, dw, db


def _im2col(xr: Tensor, KH: int, KW: int, stride: int, pad: int) -> Tuple[Tensor, int, int]:
    N, H, W, Cc = xr.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    cols = ops._f32(N * Ho * Wo * KH * KW * Cc, xr.device).view(N * Ho * Wo, KH * KW * Cc)
    _lib.call("cy_im2col", xr.data_ptr(), cols.data_ptr(), N, H, W, Cc, KH, KW, stride, pad, ops._stream())
    return cols, Ho, Wo


def _col2im(cols: Tensor, bias: Optional[Tensor], N: int, H: int, W: int, Cc: int, KH: int, KW: int, stride: int,
            pad: int) -> Tensor:
    out = torch.empty((N, H, W, Cc), dtype=torch.float32, device=cols.device)
    _lib.call("cy_col2im", cols.data_ptr(), ops._ptr(bias), out.data_ptr(), N, H, W, Cc, KH, KW, stride, pad,
              ops._stream())
    return out


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d(Cin, Cout, K, stride, pad) with bias, NHWC f32: im2col + GEMM; backward = two GEMMs + col2im.
    The patch matrix is rebuilt in backward instead of being kept (it is K*K times the input)."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int, pad: int):
        ops.require_gpu(x, weight)
        xr = rows_view(x).contiguous()
        N, H, W, Cin = xr.shape
        Cout, _, KH, KW = weight.shape
        w2 = _f32c(weight).permute(0, 2, 3, 1).reshape(Cout, KH * KW * Cin).contiguous()
        plain = KH == 1 and KW == 1 and stride == 1 and pad == 0
        if plain:
            cols, Ho, Wo = xr.view(N * H * W, Cin), H, W
        else:
            cols, Ho, Wo = _im2col(xr, KH, KW, stride, pad)
        y = _linear_fwd(cols, w2, None if bias is None else _f32c(bias))
        ctx.save_for_backward(xr, w2)
        ctx.geom = (KH, KW, stride, pad, plain, Ho, Wo, bias is not None)
        return _as_nchw(y.view(N, Ho, Wo, Cout))

    @staticmethod
    def backward(ctx, dy: Tensor):
        xr, w2 = ctx.saved_tensors
        KH, KW, stride, pad, plain, Ho, Wo, has_bias = ctx.geom
        N, H, W, Cin = xr.shape
        Cout, K2 = w2.shape
        Mo = N * Ho * Wo
        dyr = rows_view(dy).contiguous().view(Mo, Cout)
        need = ctx.needs_input_grad
        dx = dweight = db = None
        if need[1]:
            cols = xr.view(N * H * W, Cin) if plain else _im2col(xr, KH, KW, stride, pad)[0]
            dw2 = ops._f32(Cout * K2, dy.device).view(Cout, K2)
            gemm((dyr, 0), _transposed(Cout), (cols, 0), _rowmajor(K2), (dw2, 0), _rowmajor(K2), Cout, K2, Mo)
            dweight = dw2.view(Cout, KH, KW, Cin).permute(0, 3, 1, 2)
        if has_bias and need[2]:
            db = _colsum(dyr)
        if need[0]:
            dcols = ops._f32(Mo * K2, dy.device).view(Mo, K2)
            gemm((dyr, 0), _rowmajor(Cout), (w2, 0), _rowmajor(K2), (dcols, 0), _rowmajor(K2), Mo, K2, Cout)
            dx = _as_nchw(dcols.view(N, H, W, Cin) if plain else _col2im(dcols, None, N, H, W, Cin, KH, KW, stride, pad))
        return dx, dweight, db, None, None


class ConvTranspose2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d(Cin, Cout, K, stride, pad) with bias (Upsample of unet2.py:176-177): GEMM + col2im"""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int, pad: int):
        ops.require_gpu(x, weight)
        xr = rows_view(x).contiguous()
        N, H, W, Cin = xr.shape
        _, Cout, KH, KW = weight.shape
        wt = _f32c(weight).permute(2, 3, 1, 0).reshape(KH * KW * Cout, Cin).contiguous()  # [(kh, kw, co)][ci]
        cols = _linear_fwd(xr.view(N * H * W, Cin), wt, None)
        Hout, Wout = (H - 1) * stride - 2 * pad + KH, (W - 1) * stride - 2 * pad + KW
        y = _col2im(cols, None if bias is None else _f32c(bias), N, Hout, Wout, Cout, KH, KW, stride, pad)
        ctx.save_for_backward(xr, wt)
        ctx.geom = (KH, KW, stride, pad, Cout, bias is not None)
        return _as_nchw(y)

    @staticmethod
    def backward(ctx, dy: Tensor):
        xr, wt = ctx.saved_tensors
        KH, KW, stride, pad, Cout, has_bias = ctx.geom
        N, H, W, Cin = xr.shape
        need = ctx.needs_input_grad
        dyr = rows_view(dy).contiguous()
        dcols, Ho, Wo = _im2col(dyr, KH, KW, stride, pad)  # rows = the positions of the SMALL map
        assert (Ho, Wo) == (H, W)
        dx, dwt, _ = _linear_bwd(xr.view(N * H * W, Cin), wt, dcols, need[0], need[1], False)
        db = _colsum(dyr.view(-1, Cout)) if (has_bias and need[2]) else None
        dweight = None if dwt is None else dwt.view(KH, KW, Cout, Cin).permute(3, 2, 0, 1)
        return (None if dx is None else _as_nchw(dx.view(N, H, W, Cin))), dweight, db, None, None


class ChanLayerNormFn(torch.autograd.Function):
    """LayerNorm over the channel axis of a map with [1, C, 1, 1] affine parameters (unet2.py:183-194)"""

    @staticmethod
    def forward(ctx, x: Tensor, g: Tensor, b: Tensor, eps: float):
        ops.require_gpu(x, g, b)
        xr = rows_view(x).contiguous()
        N, H, W, Cc = xr.shape
        g1, b1 = _f32c(g).reshape(-1), _f32c(b).reshape(-1)
        y = torch.empty_like(xr)
        _lib.call("cy_chan_layernorm_fwd", xr.data_ptr(), g1.data_ptr(), b1.data_ptr(), y.data_ptr(), N * H * W, Cc,
                  float(eps), ops._stream())
        ctx.save_for_backward(xr, g1)
        ctx.eps, ctx.pshape = eps, tuple(g.shape)
        return _as_nchw(y)

    @staticmethod
    def backward(ctx, dy: Tensor):
        xr, g1 = ctx.saved_tensors
        N, H, W, Cc = xr.shape
        M = N * H * W
        dyr = rows_view(dy).contiguous()
        dx = torch.empty_like(xr)
        dg, db = ops._f32(Cc, dy.device), ops._f32(Cc, dy.device)
        nbytes = _lib.load().cy_chan_layernorm_bwd_ws_bytes(M, Cc)
        ws = ops._ws(nbytes, dy.device)
        _lib.call("cy_chan_layernorm_bwd", xr.data_ptr(), g1.data_ptr(), dyr.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                  db.data_ptr(), M, Cc, float(ctx.eps), ws.data_ptr(), nbytes, ops._stream())
        return _as_nchw(dx), dg.view(ctx.pshape), db.view(ctx.pshape), None


class LinearAttentionFn(torch.autograd.Function):
    """unet2.py:258-271 between `to_qkv` and `to_out`: q.softmax over a head's channels * scale, k.softmax over the
    positions, context = k v^T per (image, head), out = context^T q.  qkv: [N, 3*heads*dh, H, W]."""

    @staticmethod
    def forward(ctx, qkv: Tensor, heads: int, dh: int, scale: float):
        ops.require_gpu(qkv)
        qr = rows_view(qkv).contiguous()
        N, H, W, ld = qr.shape
        hid, n = heads * dh, H * W
        assert ld == 3 * hid
        M, dev = N * n, qr.device
        Qs = ops._f32(M * hid, dev).view(M, hid)
        _lib.call("cy_head_softmax_fwd", qr.data_ptr(), ld, 0, Qs.data_ptr(), M, heads, dh, float(scale), ops._stream())
        Ks = ops._f32(M * hid, dev).view(M, hid)
        nbytes = _lib.load().cy_col_softmax_ws_bytes(N, n, hid)
        ws = ops._ws(nbytes, dev)
        _lib.call("cy_col_softmax_fwd", qr.data_ptr(), ld, hid, Ks.data_ptr(), N, n, hid, ws.data_ptr(), nbytes,
                  ops._stream())
        ctxm = ops._f32(N * heads * dh * dh, dev).view(N, heads, dh, dh)
        l_ctx = MatLayout(dh, 1, heads * dh * dh, dh * dh)
        # context[b, h, d, e] = sum_p Ks[(b, p), h*dh + d] * v[(b, p), h*dh + e]
        gemm((Ks, 0), MatLayout(1, hid, n * hid, dh), (qr, 2 * hid), MatLayout(ld, 1, n * ld, dh), (ctxm, 0), l_ctx,
             dh, dh, n, nb1=N, nb2=heads)
        out = torch.empty((N, H, W, hid), dtype=torch.float32, device=dev)
        # out[(b, p), h*dh + e] = sum_d Qs[(b, p), h*dh + d] * context[b, h, d, e]
        gemm((Qs, 0), MatLayout(hid, 1, n * hid, dh), (ctxm, 0), l_ctx, (out, 0), MatLayout(hid, 1, n * hid, dh),
             n, dh, dh, nb1=N, nb2=heads)
        ctx.save_for_backward(qr, Qs, Ks, ctxm)
        ctx.cfg = (heads, dh, float(scale))
        return _as_nchw(out)

    @staticmethod
    def backward(ctx, dout: Tensor):
        qr, Qs, Ks, ctxm = ctx.saved_tensors
        heads, dh, scale = ctx.cfg
        N, H, W, ld = qr.shape
        hid, n = heads * dh, H * W
        M, dev = N * n, qr.device
        dO = rows_view(dout).contiguous().view(M, hid)
        l_ctx = MatLayout(dh, 1, heads * dh * dh, dh * dh)
        l_ctx_t = MatLayout(1, dh, heads * dh * dh, dh * dh)
        l_rows = MatLayout(hid, 1, n * hid, dh)       # a head's block of a [M, hid] matrix
        l_rows_t = MatLayout(1, hid, n * hid, dh)
        l_qkv = MatLayout(ld, 1, n * ld, dh)
        dctx = torch.empty_like(ctxm)
        gemm((Qs, 0), l_rows_t, (dO, 0), l_rows, (dctx, 0), l_ctx, dh, dh, n, nb1=N, nb2=heads)
        dQs = ops._f32(M * hid, dev).view(M, hid)
        gemm((dO, 0), l_rows, (ctxm, 0), l_ctx_t, (dQs, 0), l_rows, n, dh, dh, nb1=N, nb2=heads)
        dqkv = torch.empty_like(qr)
        gemm((Ks, 0), l_rows, (dctx, 0), l_ctx, (dqkv, 2 * hid), l_qkv, n, dh, dh, nb1=N, nb2=heads)      # dv
        dKs = ops._f32(M * hid, dev).view(M, hid)
        gemm((qr, 2 * hid), l_qkv, (dctx, 0), l_ctx_t, (dKs, 0), l_rows, n, dh, dh, nb1=N, nb2=heads)
        # sum_p Ks * dKs per (image, channel) in closed form: sum_e dcontext[d, e] * context[d, e]
        t = (dctx * ctxm).sum(-1).reshape(N, hid).contiguous()
        _lib.call("cy_head_softmax_bwd", Qs.data_ptr(), dQs.data_ptr(), dqkv.data_ptr(), ld, 0, M, heads, dh, scale,
                  ops._stream())
        _lib.call("cy_col_softmax_bwd", Ks.data_ptr(), dKs.data_ptr(), t.data_ptr(), dqkv.data_ptr(), ld, hid, N, n,
                  hid, ops._stream())
        return _as_nchw(dqkv), None, None, None


class AttentionFn(torch.autograd.Function):
    """unet2.py:289-302 between `to_qkv` and `to_out`: softmax(scale * q^T k) v over all positions of the map"""

    @staticmethod
    def forward(ctx, qkv: Tensor, heads: int, dh: int, scale: float):
        ops.require_gpu(qkv)
        qr = rows_view(qkv).contiguous()
        N, H, W, ld = qr.shape
        hid, n = heads * dh, H * W
        assert ld == 3 * hid
        dev = qr.device
        l_qkv = MatLayout(ld, 1, n * ld, dh)
        l_qkv_t = MatLayout(1, ld, n * ld, dh)
        l_att = MatLayout(n, 1, heads * n * n, n * n)
        attn = ops._f32(N * heads * n * n, dev).view(N, heads, n, n)
        gemm((qr, 0), l_qkv, (qr, hid), l_qkv_t, (attn, 0), l_att, n, n, dh, nb1=N, nb2=heads, alpha=scale)
        _lib.call("cy_row_softmax_fwd", attn.data_ptr(), N * heads * n, n, ops._stream())
        out = torch.empty((N, H, W, hid), dtype=torch.float32, device=dev)
        gemm((attn, 0), l_att, (qr, 2 * hid), l_qkv, (out, 0), MatLayout(hid, 1, n * hid, dh), n, dh, n, nb1=N,
             nb2=heads)
        ctx.save_for_backward(qr, attn)
        ctx.cfg = (heads, dh, float(scale))
        return _as_nchw(out)

    @staticmethod
    def backward(ctx, dout: Tensor):
        qr, attn = ctx.saved_tensors
        heads, dh, scale = ctx.cfg
        N, H, W, ld = qr.shape
        hid, n = heads * dh, H * W
        dO = rows_view(dout).contiguous().view(N * n, hid)
        l_qkv = MatLayout(ld, 1, n * ld, dh)
        l_qkv_t = MatLayout(1, ld, n * ld, dh)
        l_att = MatLayout(n, 1, heads * n * n, n * n)
        l_att_t = MatLayout(1, n, heads * n * n, n * n)
        l_rows = MatLayout(hid, 1, n * hid, dh)
        dqkv = torch.empty_like(qr)
        dsim = torch.empty_like(attn)
        gemm((dO, 0), l_rows, (qr, 2 * hid), l_qkv_t, (dsim, 0), l_att, n, n, dh, nb1=N, nb2=heads)        # d attn
        gemm((attn, 0), l_att_t, (dO, 0), l_rows, (dqkv, 2 * hid), l_qkv, n, dh, n, nb1=N, nb2=heads)      # dv
        _lib.call("cy_row_softmax_bwd", attn.data_ptr(), dsim.data_ptr(), N * heads * n, n, ops._stream())
        gemm((dsim, 0), l_att, (qr, hid), l_qkv, (dqkv, 0), l_qkv, n, dh, n, nb1=N, nb2=heads, alpha=scale)    # dq
        gemm((dsim, 0), l_att_t, (qr, 0), l_qkv, (dqkv, hid), l_qkv, n, dh, n, nb1=N, nb2=heads, alpha=scale)  # dk
        return _as_nchw(dqkv), None, None, None
