"""Functional wrappers: one call = one C-ABI op on torch CUDA tensors (NHWC float32).

These mirror the torch.nn.functional calls they stand in for (F.conv2d / F.group_norm / F.linear /
softmax-attention), with the fusions the kernels expose.  PyTorch is plumbing only here: it owns the
device memory and the stream; all arithmetic happens in libcdx.so.  The UNet builds the same argument
structs once and replays them (unet.py); these wrappers exist for tests and ad-hoc use.
"""
from __future__ import annotations

import torch

from . import _abi


def _stream(t=None) -> int:
    """The current HIP stream of tensor t's device (default: of the current device)."""
    return torch.cuda.current_stream(None if t is None else t.device).cuda_stream


def _call(op: str, args, ws_ptr, ws_bytes, t) -> None:
    """One C-ABI launch on the current stream of t's device, with that device current (hipLaunchKernelGGL launches on
    the CURRENT device: a tensor on cuda:1 while cuda:0 is current would otherwise be touched by a device-0 kernel)."""
    with torch.cuda.device(t.device):
        _abi.call(op, args, ws_ptr, ws_bytes, _stream(t))


def _ptr(t, byte_off: int = 0):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.int32, torch.float16, torch.bfloat16) and t.is_contiguous()
    return t.data_ptr() + byte_off


def _ws(nbytes: int, device):
    if nbytes == 0:
        return None, 0, None
    buf = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.float64, device=device)
    return buf.data_ptr(), nbytes, buf


class PackedConv:
    """Device-resident packed weights of one convolution (built once per layer)."""

    def __init__(self, w_oihw, bias, c0: int, c1: int = 0, device="cuda", winograd: bool = True, split: bool = True, up: bool = True):
        import numpy as np
        w = np.asarray(w_oihw, dtype=np.float32)
        self.cout, cin, self.ksize, _ = w.shape
        assert cin == c0 + c1, (cin, c0, c1)
        self.c0, self.c1 = c0, c1
        self.w = torch.from_numpy(_abi.pack_conv_weights(w, c0, c1)).to(device)
        # 3x3 layers also carry the Winograd-domain image; the library decides per launch which one to use
        self.w_wino = None
        if winograd and self.ksize == 3:
            self.w_wino = torch.from_numpy(_abi.pack_conv_weights_wino(w, c0, c1)).to(device)
        # ... and the fp16 hi|lo split image (float32 products on the fp16 matrix pipe, cdx.h CDX_TILE_SPLIT)
        self.w_split, self.split_unscale = None, 0.0
        if split:
            img, self.split_unscale = _abi.pack_conv_weights_split(w, c0, c1)
            self.w_split = torch.from_numpy(img).to(device)
        # ... and, for 3x3 layers, the four phase images that turn "nearest-2x upsample, then 3x3" into four 2x2 convolutions on the
        # low-resolution source (cdx.h wpacked_split_up; used only by launches with upsample=True)
        self.w_split_up, self.split_up_unscale = None, None
        if split and up and self.ksize == 3:
            img, self.split_up_unscale = _abi.pack_conv_weights_split_up(w, c0, c1)
            self.w_split_up = torch.from_numpy(img).to(device)
        self.bias = None if bias is None else torch.as_tensor(np.asarray(bias, np.float32)).to(device)


class PackedConv16:
    """fp16 (or bfloat16) fragment image of one convolution's weights (bias stays float32)."""

    def __init__(self, w_oihw, bias, c0: int, c1: int = 0, device="cuda", bf16: bool = False):
        import numpy as np
        w = np.asarray(w_oihw, dtype=np.float32)
        self.cout, cin, self.ksize, _ = w.shape
        assert cin == c0 + c1, (cin, c0, c1)
        self.c0, self.c1, self.bf16 = c0, c1, bf16
        self.dtype = torch.bfloat16 if bf16 else torch.float16
        if bf16:
            self.w = torch.from_numpy(_abi.pack_conv_weights_bf16(w, c0, c1).view(np.int16)).view(torch.bfloat16).to(device)
        else:
            self.w = torch.from_numpy(_abi.pack_conv_weights_f16(w, c0, c1)).to(device)
        self.bias = None if bias is None else torch.as_tensor(np.asarray(bias, np.float32)).to(device)


def conv16_args(pc: PackedConv16, src0, src1, out, *, stride=1, upsample=False, gn=None, silu=False,
                temb=None, temb_off=0, temb_ld=0, residual=None, out_ld=None) -> _abi.ConvF16Args:
    """fp16-storage convolution; src0 may be float32 (x_t buffer), out may be float32 (eps buffer)."""
    B, hin, win, c0 = src0.shape
    assert c0 == pc.c0 and (src1 is None) == (pc.c1 == 0)
    hv, wv = (hin * 2, win * 2) if upsample else (hin, win)
    hout, wout = (hv, wv) if stride == 1 else ((hv + 1) // 2, (wv + 1) // 2)
    a = _abi.ConvF16Args()
    a.src0, a.src1, a.c0, a.c1 = _ptr(src0), _ptr(src1), pc.c0, pc.c1
    a.src_is_f32 = int(src0.dtype == torch.float32)
    assert src1 is None or src1.dtype == src0.dtype
    a.batch, a.hin, a.win, a.hout, a.wout = B, hin, win, hout, wout
    a.cout, a.ksize, a.stride = pc.cout, pc.ksize, stride
    a.flags = (_abi.CONV_UPSAMPLE2X if upsample else 0) | (_abi.CONV_GN if gn is not None else 0) | (_abi.CONV_SILU if silu else 0) | \
        (_abi.CONV_BF16 if pc.bf16 else 0)
    assert src0.dtype in (torch.float32, pc.dtype) and out.dtype in (torch.float32, pc.dtype)
    a.wpacked, a.bias = _ptr(pc.w), _ptr(pc.bias)
    if gn is not None:
        a.gn_scale, a.gn_shift = _ptr(gn[0]), _ptr(gn[1])
    if temb is not None:
        a.temb, a.temb_ld = _ptr(temb, 4 * temb_off), temb_ld or temb.shape[-1]
    if residual is not None:
        assert residual.dtype == pc.dtype
    a.residual = _ptr(residual)
    a.out, a.out_is_f32, a.out_ld = _ptr(out), int(out.dtype == torch.float32), out_ld or out.shape[-1]
    assert out.shape[0] == B and out.shape[1] == hout and out.shape[2] == wout and a.out_ld >= pc.cout
    return a


def conv16_stats_buffer(a: _abi.ConvF16Args, device) -> torch.Tensor:
    import ctypes
    slots = _abi.lib().cdx_conv_f16_stats_slots(ctypes.byref(a))
    if slots <= 0:
        raise _abi.CdxError("cdx_conv_f16_stats_slots: bad arguments")
    buf = torch.full((a.batch, slots, a.cout, 2), float("nan"), dtype=torch.float64, device=device)     # (as conv_stats_buffer)
    a.stats_out = buf.data_ptr()
    return buf


def conv16(pc: PackedConv16, src0, src1=None, *, out_dtype=None, want_stats=False, **kw):
    out_dtype = out_dtype or pc.dtype
    B, hin, win, _ = src0.shape
    up, stride = kw.get("upsample", False), kw.get("stride", 1)
    hv, wv = (hin * 2, win * 2) if up else (hin, win)
    hout, wout = (hv, wv) if stride == 1 else ((hv + 1) // 2, (wv + 1) // 2)
    out = kw.pop("out", None)        # (tests: a caller-owned output, e.g. a view in front of a guard region)
    if out is None:
        out = torch.zeros(B, hout, wout, kw.get("out_ld") or pc.cout, device=src0.device, dtype=out_dtype)
    a = conv16_args(pc, src0, src1, out, **kw)
    stats = conv16_stats_buffer(a, src0.device) if want_stats else None
    _call("conv_f16", a, None, 0, src0)
    return (out, stats) if want_stats else out


def conv_args(pc: PackedConv, src0, src1, out, *, stride=1, upsample=False, gn=None, silu=False,
              temb=None, temb_off=0, temb_ld=0, residual=None, out_ld=None, stats=None,
              src_amax=None, amax_out=None) -> _abi.ConvArgs:
    """gn = (scale, shift) or (scale, shift, exp): exp = the out_exp the GroupNorm launch wrote them with (cdx.h gn_exp).
    src_amax = (amax0, amax1 | None): int32 [B, CDX_AMAX_WORDS] tensors (amax_buffer) with the float32 bit patterns of
    max |x| per image of the sources (cdx.h RANGE CONTRACT: an un-normalised launch takes the split-fp16 tile only with
    them); amax_out: such a tensor for the output."""
    B, hin, win, c0 = src0.shape
    assert c0 == pc.c0 and (src1 is None) == (pc.c1 == 0)
    if src1 is not None:
        assert tuple(src1.shape) == (B, hin, win, pc.c1)
    hv, wv = (hin * 2, win * 2) if upsample else (hin, win)
    hout, wout = (hv, wv) if stride == 1 else ((hv + 1) // 2, (wv + 1) // 2)
    flags = (_abi.CONV_UPSAMPLE2X if upsample else 0) | (_abi.CONV_GN if gn is not None else 0) | \
        (_abi.CONV_SILU if silu else 0)
    a = _abi.ConvArgs()
    a.src0, a.src1, a.c0, a.c1 = _ptr(src0), _ptr(src1), pc.c0, pc.c1
    a.batch, a.hin, a.win, a.hout, a.wout = B, hin, win, hout, wout
    a.cout, a.ksize, a.stride, a.flags = pc.cout, pc.ksize, stride, flags
    a.wpacked, a.bias = _ptr(pc.w), _ptr(pc.bias)
    a.wpacked_wino = _ptr(pc.w_wino)
    a.wpacked_split, a.wsplit_unscale = _ptr(pc.w_split), pc.split_unscale
    if upsample and pc.w_split_up is not None:
        a.wpacked_split_up = _ptr(pc.w_split_up)
        for i, u in enumerate(pc.split_up_unscale):
            a.wsplit_up_unscale[i] = u
    if gn is not None:
        a.gn_scale, a.gn_shift = _ptr(gn[0]), _ptr(gn[1])
        if len(gn) > 2:      # the exponent is STATED (cdx.h CDX_CONV_GN_EXP): only then may the launch take the split tile
            a.gn_exp, a.flags = int(gn[2]), a.flags | _abi.CONV_GN_EXP
    if src_amax is not None:
        a.src_amax0 = _ptr(src_amax[0])
        a.src_amax1 = _ptr(src_amax[1]) if len(src_amax) > 1 else None
    a.amax_out = _ptr(amax_out)
    if temb is not None:
        a.temb, a.temb_ld = _ptr(temb, 4 * temb_off), temb_ld or temb.shape[-1]
    a.residual = _ptr(residual)
    a.out, a.out_ld = _ptr(out), out_ld or out.shape[-1]
    assert out.shape[0] == B and out.shape[1] == hout and out.shape[2] == wout and a.out_ld >= pc.cout
    if stats is not None:          # float64 [B, slots, cout, 2] partial sums for gn_finalize (see conv_stats_buffer)
        assert stats.dtype == torch.float64 and stats.is_contiguous()
        a.stats_out, a.stats_slots = stats.data_ptr(), stats.shape[1]
    return a


def conv_stats_buffer(a: _abi.ConvArgs, device) -> torch.Tensor:
    """Allocate the stats_out buffer the library wants for this launch and attach it to the args."""
    import ctypes
    slots = _abi.lib().cdx_conv_stats_slots(ctypes.byref(a))
    if slots <= 0:
        raise _abi.CdxError("cdx_conv_stats_slots: bad arguments")
    # every entry is written by the launch (cdx.h): NaN-filled so that a slot a kernel skipped cannot pass for a zero sum
    buf = torch.full((a.batch, slots, a.cout, 2), float("nan"), dtype=torch.float64, device=device)
    a.stats_out, a.stats_slots = buf.data_ptr(), slots
    return buf


def gn_finalize_args(stats0, stats1, hw, gamma, beta, groups, scale, shift, eps=1e-5, mean=None, rstd=None, out_exp=0):
    a = _abi.GnFinalizeArgs()
    a.part0, a.slots0, a.c0 = stats0.data_ptr(), stats0.shape[1], stats0.shape[2]
    if stats1 is not None:
        a.part1, a.slots1, a.c1 = stats1.data_ptr(), stats1.shape[1], stats1.shape[2]
    a.batch, a.hw, a.groups, a.eps = stats0.shape[0], hw, groups, eps
    a.gamma, a.beta, a.scale, a.shift = _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift)
    a.mean, a.rstd = _ptr(mean), _ptr(rstd)
    a.out_exp = out_exp
    return a


def gn_finalize(stats0, stats1, hw, gamma, beta, groups, eps=1e-5, want_moments=False):
    B = stats0.shape[0]
    C = stats0.shape[2] + (0 if stats1 is None else stats1.shape[2])
    dev = stats0.device
    scale, shift = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
    mean = torch.empty(B, groups, device=dev) if want_moments else None
    rstd = torch.empty(B, groups, device=dev) if want_moments else None
    _call("gn_finalize_f32", gn_finalize_args(stats0, stats1, hw, gamma, beta, groups, scale, shift, eps, mean, rstd),
          None, 0, stats0)
    return (scale, shift, mean, rstd) if want_moments else (scale, shift)


def amax_buffer(batch: int, device) -> torch.Tensor:
    """Zeroed int32 [B, CDX_AMAX_WORDS] amax words (cdx.h src_amax0/1, amax_out; 64-byte aligned rows)."""
    return torch.zeros(batch, _abi.AMAX_WORDS, dtype=torch.int32, device=device)


def amax_value(words: torch.Tensor) -> torch.Tensor:
    """float32 [B] = max |x| per image from its amax words."""
    return words.view(torch.float32).amax(dim=-1)


def amax(x, channels: int | None = None, out=None):
    """int32 [B, CDX_AMAX_WORDS]: float32 bit patterns whose maximum per image is max |x[b]| over channels [0, channels)
    (cdx_amax_f32; max-combines into `out`)."""
    B, ld = x.shape[0], x.shape[-1]
    if out is None:
        out = amax_buffer(B, x.device)
    a = _abi.AmaxArgs(_ptr(x), ld, B, x.numel() // (B * ld), channels or ld, _ptr(out))
    _call("amax_f32", a, None, 0, x)
    return out


def check_finite(x, channels: int | None = None, limit: float = 0.0) -> int:
    """Debug: status word of cdx_check_finite_f32 over channels [0, channels) of an NHWC float32 tensor (syncs)."""
    ld = x.shape[-1]
    status = torch.zeros(1, dtype=torch.int32, device=x.device)
    a = _abi.CheckFiniteArgs(_ptr(x), ld, x.numel() // ld, channels or ld, limit, _ptr(status))
    _call("check_finite_f32", a, None, 0, x)
    return int(status.item())


def conv(pc: PackedConv, src0, src1=None, **kw):
    """out = conv(silu(gn(cat[src0, src1]))) + bias (+ temb) (+ residual); NHWC in, NHWC out.

    Range (cdx.h RANGE CONTRACT): an un-normalised launch gets the per-image maxima of its sources from one cdx_amax_f32
    pass per source here (the UNet takes them from the producing launches instead) unless `src_amax` is given or
    auto_range=False; a GroupNorm-ed launch takes its exponent from gn = (scale, shift, exp) (see gn_stats(act_exp=...)).
    debug=True: run cdx_check_finite_f32 over the output and raise on NaN / Inf."""
    B, hin, win, _ = src0.shape
    up, stride = kw.get("upsample", False), kw.get("stride", 1)
    hv, wv = (hin * 2, win * 2) if up else (hin, win)
    hout, wout = (hv, wv) if stride == 1 else ((hv + 1) // 2, (wv + 1) // 2)
    out = kw.pop("out", None)
    tile = kw.pop("tile", -1)        # diagnostics: force a tile shape (cdx.h CDX_TILE_*)
    debug = kw.pop("debug", False)
    auto_range = kw.pop("auto_range", True)
    if out is None:
        out = torch.empty(B, hout, wout, kw.get("out_ld") or pc.cout, device=src0.device, dtype=torch.float32)
        if out.shape[-1] != pc.cout:
            out.zero_()
    want_stats = kw.pop("want_stats", False)
    want_amax = kw.pop("want_amax", False)
    if want_amax:
        kw["amax_out"] = amax_buffer(B, src0.device)
    gn_affine = kw.pop("gn_affine", None)     # (gamma, beta, groups): GroupNorm computed here, with the exponent the tile wants
    import ctypes
    if gn_affine is not None:
        gamma, beta, groups = gn_affine
        C = src0.shape[-1] + (0 if src1 is None else src1.shape[-1])
        sc, sh = torch.empty(B, C, device=src0.device), torch.empty(B, C, device=src0.device)
        e = gn_act_exp(gamma, beta, groups, hin * win) if pc.w_split is not None else 0
        probe = conv_args(pc, src0, src1, out, gn=(sc, sh, e), **{k: v for k, v in kw.items() if k != "gn"})
        split = tile == _abi.TILE_SPLIT or (tile < 0 and _abi.lib().cdx_conv_select_tile(ctypes.byref(probe)) == _abi.TILE_SPLIT)
        e = e if split else 0
        ga = gn_stats_args(src0, src1, gamma, beta, groups, sc, sh, out_exp=e)
        wp, wb, keep = _ws(_abi.workspace_bytes("gn_stats_f32", ga), src0.device)
        _call("gn_stats_f32", ga, wp, wb, src0)
        kw["gn"] = (sc, sh, e) if split else (sc, sh)      # (the exponent is STATED only to the tile that undoes it: cdx.h CDX_CONV_GN_EXP)
    if auto_range and kw.get("gn") is None and kw.get("src_amax") is None and pc.w_split is not None:
        kw["src_amax"] = (amax(src0),) + ((amax(src1),) if src1 is not None else ())
    a = conv_args(pc, src0, src1, out, **kw)
    stats = conv_stats_buffer(a, src0.device) if want_stats else None
    with torch.cuda.device(src0.device):
        _abi.check(_abi.lib().cdx_conv_f32_tile(ctypes.byref(a), tile, None, 0, _stream(src0)), "cdx_conv_f32_tile")
    if debug:
        st = check_finite(out, pc.cout)
        if st:
            raise _abi.CdxError(f"conv (debug): output holds non-finite values (status {st})")
    res = (out,) + ((stats,) if want_stats else ()) + ((kw["amax_out"],) if want_amax else ())
    return res if len(res) > 1 else out


def gn_stats_args(src0, src1, gamma, beta, groups, scale, shift, eps=1e-5, mean=None, rstd=None, out_exp=0) -> _abi.GnStatsArgs:
    B = src0.shape[0]
    c0 = src0.shape[-1]
    hw = src0.numel() // (B * c0)
    a = _abi.GnStatsArgs()
    a.src0, a.src1, a.c0, a.c1 = _ptr(src0), _ptr(src1), c0, (0 if src1 is None else src1.shape[-1])
    a.batch, a.hw, a.groups, a.eps = B, hw, groups, eps
    a.gamma, a.beta, a.scale, a.shift = _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift)
    a.mean, a.rstd = _ptr(mean), _ptr(rstd)
    a.out_exp = out_exp
    return a


def gn_act_exp(gamma, beta, groups: int, hw: int) -> int:
    """The static activation exponent of a GroupNorm-ed split-tile input (cdx_gn_act_exp) from device / host gamma, beta."""
    return _abi.gn_act_exp(gamma.detach().cpu().numpy(), beta.detach().cpu().numpy(), groups, hw)


def gn_stats(src0, src1, gamma, beta, groups, eps=1e-5, want_moments=False, act_exp=None):
    """(scale[B,C], shift[B,C]) such that group_norm(x)[b,...,c] = x*scale[b,c] + shift[b,c].
    act_exp = "auto" | int: scale / shift are written times 2^exp and the result is (scale, shift, exp) -- pass it whole
    as gn= to a convolution that takes the split tile (cdx.h gn_exp); the f32-MFMA tiles need the plain pair."""
    B = src0.shape[0]
    C = src0.shape[-1] + (0 if src1 is None else src1.shape[-1])
    dev = src0.device
    scale = torch.empty(B, C, device=dev)
    shift = torch.empty(B, C, device=dev)
    mean = torch.empty(B, groups, device=dev) if want_moments else None
    rstd = torch.empty(B, groups, device=dev) if want_moments else None
    hw = src0.numel() // (B * src0.shape[-1])
    e = 0 if act_exp is None else gn_act_exp(gamma, beta, groups, hw) if act_exp == "auto" else int(act_exp)
    a = gn_stats_args(src0, src1, gamma, beta, groups, scale, shift, eps, mean, rstd, out_exp=e)
    wp, wb, keep = _ws(_abi.workspace_bytes("gn_stats_f32", a), dev)
    _call("gn_stats_f32", a, wp, wb, src0)
    if act_exp is not None:
        assert not want_moments
        return scale, shift, e
    return (scale, shift, mean, rstd) if want_moments else (scale, shift)


def attn_args(q, k, v, out, *, batch, nq, nk, heads, head_dim, q_ld, k_ld, v_ld, out_ld,
              q_off=0, k_off=0, v_off=0, scale=None) -> _abi.AttnArgs:
    a = _abi.AttnArgs()
    es = q.element_size()
    a.q, a.q_ld = _ptr(q, es * q_off), q_ld
    a.k, a.k_ld = _ptr(k, es * k_off), k_ld
    a.v, a.v_ld = _ptr(v, es * v_off), v_ld
    a.batch, a.nq, a.nk, a.heads, a.head_dim = batch, nq, nk, heads, head_dim
    a.scale = scale if scale is not None else head_dim ** -0.5
    a.out, a.out_ld = _ptr(out), out_ld
    return a


ATTN_OP = {torch.float32: "attn_f32", torch.float16: "attn_f16", torch.bfloat16: "attn_bf16"}


def attention(q, k, v, heads: int, head_dim: int = 64):
    """q [B,Nq,C], k/v [B,Nk,C] (C = heads*head_dim) -> softmax(q k^T / sqrt(d)) v  [B,Nq,C]; float32 or float16 I/O."""
    B, nq, c = q.shape
    nk = k.shape[1]
    out = torch.empty(B, nq, c, device=q.device, dtype=q.dtype)
    a = attn_args(q, k, v, out, batch=B, nq=nq, nk=nk, heads=heads, head_dim=head_dim,
                  q_ld=c, k_ld=k.shape[-1], v_ld=v.shape[-1], out_ld=c)
    _call(ATTN_OP[q.dtype], a, None, 0, q)
    return out


def linear_args(x, w, bias, out, *, silu_in=False, m=None, out_off=0, out_ld=None) -> _abi.LinearArgs:
    a = _abi.LinearArgs()
    a.x, a.x_ld = _ptr(x), x.shape[-1]
    a.w, a.bias = _ptr(w), _ptr(bias)
    a.m, a.n, a.k = (m or x.shape[0]), w.shape[0], w.shape[1]
    a.flags = _abi.LINEAR_SILU_IN if silu_in else 0
    a.out, a.out_ld = _ptr(out, 4 * out_off), out_ld or out.shape[-1]
    return a


def linear(x, w, bias=None, silu_in=False):
    out = torch.empty(x.shape[0], w.shape[0], device=x.device)
    _call("linear_f32", linear_args(x, w, bias, out, silu_in=silu_in), None, 0, x)
    return out


def timestep_embedding(t, dim: int):
    out = torch.empty(t.shape[0], dim, device=t.device)
    a = _abi.TimestepEmbeddingArgs(_ptr(t), t.shape[0], dim, _ptr(out))
    _call("timestep_embedding_f32", a, None, 0, t)
    return out


def gauss_fill(x, channels: int, seed: int, first_image: int, noise_stream: int):
    """x [B, HW.., ld] NHWC buffer: channels [0, channels) <- N(0,1) of stream (seed, image, noise_stream)."""
    B, ld = x.shape[0], x.shape[-1]
    a = _abi.GaussFillArgs(_ptr(x), ld, B, x.numel() // (B * ld), channels, seed, first_image, noise_stream)
    _call("gauss_fill_f32", a, None, 0, x)
    return x


def diffusion_update(x, eps, channels, coef, *, clip_x0=True, seed=0, first_image=0, noise_stream=0):
    B, ld = x.shape[0], x.shape[-1]
    a = _abi.DiffusionUpdateArgs(_ptr(x), ld, _ptr(eps), eps.shape[-1], B, x.numel() // (B * ld), channels,
                                 coef.ca, coef.cb, coef.cx, coef.c0, coef.ce, coef.sigma, int(clip_x0),
                                 seed, first_image, noise_stream)
    _call("diffusion_update_f32", a, None, 0, x)
    return x


def cond_embed(cond_nchw, x, c_off: int):
    B, cc, hc, wc = cond_nchw.shape
    _, h, w, ld = x.shape
    a = _abi.CondEmbedArgs(_ptr(cond_nchw), cc, hc, wc, _ptr(x), ld, c_off, B, h, w)
    _call("cond_embed_f32", a, None, 0, x)
    return x


def export_image(x, channels: int, lo=-1.0, hi=1.0):
    B, h, w, ld = x.shape
    out = torch.empty(B, channels, h, w, device=x.device)
    a = _abi.ExportImageArgs(_ptr(x), ld, B, h * w, channels, lo, hi, _ptr(out))
    _call("export_image_f32", a, None, 0, x)
    return out
