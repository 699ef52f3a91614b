"""UNet(cfg).forward(x, t, cond) -- the eps-prediction network (U0 of SURVEY.md section 8a) on the HIP path.

Host side only: the constructor packs and uploads the weights; `plan(batch)` allocates every activation
buffer ONCE (NHWC, resident in HBM for the whole sampling loop -- 288 GB makes reuse games pointless)
and records the forward pass as a flat list of (C-ABI entry point, argument struct) pairs; `run()`
replays that list on the current HIP stream with no Python-side tensor work, allocation or sync.
Every arithmetic operation is a libcdx.so kernel; importing this module without the built library
raises (no CPU fallback).

API contract: BASELINE.json north_star ("same UNet config dict, same sample(cond, steps) entry point");
the reference snapshot defines neither (README.md: 0 bytes), SURVEY.md Appendix A does.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _abi, ops
from .config import validate_unet_config
from .graph import build_graph
from .params import init_params


def _pad4(c: int) -> int:
    return (c + 3) // 4 * 4


def _pad8(c: int) -> int:
    """x_t | cond buffer / conv_in input channels: a multiple of 8, so that conv_in takes the split tile in every cond mode (the
    3-channel x_t of the cross-attention configuration padded to 4 went to the f32-MFMA first-group tile: 1.0 of cfg4's 101 ms)."""
    return (c + 7) // 8 * 8


class _Plan:
    """All buffers + the recorded launch list for one batch size."""

    def __init__(self, net: "UNet", batch: int):
        self.net, self.batch = net, batch
        cfg, g, dev = net.cfg, net.graph, net.device
        B, H, ch = batch, cfg["image_size"], cfg["base_channels"]
        self.calls = []          # (bound C function, args struct, ws_ptr, ws_bytes)
        self.names = []          # layer name per recorded call (error messages, debug mode)
        self.outs = {}           # conv layer name -> its output buffer (diagnostics)
        self._keep = []
        self._ws_need = 0
        self._ws_calls = []      # indices of calls that use the shared workspace
        new = lambda *shape: self._hold(torch.empty(*shape, device=dev, dtype=torch.float32))  # noqa: E731
        half = net.half                      # 16-bit storage (fp16 / bf16) and 16-bit MFMA for the UNet's activations and weights
        adt = net.adt
        newa = lambda *shape: self._hold(torch.empty(*shape, device=dev, dtype=adt))  # noqa: E731  (activations)

        self.xin_ld = _pad8(g.cin_total) if not net.half else _pad4(g.cin_total)
        self.xin = self._hold(torch.zeros(B, H, H, self.xin_ld, device=dev))     # x_t | cond | 0
        self.eps = self._hold(torch.zeros(B, H, H, _pad4(cfg["out_channels"]), device=dev))
        self.t = self._hold(torch.zeros(B, dtype=torch.int32, device=dev))
        cross = cfg["cond_mode"] == "cross_attn"
        self.ctx = None

        # ---- timestep embedding MLP + every ResBlock's temb projection in one linear ----
        emb0, emb1, emb2 = new(B, ch), new(B, g.temb_dim), new(B, g.temb_dim)
        self.tproj = new(B, net.tproj_w.shape[0])
        self._add("timestep_embedding_f32", _abi.TimestepEmbeddingArgs(self.t.data_ptr(), B, ch, emb0.data_ptr()))
        self._add("linear_f32", ops.linear_args(emb0, net.dev["temb.0.weight"], net.dev["temb.0.bias"], emb1))
        self._add("linear_f32", ops.linear_args(emb1, net.dev["temb.2.weight"], net.dev["temb.2.bias"], emb2, silu_in=True))
        self._add("linear_f32", ops.linear_args(emb2, net.tproj_w, net.tproj_b, self.tproj, silu_in=True))

        stats_of = {}            # data_ptr of a conv output -> its fused partial-sum buffer
        self.stats_of, self.gn_of = stats_of, {}      # (diagnostics: sums per tensor, (scale, shift, exp) per norm)
        # Range bookkeeping of the split-fp16 tiles (cdx.h RANGE CONTRACT).  CDX_AMAX_WORDS int32 words per (tensor, image) in
        # ONE arena, zeroed by the first launch of every forward: float32 bit patterns whose maximum is max |x| of that image,
        # written by the producing convolution's epilogue (amax_out) or by a cdx_amax_f32 pass for tensors no convolution produced.
        split = net.split and not half
        # (at most one row per convolution output + the x_t | cond buffer + one per attention output)
        self.amax_arena = self._hold(torch.zeros(2 * len(net.convs) + 16 if split else 1, B, _abi.AMAX_WORDS, dtype=torch.int32, device=dev))
        amax_slot = {}           # data_ptr of a tensor -> its row of the arena
        need_amax = set()        # data_ptrs of conv outputs an un-normalised split launch reads
        produced = {}            # data_ptr of a conv output -> the args struct of the launch that writes it

        def amax_of(t):
            k = t.data_ptr()
            if k not in amax_slot:
                amax_slot[k] = self.amax_arena[len(amax_slot)]
            return amax_slot[k]

        if split:
            self._add("fill_u32", _abi.FillU32Args(self.amax_arena.data_ptr(), self.amax_arena.numel(), 0), name="amax.zero")

        def conv(name, src0, src1=None, norm=None, **kw):
            """One convolution launch; norm = name of the GroupNorm fused on its input (its scale / shift launch is
            recorded here, right before the convolution, with the activation exponent the chosen tile wants)."""
            pc = net.convs[name]
            Bn, hin, win, _ = src0.shape
            hv, wv = (2 * hin, 2 * win) if kw.get("upsample") else (hin, win)
            s = kw.get("stride", 1)
            out = kw.pop("out", None)
            normed_later = kw.pop("normed_later", False)     # a GroupNorm will read this output
            if out is None:
                out = newa(Bn, (hv + s - 1) // s, (wv + s - 1) // s, pc.cout)
            gnp = None
            if norm is not None:
                c = src0.shape[-1] + (0 if src1 is None else src1.shape[-1])
                gnp = (new(B, c), new(B, c))
            if half:
                if norm is not None:
                    gn_launch(norm, src0, src1, gnp, 0)
                a = ops.conv16_args(pc, src0, src1, out, gn=gnp, **kw)
                if normed_later:
                    stats_of[out.data_ptr()] = self._hold(ops.conv16_stats_buffer(a, dev))
                self._add("conv_f16", a, name=name)
                return out
            a = ops.conv_args(pc, src0, src1, out, gn=gnp, **kw)
            if split and norm is not None:
                # GroupNorm-ed input: static exponent from the affine parameters, if the launch takes the split tile
                hw = src0.shape[1] * src0.shape[2]
                a.gn_exp = _abi.gn_act_exp(net.host[norm + ".weight"], net.host[norm + ".bias"], cfg["groups"], hw)
                a.flags |= _abi.CONV_GN_EXP
                if _abi.lib().cdx_conv_select_tile(ctypes.byref(a)) != _abi.TILE_SPLIT:
                    a.gn_exp, a.flags = 0, a.flags & ~_abi.CONV_GN_EXP
            elif split:
                # un-normalised input: per-image maxima from the producers, if the launch takes the split tile with them
                srcs = [src0] + ([src1] if src1 is not None else [])
                a.src_amax0 = amax_of(src0).data_ptr()
                a.src_amax1 = amax_of(src1).data_ptr() if src1 is not None else None
                if _abi.lib().cdx_conv_select_tile(ctypes.byref(a)) == _abi.TILE_SPLIT:
                    for t in srcs:
                        if t.data_ptr() in produced:
                            need_amax.add(t.data_ptr())
                        elif t.data_ptr() != getattr(self.ctx, "data_ptr", lambda: 0)():
                            # x_t | cond buffer, attention output: one pass over the tensor (ctx: once per decode, load_cond)
                            self._add("amax_f32", _abi.AmaxArgs(t.data_ptr(), t.shape[-1], Bn, t.shape[1] * t.shape[2], t.shape[-1],
                                                                amax_of(t).data_ptr()), name=name + ".amax_in")
                else:
                    a.src_amax0 = a.src_amax1 = None
            if norm is not None:
                gn_launch(norm, src0, src1, gnp, a.gn_exp)
                self.gn_of[norm] = (gnp[0], gnp[1], a.gn_exp)
            if normed_later and net.fuse_gn_stats:      # (asked AFTER the range fields are set: they decide the tile, the tile the slots)
                stats_of[out.data_ptr()] = self._hold(ops.conv_stats_buffer(a, dev))
            produced[out.data_ptr()] = a
            self.outs[name] = out
            self._add("conv_f32", a, name=name)
            return out

        def gn_launch(name, src0, src1, gnp, out_exp):
            """scale/shift of GroupNorm(cat[src0, src1]) times 2^out_exp.  When every source was written by one of our
            convs the sums come from that conv's epilogue (cdx_gn_finalize_f32: no extra pass over the activation)."""
            sc, sh = gnp
            gamma, beta = net.dev[name + ".weight"], net.dev[name + ".bias"]
            st0 = stats_of.get(src0.data_ptr())
            st1 = None if src1 is None else stats_of.get(src1.data_ptr())
            if (net.fuse_gn_stats or half) and st0 is not None and (src1 is None or st1 is not None):
                hw = src0.shape[1] * src0.shape[2]
                self._add("gn_finalize_f32", ops.gn_finalize_args(st0, st1, hw, gamma, beta, cfg["groups"], sc, sh, out_exp=out_exp), name=name)
            else:
                assert not half, "fp16 tensors always carry fused GroupNorm sums"
                self._add("gn_stats_f32", ops.gn_stats_args(src0, src1, gamma, beta, cfg["groups"], sc, sh, out_exp=out_exp), ws=True, name=name)

        def res(blk, x, skip=None):
            n = blk.name
            h1 = conv(n + ".conv1", x, skip, norm=n + ".norm1", silu=True, normed_later=True,
                      temb=self.tproj, temb_off=net.tproj_off[n], temb_ld=self.tproj.shape[-1])
            if blk.cin != blk.cout:
                r = conv(n + ".skip", x, skip)
            else:
                assert skip is None
                r = x
            return conv(n + ".conv2", h1, norm=n + ".norm2", silu=True, residual=r, normed_later=True)

        def attn(blk, x):
            n = blk.name
            Bn, hh, ww, c = x.shape
            qkv = conv(n + ".qkv", x, norm=n + ".norm")
            o = newa(Bn, hh, ww, c)
            self._add(ops.ATTN_OP[adt], ops.attn_args(qkv, qkv, qkv, o, batch=Bn, nq=hh * ww, nk=hh * ww,
                                                 heads=c // cfg["head_dim"], head_dim=cfg["head_dim"],
                                                 q_ld=3 * c, k_ld=3 * c, v_ld=3 * c, out_ld=c, k_off=c, v_off=2 * c), name=n + ".attn")
            return conv(n + ".proj", o, residual=x, normed_later=True)

        def xattn(blk, x):
            n = blk.name
            Bn, hh, ww, c = x.shape
            q = conv(n + ".q", x, norm=n + ".norm")
            kv = conv(n + ".kv", self.ctx)                       # [B, lh, lw, 2c]
            L = self.ctx.shape[1] * self.ctx.shape[2]
            o = newa(Bn, hh, ww, c)
            self._add(ops.ATTN_OP[adt], ops.attn_args(q, kv, kv, o, batch=Bn, nq=hh * ww, nk=L,
                                                 heads=c // cfg["head_dim"], head_dim=cfg["head_dim"],
                                                 q_ld=c, k_ld=2 * c, v_ld=2 * c, out_ld=c, v_off=c), name=n + ".attn")
            return conv(n + ".proj", o, residual=x, normed_later=True)

        def run_block(blk, h, skip=None):
            if blk.kind == "res":
                return res(blk, h, skip)
            if blk.kind == "attn":
                return attn(blk, h)
            if blk.kind == "xattn":
                return xattn(blk, h)
            if blk.kind == "down":
                return conv(blk.name, h, stride=2, normed_later=True)
            if blk.kind == "up":
                return conv(blk.name, h, upsample=True, normed_later=True)
            raise ValueError(blk.kind)

        if cross:
            L, D = (H // 16) ** 2, cfg["context_dim"]
            lw = 32 if L % 32 == 0 else L
            self.ctx = self._hold(torch.zeros(B, L // lw, lw, D, device=dev, dtype=adt))
            # the tokens do not change during a decode: their maxima live OUTSIDE the per-forward arena (load_cond fills them)
            self.ctx_amax = self._hold(ops.amax_buffer(B, dev))
            amax_slot[self.ctx.data_ptr()] = self.ctx_amax

        h = conv("conv_in", self.xin, normed_later=True)
        skips = [h]
        for blk in g.down:
            h = run_block(blk, h)
            if blk.push_skip:
                skips.append(h)
        for blk in g.mid:
            h = run_block(blk, h)
        for blk in g.up:
            h = run_block(blk, h, skips.pop() if blk.kind == "res" else None)
        assert not skips
        conv("out.conv", h, norm="out.norm", silu=True, out=self.eps, out_ld=self.eps.shape[-1])
        for k in need_amax:      # producers of the tensors an un-normalised split launch reads leave their maxima
            produced[k].amax_out = amax_slot[k].data_ptr()
        assert len(amax_slot) <= self.amax_arena.shape[0]

        # one shared workspace (calls are serialised on one stream)
        if self._ws_need:
            self.ws = self._hold(torch.empty((self._ws_need + 7) // 8, dtype=torch.float64, device=dev))
            for i in self._ws_calls:
                fn, a, _, nbytes = self.calls[i]
                self.calls[i] = (fn, a, self.ws.data_ptr(), nbytes)
        self.activation_bytes = sum(t.numel() * t.element_size() for t in self._keep)

    def _hold(self, t):
        self._keep.append(t)
        return t

    def _add(self, op, args, ws=False, name=""):
        nbytes = _abi.workspace_bytes(op, args) if ws else 0
        if nbytes:
            self._ws_need = max(self._ws_need, nbytes)
            self._ws_calls.append(len(self.calls))
        self.calls.append((getattr(_abi.lib(), f"cdx_{op}"), args, None, nbytes))
        self.names.append(name or op)

    def capture(self):
        """Record the launch list into a NEW hipGraph (torch.cuda.CUDAGraph on a side stream) and return it; the caller
        (a Sampler with use_graph=True) owns it and replays it with ONE launch instead of ~140-230.  Every buffer
        address and argument is static (the timestep is read from self.t on the device), so one capture serves the
        whole sampling loop.  The plan itself keeps no graph: run() is always the eager launch list.  Worth it only
        when a forward is host-bound (small images / batch 1); at cfg2 the host already runs ahead of the GPU."""
        with torch.cuda.device(self.net.device):
            self.run()                       # warm-up outside capture (lazy module loads)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.run()
        return g

    def run(self, stream: int | None = None, debug: bool = False):
        """Enqueue one UNet forward (eagerly) on `stream` (default: the current stream of the net's device): reads
        self.xin / self.t (/ self.ctx), writes self.eps.  hipLaunchKernelGGL launches on the CURRENT device, so the
        net's device is made current for the duration.  debug=True: after every float32 convolution its output is
        checked by cdx_check_finite_f32 (one sync per layer) and the first layer that stores NaN / Inf raises."""
        dev = self.net.device
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
            byref = ctypes.byref
            for i, (fn, a, wp, wb) in enumerate(self.calls):
                rc = fn(byref(a), wp, wb, st)
                if rc:
                    _abi.check(rc, fn.__name__ + " [" + self.names[i] + "]")
                if debug and fn.__name__ == "cdx_conv_f32":
                    status = torch.zeros(1, dtype=torch.int32, device=dev)
                    chk = _abi.CheckFiniteArgs(a.out, a.out_ld, a.batch * a.hout * a.wout, a.cout, 0.0, status.data_ptr())
                    _abi.call("check_finite_f32", chk, None, 0, st)
                    if int(status.item()):
                        raise _abi.CdxError(f"UNet forward (debug): layer {self.names[i]} stored non-finite values")


class UNet:
    """unet = UNet(cfg_dict[, params]); eps = unet.forward(x, t, cond)."""

    def __init__(self, cfg: dict, params: dict | None = None, *, seed: int = 0, device="cuda",
                 fuse_gn_stats: bool = True, split: bool = True):
        _abi.lib()   # fail loudly now if the extension is missing
        self.fuse_gn_stats = fuse_gn_stats   # False: every GroupNorm re-reads its input (cdx_gn_stats_f32)
        self.split = split                   # float32 layers carry the fp16 hi|lo weight image (CDX_TILE_SPLIT); False: f32-MFMA kernels only
        dt = validate_unet_config(cfg)["dtype"]
        self.half = dt in ("fp16", "bf16")
        self.bf16 = dt == "bf16"
        self.adt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[dt]
        if not torch.cuda.is_available():
            raise RuntimeError("UNet (HIP backend) needs a GPU; there is no CPU fallback in the product path")
        self.cfg = validate_unet_config(cfg)
        self.graph = build_graph(self.cfg)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"UNet (HIP backend) needs a cuda device, got {device!r}; there is no CPU fallback in the product path")
        if self.device.index is None:        # pin the ordinal now: every later launch targets THIS device
            self.device = torch.device("cuda", torch.cuda.current_device())
        if params is None:
            params = init_params(self.cfg, seed)
        g = self.graph
        missing = set(g.param_shapes) - set(params)
        if missing:
            raise KeyError(f"params lack {sorted(missing)[:5]}...")
        P = {k: np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, np.float32)
             for k, v in params.items()}
        for k, shp in g.param_shapes.items():
            if tuple(P[k].shape) != tuple(shp):
                raise ValueError(f"param {k}: shape {P[k].shape} != {shp}")

        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)  # noqa: E731
        self.dev = {}      # small tensors used as-is: norms, linears
        self.host = {}
        self.convs = {}    # name -> PackedConv
        res_blocks = [b for b in g.down + g.mid + g.up if b.kind == "res"]
        for name in g.param_shapes:
            if name.endswith(".weight") and P[name].ndim == 4:
                base = name[:-7]
                w, bias = P[name], P[base + ".bias"]
                c0, c1 = w.shape[1], 0
                if base == "conv_in":
                    cpad = _pad4(w.shape[1]) if self.half else _pad8(w.shape[1])
                    w = np.pad(w, ((0, 0), (0, cpad - w.shape[1]), (0, 0), (0, 0)))
                    c0 = cpad
                for b in res_blocks:    # up-path ResBlocks read (x, skip) as two sources
                    if b.skip_ch and base in (b.name + ".conv1", b.name + ".skip"):
                        c0, c1 = b.cin - b.skip_ch, b.skip_ch
                is_up = any(b.kind == "up" and b.name == base for b in g.up)      # (only up-sampling layers need the phase images)
                self.convs[base] = (ops.PackedConv16(w, bias, c0, c1, self.device, bf16=self.bf16) if self.half else
                                    ops.PackedConv(w, bias, c0, c1, self.device, split=split, up=is_up))
            elif ".norm" in name or name.startswith("temb."):
                self.dev[name] = up(P[name])
                if ".norm" in name:
                    self.host[name] = P[name]        # host copies: the static activation exponents (cdx_gn_act_exp)
        for b in g.down + g.mid + g.up:
            if b.kind == "xattn":   # kv projection of the context tokens runs as a 1x1 convolution
                w = P[b.name + ".kv.weight"]
                self.convs[b.name + ".kv"] = (ops.PackedConv16 if self.half else ops.PackedConv)(
                    w[:, :, None, None], P[b.name + ".kv.bias"], w.shape[1], 0, self.device, **({"bf16": self.bf16} if self.half else {"split": split}))
        # all ResBlock temb projections as one [sum(cout), temb_dim] linear
        self.tproj_off, off = {}, 0
        for b in res_blocks:
            self.tproj_off[b.name] = off
            off += b.cout
        self.tproj_w = up(np.concatenate([P[b.name + ".temb.weight"] for b in res_blocks], 0))
        self.tproj_b = up(np.concatenate([P[b.name + ".temb.bias"] for b in res_blocks], 0))
        self.weight_bytes = sum(pc.w.numel() * 4 for pc in self.convs.values()) + self.tproj_w.numel() * 4
        self._plans = {}

    def plan(self, batch: int) -> _Plan:
        if batch not in self._plans:
            with torch.cuda.device(self.device):
                self._plans[batch] = _Plan(self, batch)
        return self._plans[batch]

    @torch.no_grad()
    def forward(self, x: torch.Tensor, t: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        """x [B,C,H,W], t [B] int, cond [B,Cc,hc,wc] (concat) or [B,L,D] (cross_attn) -> eps [B,C,H,W]."""
        cfg = self.cfg
        B = x.shape[0]
        p = self.plan(B)
        dev = self.device
        C = cfg["in_channels"]
        with torch.cuda.device(dev):
            p.xin[..., :C] = x.to(dev, torch.float32).permute(0, 2, 3, 1)
            p.t.copy_(t.to(dev, torch.int32))
            load_cond(p, cfg, cond)
            p.run()
            return p.eps[..., :cfg["out_channels"]].permute(0, 3, 1, 2).contiguous()

    __call__ = forward


def load_cond(p: _Plan, cfg: dict, cond: torch.Tensor) -> None:
    dev = p.net.device
    if cfg["cond_mode"] == "concat":
        c = cond.to(dev, torch.float32).contiguous()
        assert c.shape[1] == cfg["cond_channels"], c.shape
        ops.cond_embed(c, p.xin, cfg["in_channels"])
    else:
        p.ctx.view(p.batch, -1, cfg["context_dim"]).copy_(cond.to(dev, torch.float32))   # (casts to fp16 when the plan is)
        if p.ctx.dtype == torch.float32:
            p.ctx_amax.zero_()
            ops.amax(p.ctx, out=p.ctx_amax)
