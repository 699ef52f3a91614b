"""Static description of the UNet (U0 of SURVEY.md section 8a): the ordered list of
blocks with their channel counts and resolutions, derived from the config dict.

Architecture (canonical DDPM UNet, Ho et al. 2020):
  conv_in -> down levels [ResBlock x nrb (+Attn)(+CrossAttn)] + Downsample
          -> mid ResBlock, Attn, (CrossAttn), ResBlock
          -> up levels [ResBlock(cat skip) x (nrb+1) (+Attn)(+CrossAttn)] + Upsample
          -> GN, SiLU, conv3x3.
The reference snapshot holds no model code to cite (README.md: 0 bytes).
"""
from __future__ import annotations

from dataclasses import dataclass, field

from .config import validate_unet_config


@dataclass
class Block:
    kind: str            # "res" | "attn" | "xattn" | "down" | "up"
    name: str            # parameter prefix
    cin: int             # input channels (for "res" on the up path: x channels + skip channels)
    cout: int
    res: int             # spatial resolution of the block's INPUT
    skip_ch: int = 0     # up-path ResBlock: channels popped from the skip stack (cin = x_ch + skip_ch)
    push_skip: bool = False   # push this block-group's output on the skip stack after it


@dataclass
class Graph:
    cfg: dict
    temb_dim: int
    cin_total: int                    # conv_in input channels (in_channels + cond_channels)
    down: list = field(default_factory=list)
    mid: list = field(default_factory=list)
    up: list = field(default_factory=list)
    param_shapes: dict = field(default_factory=dict)   # name -> shape (torch layout)


def build_graph(cfg: dict) -> Graph:
    cfg = validate_unet_config(cfg)
    ch = cfg["base_channels"]
    nrb = cfg["num_res_blocks"]
    mults = cfg["channel_mult"]
    cross = cfg["cond_mode"] == "cross_attn"
    g = Graph(cfg=cfg, temb_dim=4 * ch, cin_total=cfg["in_channels"] + cfg["cond_channels"])
    P = g.param_shapes

    def conv(name, ci, co, k):
        P[f"{name}.weight"] = (co, ci, k, k)
        P[f"{name}.bias"] = (co,)

    def lin(name, ci, co):
        P[f"{name}.weight"] = (co, ci)
        P[f"{name}.bias"] = (co,)

    def norm(name, c):
        P[f"{name}.weight"] = (c,)
        P[f"{name}.bias"] = (c,)

    def res(name, ci, co):
        norm(f"{name}.norm1", ci)
        conv(f"{name}.conv1", ci, co, 3)
        lin(f"{name}.temb", g.temb_dim, co)
        norm(f"{name}.norm2", co)
        conv(f"{name}.conv2", co, co, 3)
        if ci != co:
            conv(f"{name}.skip", ci, co, 1)

    def attn(name, c):
        norm(f"{name}.norm", c)
        conv(f"{name}.qkv", c, 3 * c, 1)
        conv(f"{name}.proj", c, c, 1)

    def xattn(name, c):
        norm(f"{name}.norm", c)
        conv(f"{name}.q", c, c, 1)
        lin(f"{name}.kv", cfg["context_dim"], 2 * c)
        conv(f"{name}.proj", c, c, 1)

    lin("temb.0", ch, g.temb_dim)
    lin("temb.2", g.temb_dim, g.temb_dim)
    conv("conv_in", g.cin_total, ch, 3)

    def extras(blocks, prefix, c, r):
        if r in cfg["attn_resolutions"]:
            attn(f"{prefix}.attn", c)
            blocks.append(Block("attn", f"{prefix}.attn", c, c, r))
        if cross and r in cfg["cross_attn_resolutions"]:
            xattn(f"{prefix}.xattn", c)
            blocks.append(Block("xattn", f"{prefix}.xattn", c, c, r))

    skips = [ch]
    cur, r = ch, cfg["image_size"]
    for i, m in enumerate(mults):
        co = ch * m
        for j in range(nrb):
            name = f"down.{i}.{j}"
            res(f"{name}.res", cur, co)
            g.down.append(Block("res", f"{name}.res", cur, co, r))
            cur = co
            extras(g.down, name, cur, r)
            g.down[-1].push_skip = True
            skips.append(cur)
        if i != len(mults) - 1:
            conv(f"down.{i}.ds", cur, cur, 3)
            g.down.append(Block("down", f"down.{i}.ds", cur, cur, r, push_skip=True))
            skips.append(cur)
            r //= 2

    res("mid.0.res", cur, cur)
    g.mid.append(Block("res", "mid.0.res", cur, cur, r))
    attn("mid.1.attn", cur)
    g.mid.append(Block("attn", "mid.1.attn", cur, cur, r))
    if cross:
        xattn("mid.1.xattn", cur)
        g.mid.append(Block("xattn", "mid.1.xattn", cur, cur, r))
    res("mid.2.res", cur, cur)
    g.mid.append(Block("res", "mid.2.res", cur, cur, r))

    for i in reversed(range(len(mults))):
        co = ch * mults[i]
        for j in range(nrb + 1):
            name = f"up.{i}.{j}"
            sk = skips.pop()
            res(f"{name}.res", cur + sk, co)
            g.up.append(Block("res", f"{name}.res", cur + sk, co, r, skip_ch=sk))
            cur = co
            extras(g.up, name, cur, r)
        if i != 0:
            conv(f"up.{i}.us", cur, cur, 3)
            g.up.append(Block("up", f"up.{i}.us", cur, cur, r))
            r *= 2
    assert not skips and cur == ch * mults[0]
    norm("out.norm", cur)
    conv("out.conv", cur, cfg["out_channels"], 3)
    return g
