"""ctypes binding of libcdx.so (include/cdx.h).  The ONLY module that touches the C ABI.

There is no CPU fallback: if the shared library is missing or an entry point is absent the
import of the HIP backend fails loudly (RuntimeError), as the product path must.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CDX_TUNE=1 selects the tuning build (make EXPERIMENTS=1: the same library + timing-ablation instantiations, used by
# tools/conv_bench.py only); the product always loads libcdx.so.
LIB_PATH = os.path.join(_HERE, "libcdx_tune.so" if os.environ.get("CDX_TUNE") == "1" else "libcdx.so")

ABI_VERSION = 5
CONV_UPSAMPLE2X, CONV_GN, CONV_SILU, CONV_BF16, CONV_GN_EXP = 1, 2, 4, 8, 16
LINEAR_SILU_IN = 1
CONV_KC = 32
AMAX_WORDS = 16          # words per image of the amax arrays (cdx.h CDX_AMAX_WORDS)

_f = C.c_void_p   # device / host pointers are passed as raw addresses
_i = C.c_int32


class ConvArgs(C.Structure):
    _fields_ = [("src0", _f), ("src1", _f), ("c0", _i), ("c1", _i),
                ("batch", _i), ("hin", _i), ("win", _i), ("hout", _i), ("wout", _i),
                ("cout", _i), ("ksize", _i), ("stride", _i), ("flags", _i),
                ("wpacked", _f), ("bias", _f), ("gn_scale", _f), ("gn_shift", _f),
                ("temb", _f), ("temb_ld", _i), ("residual", _f), ("out", _f), ("out_ld", _i), ("wpacked_wino", _f), ("stats_out", _f),
                ("wpacked_split", _f), ("wsplit_unscale", C.c_float), ("gn_exp", _i),
                ("src_amax0", _f), ("src_amax1", _f), ("wpacked_split_up", _f), ("wsplit_up_unscale", C.c_float * 4),
                ("stats_slots", _i), ("amax_out", _f)]


class ConvF16Args(C.Structure):
    _fields_ = [("src0", _f), ("src1", _f), ("c0", _i), ("c1", _i), ("src_is_f32", _i),
                ("batch", _i), ("hin", _i), ("win", _i), ("hout", _i), ("wout", _i),
                ("cout", _i), ("ksize", _i), ("stride", _i), ("flags", _i),
                ("wpacked", _f), ("bias", _f), ("gn_scale", _f), ("gn_shift", _f),
                ("temb", _f), ("temb_ld", _i), ("residual", _f), ("out", _f), ("out_is_f32", _i), ("out_ld", _i),
                ("stats_out", _f)]


class GnStatsArgs(C.Structure):
    _fields_ = [("src0", _f), ("src1", _f), ("c0", _i), ("c1", _i), ("batch", _i), ("hw", _i),
                ("groups", _i), ("eps", C.c_float), ("gamma", _f), ("beta", _f),
                ("scale", _f), ("shift", _f), ("mean", _f), ("rstd", _f), ("out_exp", _i)]


class GnFinalizeArgs(C.Structure):
    _fields_ = [("part0", _f), ("slots0", _i), ("c0", _i), ("part1", _f), ("slots1", _i), ("c1", _i),
                ("batch", _i), ("hw", _i), ("groups", _i), ("eps", C.c_float), ("gamma", _f), ("beta", _f),
                ("scale", _f), ("shift", _f), ("mean", _f), ("rstd", _f), ("out_exp", _i)]


class AttnArgs(C.Structure):
    _fields_ = [("q", _f), ("q_ld", _i), ("k", _f), ("k_ld", _i), ("v", _f), ("v_ld", _i),
                ("batch", _i), ("nq", _i), ("nk", _i), ("heads", _i), ("head_dim", _i),
                ("scale", C.c_float), ("out", _f), ("out_ld", _i)]


class LinearArgs(C.Structure):
    _fields_ = [("x", _f), ("x_ld", _i), ("w", _f), ("bias", _f), ("m", _i), ("n", _i), ("k", _i),
                ("flags", _i), ("out", _f), ("out_ld", _i)]


class TimestepEmbeddingArgs(C.Structure):
    _fields_ = [("t", _f), ("batch", _i), ("dim", _i), ("out", _f)]


class DiffusionUpdateArgs(C.Structure):
    _fields_ = [("x", _f), ("x_ld", _i), ("eps", _f), ("eps_ld", _i),
                ("batch", _i), ("hw", _i), ("channels", _i),
                ("ca", C.c_float), ("cb", C.c_float), ("cx", C.c_float), ("c0", C.c_float),
                ("ce", C.c_float), ("sigma", C.c_float), ("clip_x0", _i),
                ("seed", C.c_uint64), ("first_image", C.c_int64), ("noise_stream", _i)]


class GaussFillArgs(C.Structure):
    _fields_ = [("x", _f), ("x_ld", _i), ("batch", _i), ("hw", _i), ("channels", _i),
                ("seed", C.c_uint64), ("first_image", C.c_int64), ("noise_stream", _i)]


class CondEmbedArgs(C.Structure):
    _fields_ = [("cond", _f), ("cc", _i), ("hc", _i), ("wc", _i), ("x", _f), ("x_ld", _i),
                ("c_off", _i), ("batch", _i), ("h", _i), ("w", _i)]


class ExportImageArgs(C.Structure):
    _fields_ = [("x", _f), ("x_ld", _i), ("batch", _i), ("hw", _i), ("channels", _i),
                ("lo", C.c_float), ("hi", C.c_float), ("out", _f)]


class RansDecodeArgs(C.Structure):
    _fields_ = [("words", _f), ("stream_off", _f), ("stream_len", _f), ("freq", _f), ("nstreams", _i), ("nsym", _i),
                ("alphabet", _i), ("prob_bits", _i), ("qmax", _i), ("step", C.c_float), ("out", _f), ("symbols", _f), ("status", _f)]


class AmaxArgs(C.Structure):
    _fields_ = [("x", _f), ("x_ld", _i), ("batch", _i), ("n", _i), ("channels", _i), ("out", _f)]


class FillU32Args(C.Structure):
    _fields_ = [("x", _f), ("n", C.c_int64), ("value", C.c_uint32)]


class CheckFiniteArgs(C.Structure):
    _fields_ = [("x", _f), ("x_ld", _i), ("rows", C.c_int64), ("channels", _i), ("limit", C.c_float), ("status", _f)]


class ExportU8Args(C.Structure):
    _fields_ = [("x", _f), ("x_ld", _i), ("batch", _i), ("hw", _i), ("channels", _i), ("lo", C.c_float), ("hi", C.c_float), ("out", _f)]


class PsnrArgs(C.Structure):
    _fields_ = [("a", _f), ("b", _f), ("batch", _i), ("n", C.c_int64), ("range", C.c_float), ("out", _f)]


class MsssimArgs(C.Structure):
    _fields_ = [("x", _f), ("y", _f), ("batch", _i), ("channels", _i), ("h", _i), ("w", _i), ("range", C.c_float),
                ("out", _f), ("per_scale", _f)]


class TileBlendArgs(C.Structure):
    _fields_ = [("tiles", _f), ("batch", _i), ("channels", _i), ("tile", _i), ("ny", _i), ("nx", _i),
                ("y0", _f), ("x0", _f), ("h", _i), ("w", _i), ("out", _f)]


# op name -> args struct; every op has cdx_<op>(args*, ws, ws_bytes, stream) and cdx_<op>_workspace(args*)
OPS = {
    "conv_f32": ConvArgs,
    "conv_f16": ConvF16Args,
    "attn_f16": AttnArgs,
    "gn_stats_f32": GnStatsArgs,
    "gn_finalize_f32": GnFinalizeArgs,
    "attn_f32": AttnArgs,
    "attn_bf16": AttnArgs,
    "linear_f32": LinearArgs,
    "timestep_embedding_f32": TimestepEmbeddingArgs,
    "diffusion_update_f32": DiffusionUpdateArgs,
    "gauss_fill_f32": GaussFillArgs,
    "cond_embed_f32": CondEmbedArgs,
    "export_image_f32": ExportImageArgs,
    "tile_blend_f32": TileBlendArgs,
    "rans_decode_i16": RansDecodeArgs,
    "amax_f32": AmaxArgs,
    "fill_u32": FillU32Args,
    "check_finite_f32": CheckFiniteArgs,
    "export_u8": ExportU8Args,
    "psnr_f32": PsnrArgs,
    "msssim_f32": MsssimArgs,
}

# every exported symbol include/cdx.h declares (checked by tests/test_abi.py without a GPU)
TILE_SPLIT = 11
TILE_NAMES = {0: "128x128", 1: "128x64", 2: "128x32", 3: "64x128", 4: "64x64", 5: "S32x32", 6: "S64x32", 7: "wino128x128", 8: "small256x4", 9: "small256x4valu", 10: "cin8_128x128", 11: "split128x128", 12: "smallgemm512x3"}

EXPORTS = (["cdx_abi_version", "cdx_strerror", "cdx_launch_count",
            "cdx_conv_packed_floats", "cdx_conv_pack_weights_f32", "cdx_conv_select_tile", "cdx_conv_f32_tile",
            "cdx_conv_stats_slots", "cdx_conv_wino_packed_floats", "cdx_conv_pack_weights_wino_f32",
            "cdx_conv_f16_stats_slots", "cdx_conv_f16_packed_halves", "cdx_conv_pack_weights_f16",
            "cdx_conv_split_packed_halves", "cdx_conv_pack_weights_split_f16", "cdx_conv_split_up_packed_halves", "cdx_conv_pack_weights_split_up_f16", "cdx_conv_pack_weights_bf16", "cdx_gn_act_exp"]
           + [f"cdx_{op}" for op in OPS] + [f"cdx_{op}_workspace" for op in OPS])

_lib = None


def lib() -> C.CDLL:
    """Load libcdx.so once; raise if it is not built or its ABI version differs."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C conditional-diffusion-model-for-compression_amd/csrc`). "
                           "There is no CPU fallback for the HIP path.")
    L = C.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise RuntimeError(f"libcdx.so lacks symbols {missing}")
    L.cdx_abi_version.restype = C.c_int
    if L.cdx_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libcdx.so ABI {L.cdx_abi_version()} != binding {ABI_VERSION}")
    L.cdx_strerror.restype = C.c_char_p
    L.cdx_strerror.argtypes = [C.c_int]
    L.cdx_launch_count.restype = C.c_uint64
    L.cdx_conv_packed_floats.restype = C.c_size_t
    L.cdx_conv_packed_floats.argtypes = [_i, _i, _i, _i]
    L.cdx_conv_pack_weights_f32.restype = C.c_int
    L.cdx_conv_pack_weights_f32.argtypes = [_f, _i, _i, _i, _i, _f]
    L.cdx_conv_wino_packed_floats.restype = C.c_size_t
    L.cdx_conv_wino_packed_floats.argtypes = [_i, _i, _i]
    L.cdx_conv_pack_weights_wino_f32.restype = C.c_int
    L.cdx_conv_pack_weights_wino_f32.argtypes = [_f, _i, _i, _i, _f]
    L.cdx_conv_f16_stats_slots.restype = C.c_int32
    L.cdx_conv_f16_stats_slots.argtypes = [C.POINTER(ConvF16Args)]
    L.cdx_conv_f16_packed_halves.restype = C.c_size_t
    L.cdx_conv_f16_packed_halves.argtypes = [_i, _i, _i, _i]
    L.cdx_conv_pack_weights_f16.restype = C.c_int
    L.cdx_conv_pack_weights_f16.argtypes = [_f, _i, _i, _i, _i, _f]
    L.cdx_conv_pack_weights_bf16.restype = C.c_int
    L.cdx_conv_pack_weights_bf16.argtypes = [_f, _i, _i, _i, _i, _f]
    L.cdx_conv_split_packed_halves.restype = C.c_size_t
    L.cdx_conv_split_packed_halves.argtypes = [_i, _i, _i, _i]
    L.cdx_conv_pack_weights_split_f16.restype = C.c_int
    L.cdx_conv_pack_weights_split_f16.argtypes = [_f, _i, _i, _i, _i, _f, C.POINTER(C.c_float)]
    L.cdx_conv_split_up_packed_halves.restype = C.c_size_t
    L.cdx_conv_split_up_packed_halves.argtypes = [_i, _i, _i]
    L.cdx_conv_pack_weights_split_up_f16.restype = C.c_int
    L.cdx_conv_pack_weights_split_up_f16.argtypes = [_f, _i, _i, _i, _f, C.POINTER(C.c_float)]
    L.cdx_conv_select_tile.restype = C.c_int
    L.cdx_conv_select_tile.argtypes = [C.POINTER(ConvArgs)]
    L.cdx_conv_stats_slots.restype = C.c_int32
    L.cdx_conv_stats_slots.argtypes = [C.POINTER(ConvArgs)]
    L.cdx_conv_f32_tile.restype = C.c_int
    L.cdx_conv_f32_tile.argtypes = [C.POINTER(ConvArgs), _i, C.c_void_p, C.c_size_t, C.c_void_p]
    L.cdx_gn_act_exp.restype = C.c_int32
    L.cdx_gn_act_exp.argtypes = [_f, _f, _i, _i, _i]
    for op, st in OPS.items():
        fn = getattr(L, f"cdx_{op}")
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(st), C.c_void_p, C.c_size_t, C.c_void_p]
        ws = getattr(L, f"cdx_{op}_workspace")
        ws.restype = C.c_size_t
        ws.argtypes = [C.POINTER(st)]
    _lib = L
    return L


class CdxError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise CdxError(f"{what}: {lib().cdx_strerror(rc).decode()} (status {rc})")


def call(op: str, args: C.Structure, ws_ptr: int, ws_bytes: int, stream: int) -> None:
    check(getattr(lib(), f"cdx_{op}")(C.byref(args), ws_ptr, ws_bytes, stream), f"cdx_{op}")


def workspace_bytes(op: str, args: C.Structure) -> int:
    return int(getattr(lib(), f"cdx_{op}_workspace")(C.byref(args)))


def gn_act_exp(gamma, beta, groups: int, hw: int) -> int:
    """cdx_gn_act_exp on host arrays: the static activation exponent of a GroupNorm-ed split-tile input (cdx.h)."""
    import numpy as np
    g = np.ascontiguousarray(gamma, dtype=np.float32)
    b = np.ascontiguousarray(beta, dtype=np.float32)
    return int(lib().cdx_gn_act_exp(g.ctypes.data, b.ctypes.data, g.size, groups, hw))


def pack_conv_weights(w_oihw, c0: int, c1: int):
    """numpy OIHW float32 -> numpy packed float32 image (host)."""
    import numpy as np
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    cout, cin, k, _ = w.shape
    assert cin == c0 + c1
    n = int(lib().cdx_conv_packed_floats(c0, c1, cout, k))
    if n == 0:
        raise CdxError("cdx_conv_packed_floats: bad arguments")
    out = np.empty(n, np.float32)
    check(lib().cdx_conv_pack_weights_f32(w.ctypes.data, c0, c1, cout, k, out.ctypes.data), "cdx_conv_pack_weights_f32")
    return out


def pack_conv_weights_wino(w_oihw, c0: int, c1: int):
    """numpy OIHW 3x3 float32 -> Winograd-transformed, fragment-ordered float32 image (host)."""
    import numpy as np
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    cout, cin, k, _ = w.shape
    assert k == 3 and cin == c0 + c1
    n = int(lib().cdx_conv_wino_packed_floats(c0, c1, cout))
    out = np.empty(n, np.float32)
    check(lib().cdx_conv_pack_weights_wino_f32(w.ctypes.data, c0, c1, cout, out.ctypes.data), "cdx_conv_pack_weights_wino_f32")
    return out


def pack_conv_weights_f16(w_oihw, c0: int, c1: int):
    """numpy OIHW float32 -> fp16 fragment image as numpy float16 (host)."""
    import numpy as np
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    cout, cin, k, _ = w.shape
    assert cin == c0 + c1
    n = int(lib().cdx_conv_f16_packed_halves(c0, c1, cout, k))
    if n == 0:
        raise CdxError("cdx_conv_f16_packed_halves: bad arguments")
    out = np.empty(n, np.float16)
    check(lib().cdx_conv_pack_weights_f16(w.ctypes.data, c0, c1, cout, k, out.ctypes.data), "cdx_conv_pack_weights_f16")
    return out


def pack_conv_weights_split(w_oihw, c0: int, c1: int):
    """numpy OIHW float32 -> (fp16 hi|lo fragment image as numpy float16, unscale = 2^-s) for CDX_TILE_SPLIT (host)."""
    import numpy as np
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    cout, cin, k, _ = w.shape
    assert cin == c0 + c1
    n = int(lib().cdx_conv_split_packed_halves(c0, c1, cout, k))
    if n == 0:
        raise CdxError("cdx_conv_split_packed_halves: bad arguments")
    out = np.empty(n, np.float16)
    un = C.c_float(0.0)
    check(lib().cdx_conv_pack_weights_split_f16(w.ctypes.data, c0, c1, cout, k, out.ctypes.data, C.byref(un)),
          "cdx_conv_pack_weights_split_f16")
    return out, float(un.value)


def pack_conv_weights_split_up(w_oihw, c0: int, c1: int):
    """numpy OIHW 3x3 float32 -> (four fp16 hi|lo phase images back to back as numpy float16, [4] unscales): the four 2x2
    convolutions that equal the 3x3 convolution after nearest-2x upsampling (cdx.h wpacked_split_up) (host)."""
    import numpy as np
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    cout, cin, k, _ = w.shape
    assert cin == c0 + c1 and k == 3
    n = int(lib().cdx_conv_split_up_packed_halves(c0, c1, cout))
    if n == 0:
        raise CdxError("cdx_conv_split_up_packed_halves: bad arguments")
    out = np.empty(n, np.float16)
    un = (C.c_float * 4)()
    check(lib().cdx_conv_pack_weights_split_up_f16(w.ctypes.data, c0, c1, cout, out.ctypes.data, un), "cdx_conv_pack_weights_split_up_f16")
    return out, [float(v) for v in un]


def pack_conv_weights_bf16(w_oihw, c0: int, c1: int):
    """numpy OIHW float32 -> bfloat16 fragment image as numpy uint16 bit patterns (host)."""
    import numpy as np
    w = np.ascontiguousarray(w_oihw, dtype=np.float32)
    cout, cin, k, _ = w.shape
    assert cin == c0 + c1
    n = int(lib().cdx_conv_f16_packed_halves(c0, c1, cout, k))
    if n == 0:
        raise CdxError("cdx_conv_f16_packed_halves: bad arguments")
    out = np.empty(n, np.uint16)
    check(lib().cdx_conv_pack_weights_bf16(w.ctypes.data, c0, c1, cout, k, out.ctypes.data), "cdx_conv_pack_weights_bf16")
    return out
