"""Public surface of the package (re-exported by the `cdx` import shim)."""
from .config import unet_config, validate_unet_config, named_config, UNET_DEFAULTS, SCHEDULE_DEFAULTS
from .schedule import make_schedule, timestep_subsequence, step_coefficients, StepCoef
from .graph import build_graph
from .params import init_params, synthetic_batch
from . import rng

__all__ = ["unet_config", "validate_unet_config", "named_config", "UNET_DEFAULTS", "SCHEDULE_DEFAULTS",
           "make_schedule", "timestep_subsequence", "step_coefficients", "StepCoef",
           "build_graph", "init_params", "synthetic_batch", "rng"]
from .unet import UNet
from .sampler import Sampler, sample
from . import ops, _abi, shard, params, tiling
from .tiling import tile_plan, tile_origins
from .shard import shard_range, decode_shard, timed_region

__all__ += ["UNet", "Sampler", "sample", "ops", "_abi", "shard", "params", "shard_range", "decode_shard", "timed_region", "tiling", "tile_plan", "tile_origins"]
from . import context
from .context import ContextNet, context_config, init_context_params, synthetic_latent, decode_latent

__all__ += ["context", "ContextNet", "context_config", "init_context_params", "synthetic_latent", "decode_latent"]
from . import bitstream
from .bitstream import LatentDecoder, parse_latent_stream, decode_bitstreams

__all__ += ["bitstream", "LatentDecoder", "parse_latent_stream", "decode_bitstreams"]
from . import image_io
from .image_io import to_uint8, write_image, write_images, encode_png, encode_ppm, psnr, ms_ssim

__all__ += ["image_io", "to_uint8", "write_image", "write_images", "encode_png", "encode_ppm", "psnr", "ms_ssim"]
