"""CPU oracle for the reverse-diffusion decode path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference snapshot (/root/reference) holds no source, tests,
golden vectors or weights (README.md is 0 bytes; .gitignore:1-27 is its only content),
so there is nothing to pin this restatement against.  It restates, with stock
`torch.nn.functional` primitives only (conv2d, group_norm, silu, softmax, linear,
interpolate) on the CPU, the math fixed by BASELINE.json north_star + SURVEY.md
Appendix A (Ho et al. 2020 DDPM UNet; Song et al. 2021 DDIM).  It also stands in for
"the reference's own PyTorch-CPU sampler" that north_star wants timed (cpu_baseline
kind "port").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product package never does.
"""
from .unet_ref import unet_forward_ref, timestep_embedding_ref
from .sampler_ref import sample_ref, make_schedule_ref, step_coefficients_ref
from .rng_ref import normal_ref, uniform_ref, stream_key_ref
from .tiling_ref import sample_tiled_ref, blend_ref, origins_ref
from .context_ref import context_forward_ref
from .bitstream_ref import (quantise_ref, build_freq_ref, rans_encode_ref, rans_decode_ref, encode_latent_ref,
                            decode_latent_ref)

from .metrics_ref import to_uint8_ref, psnr_ref, msssim_ref, read_ppm_ref, read_png_ref

__all__ = ["to_uint8_ref", "psnr_ref", "msssim_ref", "read_ppm_ref", "read_png_ref", "unet_forward_ref", "timestep_embedding_ref", "sample_ref", "make_schedule_ref",
           "step_coefficients_ref", "normal_ref", "uniform_ref", "stream_key_ref",
           "sample_tiled_ref", "blend_ref", "origins_ref", "context_forward_ref", "quantise_ref", "build_freq_ref", "rans_encode_ref", "rans_decode_ref",
           "encode_latent_ref", "decode_latent_ref"]
