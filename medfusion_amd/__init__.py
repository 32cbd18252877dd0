"""medfusion_amd -- MI355X-native latent-diffusion sampling path for Medfusion-style models.

Drop-in for the reference's sampling API (`DiffusionPipeline.sample/denoise/forward`, `UNet.forward`,
`VAE.encode/decode`, `GaussianNoiseScheduler`) running on hand-written HIP kernels for gfx950 through a
C-ABI (include/medfusion_hip.h).  No CPU fallback: the library must load and tensors must be on the GPU.
"""
from .lib import load as load_library  # noqa: F401
from .noise import HostNoise, NoiseSource, PhiloxDeviceNoise, torch_cpu_noise  # noqa: F401
from .pipeline import DiffusionPipeline, EMAModel  # noqa: F401
from .scheduler import BasicNoiseScheduler, GaussianNoiseScheduler  # noqa: F401
from .unet import LabelEmbedder, LearnedSinusoidalPosEmb, SinusoidalPosEmb, TimeEmbbeding, UNet  # noqa: F401
from .vae import VAE, DiagonalGaussianDistribution  # noqa: F401

__version__ = "0.1.0"
