"""fastvocoder_amd -- MI355X (gfx950) native generator-inference engine with the
FastVocoder generator surface (HiFi-GAN, Multiband-HiFi-GAN + PQMF, MelGAN,
Basis-MelGAN).  See DESIGN.md; the HIP kernels live in csrc/ behind the C ABI
declared in include/fastvocoder_hip.h."""
from . import _native  # noqa: F401
from .generator import (BasisMelGANGenerator, HiFiGANGenerator, MelGANGenerator,  # noqa: F401
                        MultiBandHiFiGANGenerator, PQMF)

__all__ = ["HiFiGANGenerator", "MultiBandHiFiGANGenerator", "MelGANGenerator",
           "BasisMelGANGenerator", "PQMF"]
