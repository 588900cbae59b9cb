"""Generator classes, importable like the reference's ``model.generator``
(/root/reference/model/generator/__init__.py:1-4)."""
from .hifigan import HiFiGANGenerator, MultiBandHiFiGANGenerator  # noqa: F401
from .melgan import BasisMelGANGenerator, MelGANGenerator  # noqa: F401
from .pqmf import PQMF  # noqa: F401
