"""Import-path parity with the reference's model/generator/multiband_hifigan.py."""
from .hifigan import MultiBandHiFiGANGenerator  # noqa: F401
