"""Import-path parity with the reference's model/generator/basis_melgan.py."""
from .melgan import BasisMelGANGenerator  # noqa: F401
