"""nn.Module API of the reference's `models` package (models/__init__.py:1-6) on HIP kernels.
IterModel (dead code in the reference) is not provided."""
from .IMGPCEncoder import IMGPCEncoder  # noqa: F401
from .IMGPCEnDecoder import IMGPCEnDecoder  # noqa: F401
from .MultiHeadModel import MultiHeadModel  # noqa: F401
from .CMRAgent import CMRAgent  # noqa: F401
