"""nn.Module API of the reference's `models` package (models/__init__.py:1-6) on HIP kernels."""
from .IMGPCEncoder import IMGPCEncoder  # noqa: F401
from .IMGPCEnDecoder import IMGPCEnDecoder  # noqa: F401
from .MultiHeadModel import MultiHeadModel  # noqa: F401
from .IterModel import IterModel  # noqa: F401
from .CMRAgent import CMRAgent  # noqa: F401
