from .base import NuScenesConfiguration  # noqa: F401  (reference module name: config/NuScenesConfig.py)
