from .base import KittiConfiguration  # noqa: F401  (reference module name: config/KittiConfig.py)
