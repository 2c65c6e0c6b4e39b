from .base import KittiConfiguration, NuScenesConfiguration  # noqa: F401
