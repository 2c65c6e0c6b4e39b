"""Dataset-side pre-processing on the device."""
from .frame import preprocess_frame, random_transform  # noqa: F401
