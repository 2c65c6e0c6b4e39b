"""Dataset-side pre-processing on the device."""
from .frame import preprocess_frame, random_transform  # noqa: F401
from .loader import FrameDataset, FrameLoader, read_calib  # noqa: F401,E402
from .sampling import hip_fps, hip_nearest  # noqa: F401,E402
