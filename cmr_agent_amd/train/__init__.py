"""Update paths: flat parameter / gradient bucket, explicit HIP forward-backward of CMRAgent (agent_update)
and of MultiHeadModel (geo_update, on the reverse-mode tape of tape.py) in train() mode, fused Adam, one gradient all-reduce per
optimizer step."""
from .flatbucket import FlatBucket  # noqa: F401
from .agent_update import AgentUpdate  # noqa: F401
from .geo_update import GeoUpdate  # noqa: F401
