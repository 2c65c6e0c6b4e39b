"""Agent update path: flat parameter / gradient bucket, explicit HIP forward-backward of CMRAgent in
train() mode, fused Adam, one gradient all-reduce per optimizer step."""
from .flatbucket import FlatBucket  # noqa: F401
from .agent_update import AgentUpdate  # noqa: F401
