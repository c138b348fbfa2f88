from .distill import ArcFlowDistiller, DistillConfig  # noqa: F401
from .reducer import GradReducer, host_or_device, init_distributed  # noqa: F401
from . import checkpoint  # noqa: F401
