from .distill import ArcFlowDistiller, DistillConfig  # noqa: F401
from .reducer import GradReducer  # noqa: F401
from . import checkpoint  # noqa: F401
