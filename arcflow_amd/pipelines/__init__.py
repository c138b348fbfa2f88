from .arcflow_loader import ArcFlowLoaderMixin  # noqa: F401
from .arcflux_pipeline import ArcFluxPipeline  # noqa: F401
from .arcqwen_pipeline import ArcQwenImagePipeline  # noqa: F401
