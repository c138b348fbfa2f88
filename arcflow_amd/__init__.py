"""arcflow_amd -- MI355X-native (gfx950) engine for the ArcFlow 2-NFE text-to-image hot path.

Drop-in surface of the reference (pnotp/ArcFlow, lakonlab/pipelines): ``ArcFluxPipeline``,
``ArcQwenImagePipeline``, ``load_arcflow_adapter()``; the denoiser forward and the analytic ArcFlow
integrator run as hand-written HIP kernels behind the C ABI in include/arcflow_hip.h.
"""
__version__ = '0.1.0'

from .engine import ArcFlowModelOutput, MMDiTEngine  # noqa: F401
from .schedule import FlowMatchEulerDiscreteScheduler, retrieve_raw_timesteps  # noqa: F401


def __getattr__(name):
    if name in ('ArcFluxPipeline', 'ArcQwenImagePipeline'):
        from . import pipelines
        return getattr(pipelines, name)
    raise AttributeError(name)
