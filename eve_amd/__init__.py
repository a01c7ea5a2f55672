"""eve_amd: MI355X-native (gfx950) implementation of the EVE hot path -- the EyeNet encoder and the
RefineNet point-of-gaze refiner of swook/EVE -- as torch.nn.Module drop-ins over hand-written HIP
kernels (libeve_hip.so, C ABI in include/eve_hip.h)."""
from .config import HotPathConfig, get_config, reset_standalone_config  # noqa: F401
from .eye_net import EyeNet  # noqa: F401

__all__ = ['EyeNet', 'RefineNet', 'EVE', 'get_config', 'HotPathConfig', 'reset_standalone_config']


def __getattr__(name):
    if name == 'RefineNet':
        from .refine_net import RefineNet
        return RefineNet
    if name == 'EVE':
        from .eve import EVE
        return EVE
    raise AttributeError(name)
