"""`utils.conv2d_func` of the reference -> cnns_slfp_quantization_amd.conv2d_func (HIP path)."""
from cnns_slfp_quantization_amd.conv2d_func import *  # noqa: F401,F403
from cnns_slfp_quantization_amd.conv2d_func import __all__  # noqa: F401
