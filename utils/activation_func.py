"""`utils.activation_func` of the reference -> cnns_slfp_quantization_amd.activation_func."""
from cnns_slfp_quantization_amd.activation_func import *  # noqa: F401,F403
from cnns_slfp_quantization_amd.activation_func import __all__  # noqa: F401
