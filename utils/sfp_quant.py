"""`utils.sfp_quant` of the reference -> cnns_slfp_quantization_amd.sfp_quant (HIP path)."""
from cnns_slfp_quantization_amd.sfp_quant import *  # noqa: F401,F403
from cnns_slfp_quantization_amd.sfp_quant import __all__  # noqa: F401
