"""cnns_slfp_quantization_amd -- MI355X (gfx950) native SLFP<3,4> / SFP<3,3> quantized
conv2d forward path behind the operator API of happyxtt/CNNs_SLFP_quantization.

    csrc/            hand-written HIP kernels + the C ABI (include/slfp.h) -> libslfp_hip.so
    _lib.py          ctypes binding (no fallback: raises if the library is missing)
    sfp_quant.py     mirror of the reference's utils/sfp_quant.py
    conv2d_func.py   mirror of the reference's utils/conv2d_func.py  (conv2d_Q, conv2d_Q_bias, linear_Q)
    activation_func.py  mirror of utils/activation_func.py (outside the hot path)
    layer_specs.py   Conv2d_Q layer tables of the reference nets (shapes + calibration scales)
    sharding.py      batch-axis sharding + one-time weight broadcast (torch.distributed / RCCL)
    fusion.py        eval-BN + ReLU folded into the conv epilogues
    calibration.py   device-side max|.| statistics for Ka / Kw
"""
__version__ = "0.1.0"
