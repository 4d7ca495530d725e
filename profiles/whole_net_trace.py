"""Tuning helper: run only bench.whole_net (fused path) so that a rocprofv3 --kernel-trace --stats pass
shows where the whole-net time goes (this library's kernels vs the torch ops around them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnns_slfp_quantization_amd import layer_specs
specs = layer_specs.conv_layers("mobilenetv1_imagenet224")
print(bench.whole_net(specs, "mobilenetv1_imagenet224", 256, torch.device("cuda:0"), 10))
