"""Mirror of the reference's utils/activation_func.py (STL, Swish, Sigmoid).

Out of the hot path: plain PyTorch modules that sit between the
quantized convolutions; they exist so that `from utils.activation_func import *` resolves
for the reference nets.  Semantics follow utils/activation_func.py:6-36.
"""
import torch
import torch.nn as nn

__all__ = ["STL", "Swish", "Sigmoid", "STLFunction"]


class _StlFn(torch.autograd.Function):
    """y = x for |x| <= 1, sign(x) * (ln|x| + 1) otherwise (utils/activation_func.py:10)."""

    @staticmethod
    def forward(ctx, x):
        mag = x.abs()
        return torch.where(mag <= 1, x, torch.sign(x) * (torch.log(mag) + 1))

    @staticmethod
    def backward(ctx, g):
        # the reference scales by 1/|g| where |g| > 1 (utils/activation_func.py:16): a sign clip
        mag = g.abs()
        return torch.where(mag <= 1, torch.ones_like(g), 1 / mag) * g


def STLFunction():
    return _StlFn.apply


class STL(nn.Module):
    def __init__(self):
        super().__init__()
        self.stl = STLFunction()

    def forward(self, x):
        return self.stl(x)


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class Sigmoid(nn.Module):
    def forward(self, x):
        return torch.sigmoid(x)
