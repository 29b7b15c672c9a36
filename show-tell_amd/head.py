"""Trainable encoder head: BatchNorm1d(Linear(x)) of the reference (cnn.py:37-38,49) on HIP.

``x`` is the detached backbone output (cnn.py:47), so only the four head tensors receive
gradients; they are accumulated straight into ``param.grad`` by st_linear_bn1d_backward.
"""
import ctypes as C

import torch

from . import ops
from ._lib import ST_BF16, ST_F32, check, lib
from .rnn import _cp, _stream, grad_buffer, working_copy


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, linear, bn, train, dtype):
        dev = x.device
        B, F = x.shape
        E = linear.out_features
        dtc = ST_F32 if dtype == torch.float32 else ST_BF16
        xd = x if x.dtype == dtype else ops.cast(x.contiguous(), dtype)
        w = working_copy(linear.weight, dtype)
        z = torch.empty(B, E, device=dev, dtype=torch.float32)
        y = torch.empty(B, E, device=dev, dtype=torch.float32)
        mean = torch.empty(E, device=dev, dtype=torch.float32)
        rstd = torch.empty(E, device=dev, dtype=torch.float32)
        check(lib().st_linear_bn1d_forward(_cp(xd), _cp(w), _cp(linear.bias.data), _cp(bn.weight.data), _cp(bn.bias.data),
                                           _cp(bn.running_mean), _cp(bn.running_var), B, F, E, dtc, int(train),
                                           float(bn.momentum), float(bn.eps), _cp(z), _cp(mean), _cp(rstd), None, _cp(y),
                                           _stream()), "st_linear_bn1d_forward")
        if train:
            bn.num_batches_tracked.add_(1)
        ctx.saved = (xd, z, mean, rstd, linear, bn, train, dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xd, z, mean, rstd, linear, bn, train, dtype = ctx.saved
        B, F = xd.shape
        E = linear.out_features
        dtc = ST_F32 if dtype == torch.float32 else ST_BF16
        nbytes = lib().st_head_workspace_bytes(B, F, E, dtc)
        ws = torch.empty(nbytes, device=dy.device, dtype=torch.uint8)
        check(lib().st_linear_bn1d_backward(_cp(dy.contiguous().float()), _cp(z), _cp(xd), _cp(bn.weight.data), _cp(mean), _cp(rstd),
                                            B, F, E, dtc, int(train), _cp(grad_buffer(linear.weight)), _cp(grad_buffer(linear.bias)),
                                            _cp(grad_buffer(bn.weight)), _cp(grad_buffer(bn.bias)), _cp(ws), nbytes, _stream()),
              "st_linear_bn1d_backward")
        return None, None, None, None, None, None


def linear_bn1d(x, linear, bn, train, dtype):
    """y = bn(linear(x)); `linear.weight` is passed through autograd only as an anchor so that the
    result requires grad whenever the head is trainable."""
    return _HeadFn.apply(x, linear.weight, linear, bn, train, dtype)
