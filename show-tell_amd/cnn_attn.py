"""Drop-in for ``Attention/cnn_attn.ResNet`` (cnn_attn.py:9-52): same constructor and
attributes as ``cnn.ResNet`` but the backbone stops before the average pool and
``forward`` returns the detached feature map as ``(B, 2048, 49)`` fp32."""
import torch

from .cnn import ResNet as _ResNet


class ResNet(_ResNet):
    _AVGPOOL = False   # children()[:-2], cnn_attn.py:34

    def forward(self, x):  # Extract CNN features with spatial resolution preserved (cnn_attn.py:44-52)
        _, ncp = self._bb.forward(x, self.training, False, True)
        return ncp
