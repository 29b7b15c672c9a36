"""show-tell on MI355X: HIP/CDNA4 kernels behind the reference's nn.Module surface.

    from showtell_amd.cnn import ResNet                 # cnn.py
    from showtell_amd.rnn import RNN                    # rnn.py
    from showtell_amd.rnn_lstm import RNN as RNN_LSTM   # LSTM/rnn_lstm.py
    from showtell_amd.cnn_attn import ResNet            # Attention/cnn_attn.py
    from showtell_amd import optim                      # SGD / Adam of main.py:96-100
"""
from . import _lib  # noqa: F401
from ._lib import ShowTellHipError  # noqa: F401

__all__ = ["ShowTellHipError"]
