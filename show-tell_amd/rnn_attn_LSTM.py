"""Drop-in for ``Attention/rnn_attn_LSTM.py``: the attention decoder with ``nn.LSTM`` and an extra
``init_c`` projection for the initial cell state (rnn_attn_LSTM.py:50,55,63)."""
from .rnn_attn import Attention_Net, RNN_Attn as _RNN_Attn  # noqa: F401


class RNN_Attn(_RNN_Attn):
    cell = "lstm"
