"""Drop-in for ``LSTM/rnn_lstm.RNN`` (rnn_lstm.py:8-57): the GRU decoder with ``nn.LSTM``
(gate order i,f,g,o; state (h, c)); greedy decoding only, as in the reference."""
from .rnn import RNN as _RNN


class RNN(_RNN):
    cell = "lstm"

    def sentence_index(self, cnn_feature, return_logits=False):   # rnn_lstm.py:35 has no beam_size
        return super().sentence_index(cnn_feature, 0, return_logits)
