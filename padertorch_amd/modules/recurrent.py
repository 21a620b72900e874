"""``StatefulLSTM`` (reference: ``padertorch/modules/recurrent.py:5-47``) on the HIP BLSTM recurrence.

Same constructor; the parameters live in a ``torch.nn.LSTM`` (same ``state_dict`` keys:
``lstm.weight_ih_l0`` ...).  The time loop runs in ``csrc/lstm.hip`` (``ops.lstm.packed_lstm``); ``(h_n, c_n)`` of a call is
carried into the next one like the reference does.  Documented difference: the carried states are
constants of the next call (detached) - backpropagation does not reach across calls (the reference
would need ``retain_graph`` for that; streaming use detaches anyway).  ``ops.lstm.packed_lstm`` itself
does differentiate w.r.t. an initial state handed to it.
"""
import torch
from torch.nn.utils.rnn import PackedSequence

from ..ops.lstm import packed_lstm


class StatefulLSTM(torch.nn.Module):
    _states = None

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, bidirectional: bool = False,
                 dropout: float = 0., batch_first: bool = True, save_states: bool = True):
        super().__init__()
        self.lstm = torch.nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                                  bidirectional=bidirectional, dropout=dropout, batch_first=batch_first)
        self.hidden_size = hidden_size
        self.bidirectional = bidirectional
        self.num_layers = num_layers
        self.batch_first = batch_first
        self.save_states = save_states

    @property
    def states(self):
        return self._states

    @states.deleter
    def states(self):
        self._states = None

    @states.setter
    def states(self, states):
        self._states = states

    def forward(self, x):
        if isinstance(x, PackedSequence):           # torch.nn.LSTM takes either form (contrib/jensheit's MaskEstimator packs)
            out, states = packed_lstm(self.lstm, x, hx=self.states, return_state=True)
            self.states = tuple(s.detach() for s in states)
            if not self.save_states:
                del self.states
            return out
        assert x.dim() == 3, x.shape
        xt = x.transpose(0, 1) if self.batch_first else x            # [T, B, F]
        T, B = xt.shape[:2]
        packed = PackedSequence(xt.reshape(T * B, -1), torch.full((T,), B, dtype=torch.int64))
        if self.save_states or self.states is not None:
            out, states = packed_lstm(self.lstm, packed, hx=self.states, return_state=True)
            self.states = tuple(s.detach() for s in states)
            if not self.save_states:
                del self.states
        else:
            out = packed_lstm(self.lstm, packed)
        h = out.data.reshape(T, B, -1)
        return h.transpose(0, 1) if self.batch_first else h
