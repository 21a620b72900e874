"""``StatefulLSTM`` (reference: ``padertorch/modules/recurrent.py:5-47``) on the HIP BLSTM recurrence.

Same constructor; the parameters live in a ``torch.nn.LSTM`` (same ``state_dict`` keys:
``lstm.weight_ih_l0`` ...).  The time loop runs in ``csrc/lstm.hip`` (``ops.lstm.packed_lstm``); ``(h_n, c_n)`` of a call is
carried into the next one like the reference does - WITH their graph (``modules/recurrent.py:42-43`` stores what ``torch.nn.LSTM``
returns): a loss on a later chunk reaches the earlier chunks' inputs and the parameters through the carried states (the persistent
kernels' state gradients, ``ptmi_lstm_backward_persistent_states``), and - as with the reference - a second ``backward`` through a chunk
whose graph has been freed raises torch's error: truncated backpropagation is the caller's ``states = tuple(s.detach() ...)``.
"""
import torch
from torch.nn.utils.rnn import PackedSequence

from .._lib import leaving_native_path
from ..ops.lstm import packed_lstm, unsupported_reason


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
            why = unsupported_reason(self.lstm, x.data)
            if why is not None:
                # the host (Trainer.test_run / inference on the CPU, as the reference allows) or an LSTM the kernels do not cover:
                # torch's own LSTM, said out loud on a GPU
                if x.data.is_cuda:
                    leaving_native_path(f'StatefulLSTM({self.lstm.input_size}, {self.hidden_size})', why)
                out, states = self.lstm(x, self.states)
            elif self.save_states or self.states is not None:
                out, states = packed_lstm(self.lstm, x, hx=self.states, return_state=True)
            else:
                # no state comes in, none is kept: the plain path (hand-off planes for the next layer, dgates^T planes for the weight
                # gradients, no state-gradient work - ADVICE r4)
                return packed_lstm(self.lstm, x)
            self.states = states
            if not self.save_states:
                del self.states
            return out
        assert x.dim() == 3, x.shape
        xt = x.transpose(0, 1) if self.batch_first else x            # [T, B, F]
        T, B = xt.shape[:2]
        packed = PackedSequence(xt.reshape(T * B, -1), torch.full((T,), B, dtype=torch.int64))
        if self.save_states or self.states is not None:
            out, states = packed_lstm(self.lstm, packed, hx=self.states, return_state=True)
            self.states = states
            if not self.save_states:
                del self.states
        else:
            out = packed_lstm(self.lstm, packed)
        h = out.data.reshape(T, B, -1)
        return h.transpose(0, 1) if self.batch_first else h
