import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.nn.utils.rnn import pack_sequence
from padertorch_amd.ops import packed_lstm
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, T, H = 32, 64, 600
lstm = torch.nn.LSTM(257, H, 1, bidirectional=True).to(dev)
xs = [torch.randn(T, 257, device=dev, requires_grad=True) for _ in range(B)]
for it in range(2):
    y = packed_lstm(lstm, pack_sequence(xs))
    y.data.sum().backward()
torch.cuda.synchronize()
