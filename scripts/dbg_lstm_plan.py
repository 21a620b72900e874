import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.nn.utils.rnn import pack_sequence
from padertorch_amd.ops import lstm as L
dev = torch.device('cuda:0')
torch.manual_seed(3)
I, H, lens = 33, 40, [17, 17, 12, 9, 9, 2, 1, 1, 1, 1]
dut = torch.nn.LSTM(I, H, 1, bidirectional=True).to(dev)
xs = [torch.randn(l, I, device=dev) for l in lens]
g = torch.randn(sum(lens), 2 * H, device=dev)
res = []
for it in range(3):
    for p in dut.parameters(): p.grad = None
    xd = [x.clone().requires_grad_(True) for x in xs]
    y = L.packed_lstm(dut, pack_sequence(xd))
    (y.data * g).sum().backward()
    torch.cuda.synchronize()
    res.append(([p.grad.clone() for p in dut.parameters()], y.data.detach().clone(), [x.grad.clone() for x in xd]))
    print('it', it, 'pool', {k[2][:3]: (v[0], len(v[1])) for k, v in L._POOL.items()})
for it in (1, 2):
    print('y diff', (res[it][1] - res[0][1]).abs().max().item())
    for (n, _), a, b in zip(dut.named_parameters(), res[it][0], res[0][0]):
        print(it, n, (a - b).abs().max().item(), a.abs().max().item())
    print('dx', max((a - b).abs().max().item() for a, b in zip(res[it][2], res[0][2])))
