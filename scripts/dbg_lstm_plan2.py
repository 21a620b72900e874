import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from torch.nn.utils.rnn import pack_sequence
from padertorch_amd.ops import lstm as L
DEV = 'cuda:0'
torch.manual_seed(3)
I, H, lens = 33, 40, [17, 17, 12, 9, 9, 2, 1, 1, 1, 1]
ref = torch.nn.LSTM(I, H, 2, bidirectional=True)
dut = torch.nn.LSTM(I, H, 2, bidirectional=True)
dut.load_state_dict(ref.state_dict())
dut = dut.to(DEV)
xs = [torch.randn(l, I) for l in lens]
yr, _ = ref(pack_sequence([x.clone().requires_grad_(True) for x in xs]))
L._POOL.clear()
outs = []
for it in range(3):
    xd = [x.clone().to(DEV).requires_grad_(True) for x in xs]
    y = L.packed_lstm(dut, pack_sequence(xd))
    print('fwd', it, (y.data.detach().cpu() - yr.data.detach()).abs().max().item(), [(v[0], [w.busy for w in v[1]]) for v in L._POOL.values()])
    outs.append((y, xd))
g = torch.randn(yr.data.shape)
(yr.data * g).sum().backward()
for i, (y, xd) in enumerate(outs):
    for p in dut.parameters():
        p.grad = None
    (y.data * g.to(DEV)).sum().backward(retain_graph=True)
    torch.cuda.synchronize()
    for (n, pd), pr in zip(dut.named_parameters(), ref.parameters()):
        print('bwd', i, n, (pd.grad.cpu() - pr.grad).abs().max().item())
print('---- second experiment: checksums')
L._POOL.clear()
outs = []
sums = []
for it in range(3):
    xd = [x.clone().to(DEV).requires_grad_(True) for x in xs]
    y = L.packed_lstm(dut, pack_sequence(xd))
    torch.cuda.synchronize()
    wss = [w for v in L._POOL.values() for w in v[1]]
    sums.append([(w.gates.double().sum().item(), w.c.double().sum().item(), w.hy.double().sum().item()) for w in wss])
    outs.append((y, xd))
wss = [w for v in L._POOL.values() for w in v[1]]
after = [(w.gates.double().sum().item(), w.c.double().sum().item(), w.hy.double().sum().item()) for w in wss]
for i, (a, b) in enumerate(zip(sums[-1], after)):
    print('ws', i, 'same' if a == b else ('CHANGED', a, b))
print('ws sums after each fwd:')
for srow in sums: print([round(t[0], 3) for t in srow])
# backward only the last one first
for p in dut.parameters(): p.grad = None
(outs[2][0].data * g.to(DEV)).sum().backward()
torch.cuda.synchronize()
for (n, pd), pr in zip(dut.named_parameters(), ref.parameters()):
    if 'weight_ih' in n: print('bwd(last first)', n, (pd.grad.cpu() - pr.grad).abs().max().item())
