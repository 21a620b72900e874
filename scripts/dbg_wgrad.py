"""Where does the linear1.weight gradient error of the row-slot cases (70 examples in 32 slots, 3 x BLSTM-600: 2.8e-4 of the largest
entry against the fp64 oracle; 100 examples in 64 slots, 3 x BLSTM-64: 3.3e-4, `profiles/r5_grad_errors_vs_fp64.txt`) come from: the
weight-gradient GEMM on its actual inputs, or the inputs?  `python scripts/dbg_wgrad.py [B slots units layers]`.
Result (profiles/r6_relu_tie.txt): the GEMMs are exact to 2e-7 on their own inputs; ONE ReLU whose pre-activation is 1e-10 takes the
other branch than the fp64 oracle's - with the HIP path's own ReLU pattern the whole fp64 chain agrees to 2e-7."""
import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import padertorch_amd as pt
from padertorch_amd.ops import gemm as G, lstm as L, context as C
from padertorch_amd.contrib.examples.source_separation.pit.model import PermutationInvariantTrainingModel
import test_gpu_fullsize as F
DEV = 'cuda:0'
B, slots, units, layers = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (70, 32, 600, 2)
rng = np.random.RandomState(B + slots)
n = 4400
lens = sorted((int(x) for x in rng.randint(900, n + 1, B)), reverse=True); lens[0] = n
torch.manual_seed(B)
model = PermutationInvariantTrainingModel(units=units, recurrent_layers=layers, K=2 + B % 2).to(DEV).train()
model.row_slots = slots
s = F._waveforms(B, 2 + B % 2, n, lens, B + 1).to(DEV)
feats = pt.ops.pit_features(s.sum(1), s, lens)
for p in model.parameters():
    p.grad = torch.zeros_like(p)
C.attach(model, C.OpContext(defer_wgrad=True))
L.warm_side_stream(torch.device(DEV))
rec = []
real = G.pack_t
def spy(x, amax=None):
    rec.append(x)
    return real(x, amax)
G.pack_t = spy
masks = model(feats)
loss = model.review(feats, masks)['losses']['pit_ips_loss']
loss.backward(); L.sync_deferred(); torch.cuda.synchronize()
G.pack_t = real
print('pack_t calls', [tuple(r.shape) for r in rec])
# linear2 then linear1 backward come first: (g2, x2), (g1, x1)
for name, (g, x) in (('linear2.weight', rec[1:3]), ('linear1.weight', rec[4:6])):
    got = dict(model.named_parameters())[name].grad.double()
    want = g.double().t() @ x.double()
    sc = float(want.abs().max())
    print(name, 'GEMM error on its own inputs / max', float((got - want).abs().max()) / sc, ' max', sc,
          ' sum|ab|/max', float((g.double().abs().t() @ x.double().abs()).max()) / sc,
          ' g absmax', float(g.abs().max()), 'g nonzero rows', int((g.abs().sum(1) > 0).sum()), 'of', g.shape[0],
          ' x absmax', float(x.abs().max()))
    # dynamic range of g rows
    gr = g.abs().amax(1); gr = gr[gr > 0]
    print('    row maxima of g: min %.3e median %.3e max %.3e' % (float(gr.min()), float(gr.median()), float(gr.max())))
# --- is it the ReLU? recompute linear1's pre-activation in fp64 from the HIP path's own input h, compare the sign pattern
g2, a1, g1, h = rec[1], rec[2], rec[4], rec[5]
W1, b1, W2 = model.linear1.weight.double(), model.linear1.bias.double(), model.linear2.weight.double()
z1 = h.double() @ W1.t() + b1
mask_hip, mask_64 = a1 > 0, z1 > 0
flips = mask_hip != mask_64
print('relu sign flips (HIP mask vs fp64 pre-activation from the SAME h):', int(flips.sum()), 'of', flips.numel(),
      ' |z1| at flips max %.3e' % float(z1.abs()[flips].max() if flips.any() else 0), ' entries with |z1| < 1e-6:', int((z1.abs() < 1e-6).sum()))
dA1 = g2.double() @ W2
got = model.linear1.weight.grad.double()
for tag, mask in (('HIP relu mask', mask_hip), ('fp64 relu mask', mask_64)):
    want = (dA1 * mask).t() @ h.double()
    print('dW1 vs fp64 chain from HIP inputs with', tag, float((got - want).abs().max()) / float(want.abs().max()))
