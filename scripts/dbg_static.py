import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_ragged_graph as T
import padertorch_amd as pt
DEV='cuda:0'
src, lens, frames = T._batch(104, 64, 6400, 32, 120)
a = pt.ops.pit_features(src['y'], src['s'], lens)
b = pt.ops.pit_features(src['y'], src['s'], src['num_samples'], num_frames_dev=src['slots'].frames)
for k in ('Y_abs', 'X_abs', 'cos_phase_difference'):
    pa, pb = a[k].padded, b[k].padded
    print(k, pa.shape, pb.shape, 'host-path bad', int((~torch.isfinite(pa)).sum()), 'dev-path bad', int((~torch.isfinite(pb)).sum()))
    bad = (~torch.isfinite(pb)).reshape(pb.shape[0], pb.shape[1], -1).any(-1)
    bb, tt = torch.nonzero(bad, as_tuple=True)
    print('   at', list(zip(bb.tolist(), tt.tolist()))[:10], [(lens[i], frames[i]) for i in set(bb.tolist())])
    Tm = pa.shape[1]
    print('   equal on common frames', torch.equal(pa, pb[:, :Tm]), 'rest zero', float(torch.nan_to_num(pb[:, Tm:]).abs().sum()))
y = src['y']
print('input finite', bool(torch.isfinite(y).all()), bool(torch.isfinite(src['s']).all()))
# vary: host path with padded T
