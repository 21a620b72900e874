"""Do HIP events recorded INSIDE a captured graph (event-record nodes: torch.cuda.Event(external=True)) time the kernels of a replay?"""
import torch
dev = 'cuda:0'
x = torch.randn(1 << 26, device=dev)
y = torch.empty_like(x)
for ext in (True, False):
    try:
        evs = [torch.cuda.Event(enable_timing=True, external=ext) for _ in range(3)]
    except TypeError as e:
        print('external kwarg unsupported', e)
        continue
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            evs[0].record()
            y.copy_(x)
            evs[1].record()
            y.mul_(2.)
            y.add_(1.)
            evs[2].record()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print('external', ext, 'elapsed ms', evs[0].elapsed_time(evs[1]), evs[1].elapsed_time(evs[2]))
    except Exception as e:
        print('external', ext, 'failed:', type(e).__name__, str(e)[:200])
