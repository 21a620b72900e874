"""Does an RCCL all-reduce capture into a hipGraph ("or captured, if RCCL permits - test it")?  One-rank group on one GPU."""
import os, socket, torch, torch.distributed as dist
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
dev = torch.device('cuda:0')
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 20, device=dev)
dist.all_reduce(x)                      # communicator made outside the capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        x.mul_(2.)
        dist.all_reduce(x)
        x.add_(1.)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print('captured and replayed: x[0] =', float(x[0]), '(expected 15.0)')
    side = torch.cuda.Stream()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, capture_error_mode='thread_local'):
        x.mul_(2.)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            w = dist.all_reduce(x, async_op=True)
        y = x.new_ones(8).sum()             # main-stream work beside the collective
        w.wait()
        x.add_(1.)
    g2.replay(); torch.cuda.synchronize()
    print('async on a side stream inside a capture: x[0] =', float(x[0]), '(expected 31.0)')
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:300])
dist.destroy_process_group()
