import torch, time
dev = torch.device('cuda:0')
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M, K, N = 8096, 1200, 4800
a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
t32 = bench(lambda: a @ b.t())
print(f'fp32 [{M}x{K}]x[{K}x{N}]: {t32:.1f} us  {2*M*K*N/t32/1e6:.1f} TF')
ab, bb = a.bfloat16(), b.bfloat16()
t16 = bench(lambda: ab @ bb.t())
print(f'bf16->bf16: {t16:.1f} us  {2*M*K*N/t16/1e6:.1f} TF')
try:
    t16f = bench(lambda: torch.mm(ab, bb.t(), out_dtype=torch.float32))
    print(f'bf16->fp32 out_dtype: {t16f:.1f} us  {2*M*K*N/t16f/1e6:.1f} TF')
    ok = True
except Exception as e:
    print('out_dtype unsupported:', repr(e)[:200]); ok = False
if ok:
    def split(x):
        hi = x.bfloat16(); lo = (x - hi.float()).bfloat16(); return hi, lo
    def mm3(a, b):   # a [M,K], b [N,K]
        ah, al = split(a); bh, bl = split(b)
        out = torch.mm(ah, bh.t(), out_dtype=torch.float32)
        out = out + torch.mm(ah, bl.t(), out_dtype=torch.float32)
        out = out + torch.mm(al, bh.t(), out_dtype=torch.float32)
        return out
    t3 = bench(lambda: mm3(a, b))
    ref = (a.double() @ b.double().t())
    e32 = ((a @ b.t()).double() - ref).abs().max().item() / ref.abs().max().item()
    e3 = (mm3(a, b).double() - ref).abs().max().item() / ref.abs().max().item()
    e1 = (torch.mm(ab, bb.t(), out_dtype=torch.float32).double() - ref).abs().max().item() / ref.abs().max().item()
    print(f'bf16x3 (incl. splits): {t3:.1f} us; rel err fp32 {e32:.2e}, bf16x3 {e3:.2e}, bf16 {e1:.2e}')
    # dW shape: [N x M] x [M x K]
    g = torch.randn(M, N, device=dev)
    tw32 = bench(lambda: g.t() @ a)
    gh = g.bfloat16()
    tw16 = bench(lambda: torch.mm(gh.t(), ab, out_dtype=torch.float32))
    print(f'dW fp32 {tw32:.1f} us; bf16->fp32 {tw16:.1f} us')
    tx32 = bench(lambda: g @ b)
    tx16 = bench(lambda: torch.mm(gh, bb, out_dtype=torch.float32))
    print(f'dX fp32 {tx32:.1f} us; bf16->fp32 {tx16:.1f} us')
