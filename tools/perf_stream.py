import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N in [(131072, 384), (32768, 768), (8192, 1536), (2048, 3072), (131072, 896), (131072*4, 896)]:
    a = torch.randn(M, N, device="cuda").bfloat16(); b = torch.randn(M, N, device="cuda").bfloat16(); c = torch.empty_like(a)
    us = timeit(lambda: torch.add(a, b, out=c)); print(f"torch.add bf16 {M}x{N}: {us:8.1f} us {3*M*N*2/us/1e3:8.1f} GB/s")
    us = timeit(lambda: c.copy_(a)); print(f"torch.copy bf16 {M}x{N}: {us:8.1f} us {2*M*N*2/us/1e3:8.1f} GB/s")
    us = timeit(lambda: torch.nn.functional.gelu(a)); print(f"torch.gelu bf16 {M}x{N}: {us:8.1f} us {2*M*N*2/us/1e3:8.1f} GB/s")
    x = a.view(-1, 96) if N == 384 else a
    us = timeit(lambda: torch.nn.functional.layer_norm(x, (x.shape[1],))); print(f"torch.layer_norm bf16 {x.shape}: {us:8.1f} us {2*M*N*2/us/1e3:8.1f} GB/s")
