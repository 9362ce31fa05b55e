import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import loss_ref, unext2_ref
nt = int(sys.argv[1]); torch.set_num_threads(nt)
kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True)
m = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=0)
x = torch.randn(2, 1, 5, 256, 256); t = torch.rand(2, 2, 5, 256, 256)
for i in range(2):
    t0 = time.perf_counter(); y = m(x); t1 = time.perf_counter()
    l = loss_ref.mixed_loss(y, t, 0.5, 0, 0.5); t2 = time.perf_counter()
    l.backward(); t3 = time.perf_counter()
    print(f"threads={nt} iter{i}: fwd {t1-t0:.2f}s loss {t2-t1:.2f}s bwd {t3-t2:.2f}s", flush=True)
