"""GPU-only hunt: new model each iteration (fresh allocations), idle gaps, compare all gradients with the first run."""
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import unext2_ref  # noqa: E402
from viscy_amd.unext2 import UNeXt2  # noqa: E402

kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True)
torch.manual_seed(0)
ref = unext2_ref.randomize_(unext2_ref.UNeXt2(**kw), seed=7).eval()
sd = ref.state_dict()
x = torch.randn(2, 1, 5, 128, 128)
dout = torch.randn(2, 2, 5, 128, 128)
gold = None
import viscy_amd.ops as O  # noqa: E402

REC = {}
_b1, _b2 = O.head_out_bwd1, O.head_out_bwd2


def b1(U, ssum, ssq, w2, alpha, dout, S1, S2, dalpha, *a, **k):
    REC["S_before_b1"] = torch.stack([S1, S2]).clone()
    REC["dalpha_before"] = dalpha.clone()
    REC["S1_ptr"] = torch.tensor([S1.data_ptr() % (1 << 21), dalpha.data_ptr() % (1 << 21)], dtype=torch.float64)
    r = _b1(U, ssum, ssq, w2, alpha, dout, S1, S2, dalpha, *a, **k)
    REC["dalpha_after"] = dalpha.clone()
    REC["S_after_b1"] = torch.stack([S1, S2]).clone()
    REC["stats_b1"] = torch.stack([ssum, ssq]).clone()
    REC["act_cs"] = r[0].float().view(-1, 32).sum(0)
    REC["dv_cs"] = r[1].float().view(-1, 8).sum(0)
    return r


def b2(U, ssum, ssq, w2, alpha, dv, S1, S2, *a, **k):
    REC["S_before_b2"] = torch.stack([S1, S2]).clone()
    REC["stats_b2"] = torch.stack([ssum, ssq]).clone()
    REC["U_cs"] = U.float().view(-1, 32).sum(0)
    REC["w2"] = w2.clone()
    r = _b2(U, ssum, ssq, w2, alpha, dv, S1, S2, *a, **k)
    REC["dU_cs"] = r.float().view(-1, 32).abs().sum(0)
    return r


O.head_out_bwd1, O.head_out_bwd2 = b1, b2
gold_rec = None
mode = sys.argv[2] if len(sys.argv) > 2 else "flat"
nbad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    m = UNeXt2(**kw)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    m.compute_dtype, m.grad_mode = torch.float32, mode
    if it % 2:
        time.sleep(1.5)
        junk = torch.randn(1 << 26, device="cuda")  # dirty a lot of memory
        del junk
        torch.cuda.empty_cache() if it % 4 == 1 else None
    if mode == "flat":
        m.engine().flat_grad.zero_()
    out = m(x.cuda())
    out.backward(dout.cuda())
    g = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    o = out.detach().cpu()
    rec = {k: v.detach().cpu().clone() for k, v in REC.items()}
    if gold is None:
        gold, gold_out, gold_rec = g, o, rec
        continue
    bad = [(n, ((g[n] - gold[n]).abs().max() / gold[n].abs().max().clamp_min(1e-12)).item()) for n in g if n != "head.conv.0.conv.bias"]
    bad = [(n, e) for n, e in bad if e > 1e-4]
    print(it, "fwd dev", ((o - gold_out).abs().max() / gold_out.abs().max()).item(), "n bad params", len(bad), flush=True)
    if bad:
        nbad += 1
        print("      S_before_b1 absmax", rec["S_before_b1"].abs().max().item(), "dalpha", rec["dalpha_before"].item(), rec["dalpha_after"].item(),
              "gold dalpha", gold_rec["dalpha_after"].item(), "ptrs", rec["S1_ptr"].tolist(), "gold ptrs", gold_rec["S1_ptr"].tolist())
        for k in rec:
            if k == "S1_ptr":
                continue
            d = (rec[k] - gold_rec[k]).abs() / gold_rec[k].abs().max().clamp_min(1e-20)
            if d.max() > 1e-5:
                if k == "S_after_b1":
                    print("      S1[1][25] gold", gold_rec[k][0, 1, 25].item(), "bad", rec[k][0, 1, 25].item(), "S2 gold", gold_rec[k][1, 1, 25].item(), "bad", rec[k][1, 1, 25].item())
                print("   REC", k, tuple(rec[k].shape), "max dev", d.max().item(), "at", (d > 1e-5).nonzero().tolist()[:6])
        if nbad <= 2:
            for n, e in bad[-14:]:
                print("   ", n, f"{e:.2e}")
            n = "head.conv.0.conv.weight"
            d = (g[n] - gold[n]).abs().view(32, 8, 27)
            thr = 1e-3 * gold[n].abs().max()
            print("   bad per cmid", (d > thr).sum((1, 2)).tolist())
            print("   bad per c3", (d > thr).sum((0, 2)).tolist())
            print("   bad per tap", (d > thr).sum((0, 1)).tolist())
print("misses", nbad)
