import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import unext2_ref as R
from viscy_amd.unext2 import UNeXt2
torch.set_num_threads(32)
kw = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True, head_expansion_ratio=4, decoder_conv_blocks=2)
names = ["head.conv.1.weight", "head.conv.0.conv.weight", "decoder.decoder_stages.2.conv.blocks.1.mlp.fc2.weight", "decoder.decoder_stages.2.conv.blocks.1.mlp.fc1.weight", "encoder_stages.stages_2.blocks.4.mlp.fc2.weight", "stem.conv.weight"]
for S in (256, 512, 1024):
    o = R.UNeXt2(**kw); R.randomize_(o, seed=17); o.eval()
    x = torch.randn((1, 1, 5, S, S), generator=torch.Generator().manual_seed(4096))
    y = o(x)
    c = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)) / y.numel() ** 0.5
    (y * c).sum().backward()
    ref = {k: p.grad.clone().double().flatten() for k, p in o.named_parameters()}
    m = UNeXt2(**kw); m.load_state_dict(o.state_dict(), strict=True); m = m.cuda()
    m.compute_dtype, m.grad_mode = torch.float32, "flat"
    eng = m.engine(); eng.flat_grad.zero_()
    yy = m(x.cuda())
    print(S, "forward rel err", ((yy.cpu() - y.detach()).abs().max() / y.detach().abs().max()).item())
    (yy * c.cuda()).sum().backward()
    for k, p in m.named_parameters():
        if k in names:
            g = eng.g(p).double().flatten().cpu(); r = ref[k]
            print(f"  {k:60s} scale {(g @ r / (r @ r)).item():.6f} rel {((g - r).norm() / r.norm()).item():.2e} 1-cos {1 - torch.nn.functional.cosine_similarity(g, r, dim=0).item():.2e}")
    del m, eng, yy
    torch.cuda.empty_cache()
