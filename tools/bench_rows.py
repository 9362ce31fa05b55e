"""Measurement of the widened SURVEY §8(f) rows and the (a9)/(a10)/(a14) boundary pieces — NOT the headline metric
(that is bench.py).  One JSON object per row on stdout; run on the GPU box:

    python tools/bench_rows.py > gpurun_out/rows.json

rows:
  predict_2048      config 4: (1,1,21,2048,2048) Z-sliding-window prediction, 17 forwards + blend, fp32
                    ("32-true", the reference's predict precision) and bf16, eager and hipGraph-captured windows
  fwd_2048_b8       UNeXt2 forward at the roofline target shape (B=8, Z=5, 2048x2048), bf16
  fcmae_pretrain    FCMAE masked pre-training step (mask_ratio 0.5, MaskedMSELoss, fused AdamW), B=256, vs the dense step
  dynaclr_train     ContrastiveModule step (convnext_tiny / convnextv2_tiny trunk, 2 ch x 15 slices, 256^2, NT-Xent, fused AdamW)
  augment_chain     normalise + affine + crop + contrast + scale + noise + smooth on (B,2,15,384,384)-class batches
"""

from __future__ import annotations

import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def row_predict():
    from viscy_amd.vsunet import VSUNet

    out = {}
    cfg = dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True)
    vs = VSUNet("UNeXt2", cfg).cuda().eval()
    vs.on_predict_start()
    x = torch.randn(1, 1, 21, 2048, 2048, device="cuda")
    for dt, name in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        vs.model.compute_dtype = dt
        for graph in (False, True):
            vs.predict_graph = graph
            vs._infer_step = None
            with torch.no_grad():
                s = timed(lambda: vs.predict_sliding_windows(x, out_channel=2), iters=2, warmup=1)
            out[f"{name}_{'graph' if graph else 'eager'}"] = {"s_per_fov": round(s, 4), "windows": 17,
                                                              "mvox_out_per_s": round(2 * 21 * 2048 * 2048 / s / 1e6, 1),
                                                              "fwd_tflops": round(17 * 1.444 / s, 1)}
    return {"row": "predict_2048", "shape": [1, 1, 21, 2048, 2048], **out}


def row_fwd2048():
    from viscy_amd.unext2 import UNeXt2

    m = UNeXt2(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True).cuda().eval()
    m.compute_dtype = torch.bfloat16
    x = torch.randn(8, 1, 5, 2048, 2048, device="cuda")
    with torch.no_grad():
        s = timed(lambda: m(x), iters=3, warmup=1)
    # 1.444 TFLOP and 4.83 GB ("two-pass floor") per 2048^2 sample forward
    return {"row": "fwd_2048_b8", "ms": round(s * 1e3, 2), "tflops": round(8 * 1.444 / s, 1), "frac_mfma_peak": round(8 * 1.444 / s / 2500, 4),
            "algorithmic_GBps": round(8 * 4.83 / s, 1), "frac_hbm_peak": round(8 * 4.83 / s / 8000, 4),
            "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1)}


def row_fcmae():
    from viscy_amd.losses import MaskedMSELoss, MixedLoss
    from viscy_amd.vsunet import FcmaeUNet, VSUNet

    B = 256
    kw = dict(in_channels=1, out_channels=1, encoder_blocks=[3, 3, 9, 3], dims=[96, 192, 384, 768], decoder_conv_blocks=2,
              in_stack_depth=5, pretraining=True)
    x = torch.randn(B, 1, 5, 256, 256, device="cuda")
    out = {"row": "fcmae_pretrain", "B": B}
    for ratio in (0.5, 0.0):
        torch.cuda.empty_cache()
        if ratio:
            vs = FcmaeUNet(fit_mask_ratio=ratio, model_config=kw, loss_function=MaskedMSELoss(), lr=2e-4).cuda()
        else:
            vs = FcmaeUNet(model_config={**kw, "pretraining": False}, loss_function=MixedLoss(0.0, 1.0, 0.0), lr=2e-4).cuda()
        vs.model.compute_dtype = torch.bfloat16
        opt = vs.configure_optimizers(t_total=100)

        def step():
            opt.zero_grad()
            loss = vs.training_step({"source": x, "target": x}, 99)
            loss.backward()
            opt.step()

        s = timed(step, iters=5, warmup=2)
        out["masked_0.5" if ratio else "dense"] = {"ms_per_step": round(s * 1e3, 1), "patches_per_s": round(B / s, 1)}
        if ratio:
            gs = vs.make_pretrain_step(opt)
            sg = timed(lambda: gs(x, x), iters=5, warmup=2)
            out["masked_0.5"].update(graph_ms_per_step=round(sg * 1e3, 1), graph_patches_per_s=round(B / sg, 1))
            del gs
        del vs, opt
    return out


def row_dynaclr():
    """DynaCLR training step (SURVEY §8 f3 / BASELINE config 5 shape): convnext_tiny trunk, 2 ch x 15 slices, 256x256 patches,
    NT-Xent(T = 0.07 ... here 0.2) on (anchor, positive) batches, fused AdamW, bf16, eager launches"""
    from viscy_amd.contrastive import ContrastiveEncoder, ContrastiveModule, NTXentLoss

    out = {"row": "dynaclr_train"}
    for backbone in ("convnext_tiny", "convnextv2_tiny"):
        for B in (64, 256):
            torch.cuda.empty_cache()
            enc = ContrastiveEncoder(backbone, in_channels=2, in_stack_depth=15, embedding_dim=768, projection_dim=128)
            mod = ContrastiveModule(enc, loss_function=NTXentLoss(temperature=0.2), lr=2e-4).cuda()
            enc.compute_dtype = torch.bfloat16
            a = torch.randn(B, 2, 15, 256, 256, device="cuda")
            batch = {"anchor": a, "positive": a + 0.1 * torch.randn_like(a)}
            opt = mod.configure_optimizers(t_total=100)
            mod.train()

            def step():
                opt.zero_grad()
                mod.training_step(batch, 0).backward()
                opt.step()

            s = timed(step, iters=5, warmup=2)
            gs = mod.make_train_step(opt)
            sg = timed(lambda: gs(batch["anchor"], batch["positive"]), iters=5, warmup=2)
            out[f"{backbone}_B{B}"] = {"ms_per_step": round(s * 1e3, 1), "pairs_per_s": round(B / s, 1), "images_per_s": round(2 * B / s, 1),
                                        "graph_ms_per_step": round(sg * 1e3, 1), "graph_images_per_s": round(2 * B / sg, 1)}
            del enc, mod, opt, batch, a, gs
    return out


def row_augment():
    from viscy_amd import transforms as T

    B = 32
    src = torch.rand(B, 1, 20, 600, 600, device="cuda") * 100
    tgt = torch.rand(B, 2, 20, 600, 600, device="cuda") * 100
    keys = ["source", "target"]
    chain = [
        T.BatchedRandAffined(keys=keys, prob=0.8, rotate_range=[3.14, 0.0, 0.0], scale_range=[[0.7, 1.3], [0.5, 1.5], [0.5, 1.5]],
                             shear_range=[0.0, 0.05, 0.05]),  # the published VSCyto3D fine-tuning recipe
        T.BatchedCenterSpatialCropd(keys=keys, roi_size=(15, 384, 384)),
        T.BatchedRandAdjustContrastd(keys=["source"], prob=0.5, gamma=(0.8, 1.2)),
        T.BatchedRandScaleIntensityd(keys=["source"], prob=0.5, factors=0.5),
        T.BatchedRandGaussianNoised(keys=["source"], prob=0.5, mean=0.0, std=0.3),
        T.BatchedRandGaussianSmoothd(keys=["source"], prob=0.5, sigma_x=(0.25, 0.75), sigma_y=(0.25, 0.75), sigma_z=(0.0, 0.0)),
    ]

    def run(ch=chain):
        b = {"source": src, "target": tgt}
        for t in ch:
            b = t(b)
        return b

    s_unfused = timed(run, iters=5, warmup=2)
    fused = T.fuse_affine_crop(chain)  # what HCSDataModule applies
    s = timed(lambda: run(fused), iters=5, warmup=2)
    vox_in = B * 3 * 20 * 600 * 600
    per = {}
    b = {"source": src, "target": tgt}
    for t in chain:  # per-transform split (each timed on the batch the previous one produced)
        t.prob = 1.0
        keep = dict(b)
        per[type(t).__name__] = round(timed(lambda: t(dict(keep)), iters=5, warmup=1) * 1e3, 2)
        b = t(dict(keep))
    return {"row": "augment_chain", "B": B, "in": [B, "1+2", 20, 600, 600], "out": [15, 384, 384], "ms": round(s * 1e3, 2),
            "samples_per_s": round(B / s, 1), "input_GBps": round(vox_in * 4 / s / 1e9, 1), "ms_unfused": round(s_unfused * 1e3, 2), "ms_each_prob1": per}


if __name__ == "__main__":
    want = sys.argv[1:] or ["augment", "fcmae", "dynaclr", "fwd2048", "predict"]
    rows = {"augment": row_augment, "fcmae": row_fcmae, "dynaclr": row_dynaclr, "fwd2048": row_fwd2048, "predict": row_predict}
    for w in want:
        torch.cuda.reset_peak_memory_stats()
        try:
            r = rows[w]()
        except Exception as e:  # one failing row must not hide the others
            r = {"row": w, "error": f"{type(e).__name__}: {e}"}
        print(json.dumps(r), flush=True)
        torch.cuda.empty_cache()
