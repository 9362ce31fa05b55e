"""BUILD CONTAINER ONLY (needs /root/reference): times the directly importable pieces of the reference beside their
restatements under oracle/ on this host's cores (CPU-baseline row: "to show the restatement is not slower /
faster by construction").  Writes profiles/r02_cpu_ref_vs_port.json.

    python tools/cpu_ref_vs_port.py
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import validate_against_reference as V  # noqa: E402  (module-level helpers only; nothing is validated here)
from oracle import loss_ref, unext2_ref  # noqa: E402


def timeit(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    out = {"host_cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "rows": []}
    # ---- UNeXt2Stem (reference components/stems.py imported directly)
    ref = V._load("ref_stems_t", f"{V.REF}/viscy-models/src/viscy_models/components/stems.py")
    torch.manual_seed(0)
    r = ref.UNeXt2Stem(1, 96, (5, 4, 4), 5)
    o = unext2_ref.UNeXt2Stem(1, 96, (5, 4, 4), 5)
    o.load_state_dict(r.state_dict())
    x = torch.randn(2, 1, 5, 256, 256)
    with torch.no_grad():
        tr, to = timeit(lambda: r(x)), timeit(lambda: o(x))
    out["rows"].append({"piece": "UNeXt2Stem forward (2,1,5,256,256)", "reference_s": tr, "port_s": to, "port_over_reference": to / tr})
    # ---- ms_ssim_25d / MixedLoss (reference metrics.py + mixed_loss.py on empty third-party stubs)
    V._stub("skimage"); V._stub("skimage.measure", label=None, regionprops=None)
    V._stub("torchmetrics"); V._stub("torchmetrics.detection"); V._stub("torchmetrics.detection.mean_ap", MeanAveragePrecision=None)
    V._stub("torchvision"); V._stub("torchvision.ops", masks_to_boxes=None)
    V._stub("viscy_utils"); V._stub("viscy_utils.evaluation")
    metrics = V._load("viscy_utils.evaluation.metrics", f"{V.REF}/viscy-utils/src/viscy_utils/evaluation/metrics.py")
    V._stub("viscy_utils.losses")
    ml = V._load("viscy_utils.losses.mixed_loss", f"{V.REF}/viscy-utils/src/viscy_utils/losses/mixed_loss.py")
    g = torch.Generator().manual_seed(0)
    t = torch.rand((2, 2, 5, 256, 256), generator=g)
    p = t + 0.1 * torch.randn(t.shape, generator=g)

    def ref_loss():
        q = p.clone().requires_grad_(True)
        ml.MixedLoss(0.5, 0.0, 0.5)(q, t).backward()

    def port_loss():
        q = p.clone().requires_grad_(True)
        loss_ref.mixed_loss(q, t, 0.5, 0.0, 0.5).backward()

    tr, to = timeit(ref_loss, 3), timeit(port_loss, 3)
    out["rows"].append({"piece": "MixedLoss(0.5,0,0.5) forward+backward (2,2,5,256,256)", "reference_s": tr, "port_s": to, "port_over_reference": to / tr})
    with torch.no_grad():
        tr, to = timeit(lambda: metrics.ms_ssim_25d(p, t, clamp=True), 3), timeit(lambda: loss_ref.ms_ssim_25d(p, t, clamp=True), 3)
    out["rows"].append({"piece": "ms_ssim_25d forward (2,2,5,256,256)", "reference_s": tr, "port_s": to, "port_over_reference": to / tr})
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r02_cpu_ref_vs_port.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
