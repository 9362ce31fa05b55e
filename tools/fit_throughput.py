"""`Trainer.fit` (the loop behind `python -m viscy_amd fit`) timed at the bench configuration: an in-HBM synthetic data module
hands the trainer the bench batch every step, so what is timed is the trainer's own step path (captured TrainStep vs the eager
autograd loop) — VERDICT r2 item 4: "within 5 % of bench.py"."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch, nonzero_grn_  # noqa: E402
from viscy_amd.losses import MixedLoss  # noqa: E402
from viscy_amd.trainer import Trainer  # noqa: E402
from viscy_amd.vsunet import VSUNet  # noqa: E402

B = int(os.environ.get("B", 512))
STEPS = int(os.environ.get("STEPS", 12))


class SyntheticDM:
    """the data-module surface Trainer.fit touches, serving one resident device batch; the clock starts when batch `skip`
    is requested (the captured step exists by then)"""

    def __init__(self, x, t, steps, skip):
        self.batch, self.steps, self.skip, self.training, self.t0 = {"source": x, "target": t}, steps, skip, True, None

    def prepare_data(self): pass
    def setup(self, stage): pass
    def val_dataloader(self): return []
    def on_after_batch_transfer(self, batch, idx): return batch

    def train_dataloader(self):
        dm = self

        class _Loader:
            def __len__(self): return dm.steps + dm.skip

            def __iter__(self):
                for i in range(dm.steps + dm.skip):
                    if i == dm.skip:
                        torch.cuda.synchronize()
                        dm.t0 = time.perf_counter()
                    yield dm.batch

        return _Loader()


dev = torch.device("cuda", 0)
x, t = make_batch(B, 256, 256, dev)
out = {}
for mode in ("graph", "eager"):
    torch.manual_seed(42)
    module = VSUNet("UNeXt2", dict(in_channels=1, out_channels=2, in_stack_depth=5, backbone="convnextv2_tiny", head_pool=True,
                                   head_expansion_ratio=4, decoder_conv_blocks=2), loss_function=MixedLoss(0.5, 0.0, 0.5), lr=2e-4,
                    schedule="WarmupCosine")
    module.to(dev)
    nonzero_grn_(module.model)
    module.on_validation_epoch_end = lambda: None
    tr = Trainer(max_epochs=1, precision="bf16-mixed", graph_step=(mode == "graph"))
    dm = SyntheticDM(x, t, STEPS, skip=3)
    tr.fit(module, dm)  # (fit ends with a device synchronisation)
    dt = (time.perf_counter() - dm.t0) / STEPS
    out[mode] = {"ms_per_step": round(dt * 1e3, 2), "patches_per_s": round(B / dt, 1), "graph_steps": tr.graph_steps,
                 "final_loss": float(module.logged["loss/train"][-1])}
    del module, tr
    torch.cuda.empty_cache()
print(json.dumps(out))
