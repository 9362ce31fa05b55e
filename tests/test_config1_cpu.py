"""BASELINE.json configs[0]: "Cytoland VSUNet '2D' tiny (1→1 ch, 256×256): one training step on PyTorch CPU, synthetic HCS
zarr (plumbing, no GPU)".  The model is the plain-PyTorch restatement of the reference's Unet2d (viscy_amd/unet2d.py, pinned
bit-exactly to the reference by oracle/validate_against_reference.py::g5_unet2d → tests/golden/unet2d.pt); everything around
it is the same HCSDataModule / VSUNet / Trainer code the accelerated path uses."""

import os

import numpy as np
import pytest
import torch
from torch import nn

from tests.conftest import load_golden
from viscy_amd.data import HCSDataModule
from viscy_amd.data.ome_zarr import open_ome_zarr, write_hcs_plate
from viscy_amd.trainer import Trainer
from viscy_amd.transforms import NormalizeSampled
from viscy_amd.unet2d import Unet2d
from viscy_amd.vsunet import VSUNet


def test_unet2d_matches_the_reference_golden():
    g = load_golden("unet2d.pt")  # written by the reference's own Unet2d (oracle/validate_against_reference.py::g5_unet2d)
    net = Unet2d(**g["kwargs"])
    assert list(net.state_dict()) == list(g["state_dict"])  # same keys in the same order
    net.load_state_dict(g["state_dict"])
    net.train()  # the generator's order: train pass, eval pass (BatchNorm statistics after one update), train pass + backward
    assert torch.equal(net(g["x"]), g["y_train"])
    net.eval()
    with torch.no_grad():
        assert torch.equal(net(g["x"]), g["y_eval"])
    net.train()
    net(g["x"]).square().mean().backward()
    assert torch.equal(net.down_conv_block_0.Conv2d_0.weight.grad, g["grad_first_conv"])


def test_unet2d_constructor_contract():
    net = Unet2d()
    assert net.num_filters == [16, 32, 64, 128, 256] and net.task == "seg"
    assert net.down_conv_block_0.drop_p == 0.0  # the reference builds Dropout2d(int(0.2)): the identity
    assert "terminal_block.resid_conv.weight" in net.state_dict() and "terminal_block.batch_norm_0.weight" not in net.state_dict()
    with pytest.raises(AssertionError, match="num_blocks \\+ 1"):
        Unet2d(num_filters=(4, 8))
    with pytest.raises(ValueError, match="odd"):
        Unet2d(kernel_size=(2, 3))
    x = torch.randn(1, 1, 1, 32, 48)
    with pytest.raises(AssertionError, match="square"):
        net(x, validate_input=True)
    assert net(x).shape == (1, 1, 1, 32, 48)
    # seg task: the terminal block ends in ReLU; reg: linear
    assert net(x).min() >= 0 and Unet2d(task="reg")(x).min() < 0


@pytest.fixture()
def plate_256(tmp_path):
    rng = np.random.default_rng(7)
    pos = {f"A/{c}/0": rng.random((1, 2, 1, 256, 256), dtype=np.float32) for c in (1, 2, 3, 4)}
    meta = {ch: {"fov_statistics": {"mean": 0.5, "std": 0.29}, "dataset_statistics": {"mean": 0.5, "std": 0.29}}
            for ch in ("Phase3D", "Nuclei")}
    path = str(tmp_path / "c1.zarr")
    write_hcs_plate(path, pos, ["Phase3D", "Nuclei"], norm_meta=meta)
    return path, pos


def _datamodule(path):
    return HCSDataModule(path, "Phase3D", "Nuclei", z_window_size=1, batch_size=2, num_workers=0, yx_patch_size=(256, 256),
                         normalizations=[NormalizeSampled(["Phase3D", "Nuclei"], "fov_statistics")], split_ratio=0.5,
                         normalize_on_device=False)


def test_config1_one_training_step_on_cpu(plate_256, tmp_path):
    path, _ = plate_256
    cfg = dict(in_channels=1, out_channels=1, num_blocks=2, num_filters=(8, 16, 32), task="reg")
    torch.manual_seed(3)
    module = VSUNet(architecture="2D", model_config=cfg, lr=2e-3, schedule="WarmupCosine", warmup_steps=0)
    assert isinstance(module.model, Unet2d) and isinstance(module.loss_function, nn.MSELoss)  # engine.py:197 default
    before = {k: v.clone() for k, v in module.state_dict().items()}

    # the same step by hand: same split (seeded), same batch, torch AdamW at the schedule's first learning rate
    torch.manual_seed(3)
    twin = Unet2d(**cfg)
    dm0 = _datamodule(path)
    dm0.setup("fit")
    torch.manual_seed(11)
    batch = next(iter(dm0.train_dataloader()))
    assert batch["source"].shape == (2, 1, 1, 256, 256)
    opt = torch.optim.AdamW(twin.parameters(), lr=2e-3)
    twin.train()
    want_loss = nn.functional.mse_loss(twin(batch["source"]), batch["target"])
    want_loss.backward()
    opt.step()

    dm = _datamodule(path)
    tr = Trainer(fast_dev_run=True, precision="32-true", default_root_dir=str(tmp_path / "run"))
    torch.manual_seed(11)
    tr.fit(module, dm)
    assert tr.global_step == 1 and tr.finished
    assert module.logged["loss/train"][-1] == pytest.approx(want_loss.item(), rel=1e-6)
    assert len(module.logged["loss/val/0"]) == 1 and np.isfinite(float(module.logged["loss/val/0"][0]))
    for k, v in twin.state_dict().items():
        if k.endswith("num_batches_tracked"):
            continue
        torch.testing.assert_close(module.state_dict()["model." + k], v, rtol=1e-6, atol=1e-7, msg=k)
    moved = [k for k, v in module.state_dict().items() if v.is_floating_point() and not torch.equal(v, before[k])]
    assert any("down_conv_block_0.Conv2d_0.weight" in k for k in moved) and any("terminal_block.Conv2d_0" in k for k in moved)

    # Lightning-layout checkpoint → a fresh module (the reference's ckpt_path contract) → predict over the plate
    ckpt = os.path.join(str(tmp_path / "run"), "checkpoints", "last.ckpt")
    again = VSUNet(architecture="2D", model_config=cfg, ckpt_path=ckpt)
    for k, v in module.state_dict().items():
        assert torch.equal(again.state_dict()[k], v), k
    from viscy_amd.prediction_writer import HCSPredictionWriter

    out = str(tmp_path / "pred.zarr")
    dmp = HCSDataModule(path, "Phase3D", "Nuclei", z_window_size=1, batch_size=2, num_workers=0,
                        normalizations=[NormalizeSampled(["Phase3D"], "fov_statistics")], normalize_on_device=False)
    Trainer(precision="32-true", callbacks=[HCSPredictionWriter(out)]).predict(again, dmp)
    again.eval()
    for name, p in open_ome_zarr(out).positions():
        assert p.channel_names == ["Nuclei_prediction"] and p["0"].shape == (1, 1, 1, 256, 256)
        src = torch.from_numpy(plate_256[1][name][:, :1])
        with torch.no_grad():
            want = again.model((src - 0.5) / (0.29 + 1e-8))
        torch.testing.assert_close(torch.from_numpy(p["0"][:, :, :]), want, rtol=1e-5, atol=1e-6)
