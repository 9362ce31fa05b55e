import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available() or os.environ.get("VSX_TEST_SELF"):
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


def build_sampling_plate(path: str):
    """The HCS plate behind tests/golden/hcs_sampling.pt (G11: the reference's SlidingWindowDataset ran over exactly this
    plate): 3 FOVs (2, 3, 7, 16, 24) from rng(7) with background stripes / an empty target timepoint, per-timepoint
    normalisation statistics, and a full-channel uint8 ``fg_mask`` array (target channels > 0.5).  Returns the arrays."""
    import numpy as np

    from viscy_amd.data import open_ome_zarr, write_hcs_plate

    rng = np.random.default_rng(7)
    pos = {}
    for i, name in enumerate(("A/1/0", "A/1/1", "B/2/0")):
        img = rng.random((2, 3, 7, 16, 24), dtype=np.float32)
        img[:, 2, :, :, : 6 * i] = 0.0          # FOV i: the left 6 i columns of the last channel are background
        img[1, 2] *= (i != 1)                    # FOV 1, t = 1: empty target
        pos[name] = img
    ch = ["Phase", "Membrane", "Nuclei"]
    meta = {c: {"fov_statistics": {"mean": 0.5, "std": 0.29},
                "timepoint_statistics": {"0": {"mean": 0.4, "std": 0.3}, "1": {"mean": 0.6, "std": 0.2}}} for c in ch}
    write_hcs_plate(path, pos, ch, norm_meta=meta)
    plate = open_ome_zarr(path, mode="r+")
    for name, p_ in plate.positions():
        mask = np.zeros_like(pos[name], dtype=np.uint8)
        mask[:, 1:] = (pos[name][:, 1:] > 0.5).astype(np.uint8)
        p_.create_image("fg_mask", mask, chunks=(1, 1, 1, 16, 24))
    return pos, ch
