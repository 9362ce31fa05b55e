"""CPU: host-side logic around the hot path — OME-Zarr HCS I/O, HCSDataModule batch contract, DP sharding
(world_size-2 gloo), LR schedule, CPU-worker normalisation against the reference-generated golden."""

import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import load_golden
from viscy_amd.data import HCSDataModule, SlidingWindowDataset, open_ome_zarr, write_hcs_plate
from viscy_amd.optim import warmup_cosine_lambda
from viscy_amd.parallel import FlatDataParallel, shard_indices
from viscy_amd.transforms import MinMaxSampled, NormalizeSampled


@pytest.fixture(scope="module")
def tiny_hcs_zarr():
    """4 FOVs (1,2,5,128,128) float32 rng(42) with norm meta mean .5 / std .29 — the reference's integration fixture
    (applications/cytoland/tests/conftest.py:243-268)."""
    d = tempfile.mkdtemp()
    rng = np.random.default_rng(42)
    pos = {f"A/{c}/{f}": rng.random((1, 2, 5, 128, 128), dtype=np.float32) for c in (1, 2) for f in (0, 1)}
    meta = {ch: {"fov_statistics": {"mean": 0.5, "std": 0.29}, "dataset_statistics": {"mean": 0.5, "std": 0.29}}
            for ch in ("Phase3D", "Nuclei")}
    path = os.path.join(d, "tiny.zarr")
    write_hcs_plate(path, pos, ["Phase3D", "Nuclei"], norm_meta=meta)
    return path, pos


@pytest.mark.parametrize("compress,chunks", [(False, None), (True, (1, 1, 1, 64, 128)), (False, (1, 1, 2, 50, 70))])
def test_zarr_roundtrip(compress, chunks):
    d = tempfile.mkdtemp()
    rng = np.random.default_rng(0)
    arr = rng.random((2, 3, 7, 90, 130), dtype=np.float32)
    write_hcs_plate(os.path.join(d, "p.zarr"), {"B/3/0": arr}, ["a", "b", "c"], chunks=chunks, compress=compress)
    plate = open_ome_zarr(os.path.join(d, "p.zarr"))
    (name, pos), = list(plate.positions())
    assert name == "B/3/0" and pos.channel_names == ["a", "b", "c"] and pos.get_channel_index("c") == 2
    img = pos["0"]
    assert img.shape == arr.shape and (img.frames, img.channels, img.slices, img.height, img.width) == arr.shape
    np.testing.assert_array_equal(img.oindex[slice(1, 2), [2, 0], slice(2, 6)], arr[1:2][:, [2, 0], 2:6])
    with pytest.raises(FileNotFoundError):
        open_ome_zarr(os.path.join(d, "missing.zarr"))


def test_datamodule_fit_contract(tiny_hcs_zarr):
    path, pos = tiny_hcs_zarr
    dm = HCSDataModule(path, "Phase3D", "Nuclei", z_window_size=5, batch_size=2, num_workers=0, yx_patch_size=(128, 128),
                       normalizations=[NormalizeSampled(["Phase3D", "Nuclei"], "fov_statistics")], split_ratio=0.5,
                       normalize_on_device=False)  # the reference's worker-side order (hcs.py:783)
    assert HCSDataModule(path, "Phase3D", "Nuclei", 5, normalizations=[NormalizeSampled(["Phase3D"], "fov_statistics")]).normalize_on_device
    dm.setup("fit")
    assert len(dm.train_dataset) == 2 and len(dm.val_dataset) == 2
    b = next(iter(dm.train_dataloader()))
    assert b["source"].shape == (2, 1, 5, 128, 128) and b["target"].shape == (2, 1, 5, 128, 128)
    assert b["source"].dtype == torch.float32
    names, t, z = b["index"]
    assert all(n.startswith("/A/") and n.endswith("/0") for n in names) and t.tolist() == [0, 0] and z.tolist() == [0, 0]
    assert b["norm_meta"]["Phase3D"]["fov_statistics"]["mean"].shape == (2,)
    raw = pos[names[0][1:-2]][0, 0]
    torch.testing.assert_close(b["source"][0, 0], (torch.from_numpy(raw) - 0.5) / (0.29 + 1e-8))
    assert dm.on_after_batch_transfer(b, 0) is b
    # same seed → same split; wrong patch size → the reference's error
    dm2 = HCSDataModule(path, "Phase3D", "Nuclei", z_window_size=5, batch_size=2, num_workers=0, yx_patch_size=(64, 64), split_ratio=0.5)
    dm2.setup("fit")
    assert [p.name for p in dm2.train_dataset.positions] == [p.name for p in dm.train_dataset.positions]
    with pytest.raises(ValueError, match="does not match expected"):
        dm2.on_after_batch_transfer(next(iter(dm2.train_dataloader())), 0)
    with pytest.raises(ValueError, match="No positions left"):
        HCSDataModule(path, "Phase3D", "Nuclei", 5, include_fov_names=["Z/9/9"]).setup("fit")


def test_sliding_windows_and_predict(tiny_hcs_zarr):
    path, pos = tiny_hcs_zarr
    dm = HCSDataModule(path, "Phase3D", "Nuclei", z_window_size=3, batch_size=4, num_workers=0)
    dm.setup("predict")
    ds = dm.predict_dataset
    assert len(ds) == 4 * (5 - 3 + 1) and "target" not in ds[0]
    s = ds[4]  # second FOV, z = 1
    assert s["index"][2] == 1 and s["source"].shape == (1, 3, 128, 128)
    with pytest.raises(ValueError, match="exceeds"):
        SlidingWindowDataset(ds.positions, {"source": ["Phase3D"]}, z_window_size=9)


def test_normalize_cpu_worker_path_matches_reference_golden():
    g = load_golden("normalize.pt")  # produced by the reference's NormalizeSampled / MinMaxSampled
    meta = {"ch": {"fov_statistics": {"mean": g["mean"], "std": g["std"]}}}
    out = NormalizeSampled(["ch"], "fov_statistics")({"ch": g["x"].clone(), "norm_meta": meta})
    assert torch.equal(out["ch"], g["y"]) and "norm_meta" in out
    kat = {"ch": {"fov_statistics": {"mean": torch.tensor(60.0), "std": torch.tensor(10.0), "p1": torch.tensor(55.0), "p99": torch.tensor(65.0)}}}
    assert torch.equal(NormalizeSampled("ch", "fov_statistics", remove_meta=True)({"ch": g["kat_in"].clone(), "norm_meta": kat})["ch"], g["kat_out"])
    assert torch.equal(MinMaxSampled("ch", "fov_statistics")({"ch": g["kat_in"].clone(), "norm_meta": kat})["ch"], g["mm_out"])
    with pytest.raises(ValueError, match="Invalid data_range"):
        MinMaxSampled("ch", "fov_statistics", data_range="nope")


def test_warmup_cosine_schedule():
    # MONAI WarmupCosineSchedule(warmup_steps=3, t_total=10, warmup_multiplier=1e-3)
    lam = [warmup_cosine_lambda(s, 3, 10, 1e-3) for s in range(11)]
    assert lam[0] == pytest.approx(1e-3) and lam[3] == pytest.approx(1.0) and lam[10] == pytest.approx(0.0, abs=1e-12)
    assert all(a < b for a, b in zip(lam[:3], lam[1:4])) and all(a > b for a, b in zip(lam[3:10], lam[4:11]))
    assert lam[1] == pytest.approx(1e-3 + (1 - 1e-3) / 3)


def test_shard_indices_equal_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler

    ds = list(range(103))
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            ref = DistributedSampler(ds, num_replicas=world, rank=r, shuffle=True, seed=42, drop_last=True)
            ref.set_epoch(3)
            mine = shard_indices(len(ds), r, world, seed=42, epoch=3, shuffle=True, drop_last=True)
            assert mine == list(ref)
            seen += mine
        assert len(set(seen)) == len(seen) == 103 // world * world  # disjoint shards


class _FakeEngine:
    def __init__(self, n, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.flat = torch.randn(n, generator=g)
        self.flat_grad = torch.zeros(n)
        self.bucket_bounds = [(0, n // 3), (n // 3, 2 * n // 3), (2 * n // 3, n)]
        self.on_bucket_ready = None


class _FakeOpt:
    grad_scale = 1.0


def _ddp_worker(rank, world, init_file, out):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    eng, opt = _FakeEngine(1000, rank), _FakeOpt()
    ddp = FlatDataParallel(eng, opt)
    params_after_broadcast = eng.flat.clone()
    eng.flat_grad.copy_(torch.arange(1000, dtype=torch.float32) * (rank + 1))
    for i in range(3):  # what Engine.backward does as buckets complete
        eng.on_bucket_ready(i)
    ddp.finish()
    loss = ddp.all_reduce_mean(torch.tensor(float(rank + 1)))
    out[rank] = (params_after_broadcast, eng.flat_grad.clone(), opt.grad_scale, float(loss))
    dist.destroy_process_group()


def test_flat_data_parallel_gloo_world2():
    world = 2
    init_file = tempfile.mktemp()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ddp_worker, args=(world, init_file, out), nprocs=world, join=True)
    (p0, g0, s0, l0), (p1, g1, s1, l1) = out[0], out[1]
    assert torch.equal(p0, p1)                       # rank 0's parameters were broadcast
    expect = torch.arange(1000, dtype=torch.float32) * 3.0
    assert torch.equal(g0, expect) and torch.equal(g1, expect)   # SUM all-reduce of every bucket
    assert s0 == s1 == 0.5                           # averaging folded into the optimizer's grad_scale
    assert l0 == l1 == 1.5                           # sync_dist mean of a logged scalar


# ------------------------------------------------------------------------------------------------ HCSPredictionWriter (§8 f1)
def _expected_blend(preds, Z, zw):
    """independent statement of the reference's out-of-core feathering (prediction_writer.py:74-111,300-326): windows
    arrive in Z order; a window at offset z is merged into what is already stored there."""
    C, _, Y, X = preds[0].shape
    vol = np.zeros((C, Z, Y, X), np.float32)
    for z, new in enumerate(preds):
        if z == 0:
            vol[:, :zw] = new
            continue
        samples = min(z + 1, zw)
        f = np.array([min(i + 1, samples) for i in reversed(range(zw))], np.float64)[None, :, None, None]
        vol[:, z : z + zw] = (vol[:, z : z + zw] * (f - 1) / f + new / f).astype(np.float32)
    return vol


class _StubTrainer:
    def __init__(self, dm):
        self.datamodule = dm


def test_prediction_writer_blends_z_windows_and_appends_channels(tiny_hcs_zarr, tmp_path):
    from viscy_amd.prediction_writer import HCSPredictionWriter

    path, pos = tiny_hcs_zarr
    zw = 3
    dm = HCSDataModule(path, "Phase3D", ["Nuclei", "Membrane"], z_window_size=zw, batch_size=2, num_workers=0)
    dm.setup("predict")
    out = str(tmp_path / "pred.zarr")
    w = HCSPredictionWriter(out)
    assert w.interval == "batch"
    w.on_predict_start(_StubTrainer(dm), None)
    rng = np.random.default_rng(0)
    sent = {}
    for j, batch in enumerate(dm.predict_dataloader()):
        B = batch["source"].shape[0]
        pred = torch.from_numpy(rng.random((B, 2, zw, 128, 128), dtype=np.float32))
        for i in range(B):
            sent.setdefault(batch["index"][0][i], []).append(pred[i].numpy())
        w.write_on_batch_end(None, None, pred, None, batch, j, 0)
    w.on_predict_end(None, None)
    plate = open_ome_zarr(out)
    names = [n for n, _ in plate.positions()]
    assert sorted(names) == sorted(n[1:-2] for n in sent)  # "/A/1/0/0" -> "A/1/0"
    Z = 5
    for n, p in plate.positions():
        assert p.channel_names == ["Nuclei_prediction", "Membrane_prediction"]
        img = p["0"]
        assert img.shape == (1, 2, Z, 128, 128) and img.chunks == (1, 1, 1, 128, 128)
        assert p.scale == [1.0] * 5
        got = img.oindex[slice(0, 1), [0, 1], slice(0, Z)][0]
        np.testing.assert_allclose(got, _expected_blend(sent["/" + n + "/0"], Z, zw), rtol=1e-6, atol=1e-7)
    # an existing store: the same channels are refused, overwrite replaces, write_input is refused
    with pytest.raises(FileExistsError, match="already exists"):
        HCSPredictionWriter(out).on_predict_start(_StubTrainer(dm), None)
    with pytest.raises(FileExistsError, match="existing store"):
        HCSPredictionWriter(out, write_input=True).on_predict_start(_StubTrainer(dm), None)
    HCSPredictionWriter(out, overwrite=True).on_predict_start(_StubTrainer(dm), None)
    # appending prediction channels to a copy of the input plate (r+): arrays grow along C, existing data stays
    import shutil

    both = str(tmp_path / "both.zarr")
    shutil.copytree(path, both)
    w2 = HCSPredictionWriter(both)
    w2.on_predict_start(_StubTrainer(dm), None)
    for j, batch in enumerate(dm.predict_dataloader()):
        w2.write_on_batch_end(None, None, torch.ones(batch["source"].shape[0], 2, zw, 128, 128), None, batch, j, 0)
    w2.on_predict_end(None, None)   # the last open stack reaches the store here (round 5: every slice is written once)
    p0 = next(iter(open_ome_zarr(both).positions()))[1]
    assert p0.channel_names[-2:] == ["Nuclei_prediction", "Membrane_prediction"] and p0["0"].shape[1] == len(p0.channel_names)
    raw = pos[next(iter(pos))]
    np.testing.assert_array_equal(p0["0"].oindex[slice(0, 1), [0], slice(0, 5)][0, 0], raw[0, 0])
    np.testing.assert_allclose(p0["0"].oindex[slice(0, 1), [len(p0.channel_names) - 1], slice(0, 5)], 1.0)


def test_prediction_writer_writes_every_slice_once_and_reads_nothing_back(tiny_hcs_zarr, tmp_path, monkeypatch):
    """round 5: consecutive Z windows are blended in a running stack next to the prediction; the store sees each (channel, slice)
    chunk exactly once and is never read.  Windows in another order take the reference's read-blend-write path: same result."""
    from viscy_amd.data import ome_zarr
    from viscy_amd.prediction_writer import HCSPredictionWriter

    path, pos = tiny_hcs_zarr
    zw, Z = 3, 5
    dm = HCSDataModule(path, "Phase3D", ["Nuclei", "Membrane"], z_window_size=zw, batch_size=2, num_workers=0)
    dm.setup("predict")
    batches = list(dm.predict_dataloader())
    rng = np.random.default_rng(1)
    preds = [torch.from_numpy(rng.random((b["source"].shape[0], 2, zw, 128, 128), dtype=np.float32)) for b in batches]
    writes, reads = [], []
    orig_w, orig_r = ome_zarr.ImageArray._write_chunk, ome_zarr.ImageArray._chunk
    monkeypatch.setattr(ome_zarr.ImageArray, "_write_chunk", lambda self, idx, data: (writes.append((self.path, idx)), orig_w(self, idx, data))[1])
    monkeypatch.setattr(ome_zarr.ImageArray, "_chunk", lambda self, idx: (reads.append((self.path, idx)), orig_r(self, idx))[1])

    def run(out, order):
        w = HCSPredictionWriter(out)
        w.on_predict_start(_StubTrainer(dm), None)
        for j in order:
            w.write_on_batch_end(None, None, preds[j], None, batches[j], j, 0)
        w.on_predict_end(None, None)

    a = str(tmp_path / "in_order.zarr")
    run(a, range(len(batches)))
    n_pos = len(list(open_ome_zarr(a).positions()))
    assert len(writes) == len(set(writes)) == n_pos * 2 * Z and reads == []   # every chunk once, nothing read back
    writes.clear()
    b = str(tmp_path / "reversed.zarr")
    run(b, reversed(range(len(batches))))                                      # windows arrive z-descending across batches
    assert len(reads) > 0                                                      # ... so the store had to be consulted
    reads.clear()
    monkeypatch.undo()
    sent = {}
    for bt, pr in zip(batches, preds):
        for i in range(pr.shape[0]):
            sent.setdefault(bt["index"][0][i], []).append(pr[i].numpy())
    for n, p in open_ome_zarr(a).positions():
        got = p["0"].oindex[slice(0, 1), [0, 1], slice(0, Z)][0]
        np.testing.assert_allclose(got, _expected_blend(sent["/" + n + "/0"], Z, zw), rtol=1e-6, atol=1e-7)


def test_prediction_writer_2d_target_and_write_input(tiny_hcs_zarr, tmp_path):
    """target_2d: one slice per window written at z + z_window // 2, no blending; write_input stores the centre slices."""
    from viscy_amd.prediction_writer import HCSPredictionWriter

    path, pos = tiny_hcs_zarr
    dm = HCSDataModule(path, "Phase3D", "Nuclei", z_window_size=3, batch_size=3, num_workers=0, target_2d=True)
    dm.setup("predict")
    out = str(tmp_path / "p2d.zarr")
    w = HCSPredictionWriter(out, write_input=True)
    w.on_predict_start(_StubTrainer(dm), None)
    assert w.z_padding == 1
    for j, batch in enumerate(dm.predict_dataloader()):
        z = batch["index"][2].float().view(-1, 1, 1, 1, 1)
        w.write_on_batch_end(None, None, z + torch.zeros(len(z), 1, 1, 128, 128), None, batch, j, 0)
    for n, p in open_ome_zarr(out).positions():
        assert p.channel_names == ["Phase3D", "Nuclei", "Nuclei_prediction"]
        img = p["0"]
        assert img.shape == (1, 3, 4, 128, 128)  # windows z = 0..2 land on slices 1..3
        got = img.oindex[slice(0, 1), [2], slice(0, 4)][0, 0]
        for zi in range(3):
            np.testing.assert_allclose(got[zi + 1], float(zi))
        np.testing.assert_array_equal(img.oindex[slice(0, 1), [0], slice(1, 4)][0, 0], pos[n][0, 0][1:4])


# ------------------------------------------------------------------------------------------------ DynaCLR global negatives (§8 f3)
def _gather_worker(rank, world, init_file, out):
    from oracle.contrastive_ref import NTXentLoss as RefLoss
    from viscy_amd.parallel import all_gather_with_local_grad, scale_for_mean_reduction

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(world, 2, 3, 10, generator=g)          # [rank, anchor/positive, B, features]
    P = torch.randn(10, 6, generator=g).requires_grad_(True)  # the shared "model"
    a, p = X[rank, 0] @ P, X[rank, 1] @ P
    ga, gp = all_gather_with_local_grad(a), all_gather_with_local_grad(p)
    idx = torch.arange(ga.shape[0])
    loss = RefLoss(temperature=0.3)(torch.cat((ga, gp)), torch.cat((idx, idx)))
    scale_for_mean_reduction(loss).backward()
    grad = P.grad.clone()
    dist.all_reduce(grad)                                   # what FlatDataParallel does, then 1 / world in the optimiser
    out[rank] = (float(loss), grad / world, ga.detach().clone())
    dist.destroy_process_group()


def test_all_gather_with_local_grad_reproduces_the_global_batch_gradient():
    """world-2 gloo: gathered NT-Xent (every rank evaluates the global batch, differentiates its own rows) + mean-reduced
    data-parallel gradients == one process holding the whole batch"""
    from oracle.contrastive_ref import NTXentLoss as RefLoss

    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gather_worker, args=(world, tempfile.mktemp(), out), nprocs=world, join=True)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(world, 2, 3, 10, generator=g)
    P = torch.randn(10, 6, generator=g).requires_grad_(True)
    a, p = X[:, 0].reshape(-1, 10) @ P, X[:, 1].reshape(-1, 10) @ P   # rank-major, as the gather concatenates
    idx = torch.arange(a.shape[0])
    loss = RefLoss(temperature=0.3)(torch.cat((a, p)), torch.cat((idx, idx)))
    loss.backward()
    for r in range(world):
        l_r, g_r, ga_r = out[r]
        assert abs(l_r - loss.item()) < 1e-6
        torch.testing.assert_close(ga_r, a.detach(), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(g_r, P.grad, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------ mmap preload (§8 f4)
def test_mmap_preload_serves_the_same_batches(tiny_hcs_zarr, tmp_path):
    """hcs.py:241-349 protocol: prepare_data stages the fit FOVs once (`.done` marker, partial caches rebuilt, fingerprint
    per channel / filter set); the fit batches read from the buffer are identical to those read from the zarr store."""
    path, _ = tiny_hcs_zarr
    kw = dict(z_window_size=3, batch_size=2, num_workers=0, normalizations=[NormalizeSampled(["Phase3D"], "fov_statistics")],
              normalize_on_device=False)
    plain = HCSDataModule(path, "Phase3D", "Nuclei", **kw)
    plain.setup("fit")
    mm = HCSDataModule(path, "Phase3D", "Nuclei", mmap_preload=True, scratch_dir=tmp_path, **kw)
    with pytest.raises(RuntimeError, match="prepare_data"):
        mm.setup("fit")
    cache = mm._mmap_cache_dir
    assert str(cache).startswith(str(tmp_path)) and cache != HCSDataModule(path, "Nuclei", "Phase3D", mmap_preload=True,
                                                                           scratch_dir=tmp_path, **kw)._mmap_cache_dir
    cache.mkdir(parents=True)
    (cache / "data.mmap").write_bytes(b"partial")       # a killed preload left debris without the marker
    mm.prepare_data()
    assert (cache / ".done").exists()
    stamp = (cache / "data.mmap").stat().st_mtime_ns
    mm.prepare_data()                                    # second call: cache found, nothing rewritten
    assert (cache / "data.mmap").stat().st_mtime_ns == stamp
    mm.setup("fit")
    assert len(mm.train_dataset) == len(plain.train_dataset) and len(mm.val_dataset) == len(plain.val_dataset)
    for ds_a, ds_b in ((plain.train_dataset, mm.train_dataset), (plain.val_dataset, mm.val_dataset)):
        for i in range(len(ds_a)):
            a, b = ds_a[i], ds_b[i]
            assert a["index"] == b["index"]
            assert torch.equal(a["source"], b["source"]) and torch.equal(a["target"], b["target"])
    ba, bb = next(iter(plain.val_dataloader())), next(iter(mm.val_dataloader()))
    assert torch.equal(ba["source"], bb["source"]) and ba["index"][0] == bb["index"][0]
    mm.setup("predict")                                  # predict reads the store directly
    assert not isinstance(mm.predict_dataset.positions[0], type(mm.train_dataset.positions[0]))


def test_rand_weighted_cropd_multi_sample_host_crop(tiny_hcs_zarr):
    """CPU-worker RandWeightedCropd (MONAI semantics): N crops per stack, inside the volume, centres follow the weight map,
    reproducible from the random state, uniform when the map is empty; HCSDataModule flattens them into the batch"""
    from viscy_amd.transforms import RandWeightedCropd

    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 5, 96, 128, generator=g)
    w = torch.zeros(1, 5, 96, 128)
    w[:, :, 60:70, 90:100] = 1.0                       # all the mass in one blob
    t = RandWeightedCropd(["source", "target", "weight"], w_key="weight", spatial_size=(-1, 32, 48), num_samples=6).set_random_state(7)
    outs = t({"source": img, "target": img * 2, "weight": w, "norm_meta": None})
    assert isinstance(outs, list) and len(outs) == 6
    for o in outs:
        assert o["source"].shape == (1, 5, 32, 48) and torch.equal(o["target"], o["source"] * 2)
        assert o["weight"].sum() > 0                    # every window contains part of the blob: centres sit on it
    again = RandWeightedCropd(["source", "target", "weight"], "weight", (-1, 32, 48), 6).set_random_state(7)(
        {"source": img, "target": img * 2, "weight": w})
    assert all(torch.equal(a["source"], b["source"]) for a, b in zip(outs, again))
    flat = RandWeightedCropd(["source"], "weight", (5, 32, 48), 200).set_random_state(1)({"source": img, "weight": torch.zeros_like(w)})
    ys = {int((img[0, 0] == o["source"][0, 0, 0, 0]).nonzero()[0, 0]) for o in flat}
    assert len(ys) > 20                                 # empty map -> uniform over the valid centres
    # through the data module: 2 stacks x 3 samples = batch of 6 patches of the crop size
    path, _ = tiny_hcs_zarr
    dm = HCSDataModule(path, "Phase3D", "Nuclei", z_window_size=3, batch_size=6, num_workers=0, yx_patch_size=(32, 32),
                       augmentations=[RandWeightedCropd(["Phase3D", "Nuclei"], w_key="Nuclei", spatial_size=(-1, 32, 32), num_samples=3)])
    dm.setup("fit")
    assert dm.train_patches_per_stack == 3
    b = next(iter(dm.train_dataloader()))
    assert b["source"].shape == (6, 1, 3, 32, 32) and b["target"].shape == (6, 1, 3, 32, 32)


# ------------------------------------------------------------------------------------------------ BatchedRandAffined parameters
def test_affine_arguments_follow_the_reference_conventions():
    """host half of BatchedRandAffined — argument parsing and parameter sampling as viscy_transforms/_affine.py:165-357 and
    its tests (test_affine.py): ZYX -> kornia XYZ order, radians -> degrees, scale ranges, isotropic scale, 3-value shear
    shorthand, Z-shear scaling, the safe-crop scale floor; matrix composition vs the oracle restatement of kornia"""
    import math

    from oracle import transforms_ref as R
    from viscy_amd.transforms import BatchedRandAffined, kornia_affine_matrix3d

    # the published VSCyto3D fine-tuning recipe (finetune_a549_infected.yml)
    t = BatchedRandAffined(keys=["source", "target"], prob=0.8, rotate_range=[3.14, 0, 0], shear_range=[0.0, 0.05, 0.05],
                           scale_range=[[0.7, 1.3], [0.5, 1.5], [0.5, 1.5]])
    t.generator = torch.Generator().manual_seed(0)
    assert t.degrees[0] == (0.0, 0.0) and t.degrees[1] == (0.0, 0.0) and abs(t.degrees[2][1] - math.degrees(3.14)) < 1e-9
    assert t.scale == [(0.5, 1.5), (0.5, 1.5), (0.7, 1.3)]                       # (x, y, z)
    assert t.shears == [(0.0, 0.0)] * 3 + [(-0.05, 0.05), (-0.05, 0.05), (-0.0, 0.0)]
    prm = t.sample_parameters((64, 1, 20, 600, 600))
    assert prm["scale"][:, 2].min() >= 0.7 and prm["scale"][:, 2].max() <= 1.3   # test_affine_per_axis_scale
    assert prm["scale"][:, :2].min() >= 0.5 and prm["scale"][:, :2].max() <= 1.5
    assert (prm["angles"][:, :2] == 0).all() and prm["angles"][:, 2].abs().max() > 90
    assert prm["shears"][:, 3:5].abs().max() <= 0.05 * 20 / 600 + 1e-12           # scale_z_shear: Z facets * depth / yx
    assert 0.6 < prm["apply"].float().mean() < 0.95
    m = t.randomize((64, 1, 20, 600, 600))
    assert m.shape == (64, 3, 4) and m.dtype == torch.float32
    # flat range: independent per axis inside the range (test_affine_scale_range_not_inverted / anisotropic default)
    t2 = BatchedRandAffined(keys=["img"], prob=1.0, scale_range=[0.5, 1.5])
    s2 = t2.sample_parameters((8, 1, 8, 32, 32))["scale"]
    assert s2.min() >= 0.5 and s2.max() <= 1.5 and not all(torch.equal(s2[i, :1].expand(3), s2[i]) for i in range(8))
    t3 = BatchedRandAffined(keys=["img"], prob=1.0, scale_range=[0.5, 1.5], isotropic_scale=True)
    s3 = t3.sample_parameters((8, 1, 8, 32, 32))["scale"]
    assert torch.equal(s3[:, 0], s3[:, 1]) and torch.equal(s3[:, 0], s3[:, 2])
    with pytest.raises(ValueError, match="isotropic_scale=True cannot be combined"):
        BatchedRandAffined(keys=["img"], scale_range=[[0.9, 1.1], [0.5, 1.5], [0.5, 1.5]], isotropic_scale=True)
    with pytest.raises(ValueError, match="scale_range must be"):
        BatchedRandAffined(keys=["img"], scale_range=[0.5, 1.0, 1.5, 2.0])
    # test_compute_scale_floor_known_angles
    ang = torch.tensor([[0, 0, 0.0], [0, 0, 45.0], [0, 0, 90.0], [0, 0, 180.0]])
    fl = BatchedRandAffined._compute_scale_floor(ang, torch.Size([4, 1, 13, 624, 624]), (8, 512, 512))
    Rr = 624 / 512
    assert math.isclose(fl[0, 0].item(), 1 / Rr, rel_tol=1e-5) and math.isclose(fl[1, 0].item(), math.sqrt(2) / Rr, rel_tol=1e-5)
    assert math.isclose(fl[2, 0].item(), 1 / Rr, rel_tol=1e-5) and all(math.isclose(fl[i, 2].item(), 8 / 13, rel_tol=1e-5) for i in range(4))
    t4 = BatchedRandAffined(keys=["a"], prob=1.0, rotate_range=[3.14, 0, 0], scale_range=[[0.7, 1.3], [0.5, 1.5], [0.5, 1.5]],
                            safe_crop_size=[8, 512, 512])
    p4 = t4.sample_parameters((16, 1, 13, 624, 624))
    assert (p4["scale"] >= BatchedRandAffined._compute_scale_floor(p4["angles"], (16, 1, 13, 624, 624), (8, 512, 512)) - 1e-9).all()
    # matrix composition == the oracle's restatement of kornia's get_affine_matrix3d
    g = torch.Generator().manual_seed(1)
    ang = (torch.rand(6, 3, generator=g) - 0.5) * 120
    sc = 0.5 + torch.rand(6, 3, generator=g)
    sh = (torch.rand(6, 6, generator=g) - 0.5) * 20
    tr = (torch.rand(6, 3, generator=g) - 0.5) * 8
    torch.testing.assert_close(kornia_affine_matrix3d(ang, sc, sh, tr, (9, 40, 56)), R.kornia_affine_matrix3d(ang, sc, sh, tr, (9, 40, 56)),
                               rtol=1e-10, atol=1e-9)
    # a pure rotation about Z keeps the centre and rotates the YX plane
    M = kornia_affine_matrix3d(torch.tensor([[0.0, 0.0, 90.0]]), torch.ones(1, 3), torch.zeros(1, 6), torch.zeros(1, 3), (5, 11, 11))[0]
    c = torch.tensor([5.0, 5.0, 2.0, 1.0], dtype=torch.float64)
    torch.testing.assert_close(M @ c, c)
    assert abs(M[2, 2] - 1) < 1e-12 and abs(M[0, 0]) < 1e-12 and abs(abs(M[0, 1]) - 1) < 1e-12


# ------------------------------------------------------------------------------------------------ CombinedDataModule
def test_combined_loader_modes_and_datamodule(tiny_hcs_zarr):
    """viscy_data/combined.py:22-120 + the CombinedLoader semantics it relies on"""
    from viscy_amd.data import CombinedDataModule, CombinedLoader, CombineMode

    a, b = [1, 2, 3, 4, 5], ["x", "y"]
    assert [o for o, _, _ in CombinedLoader([a, b], "min_size")] == [[1, "x"], [2, "y"]]
    assert [o for o, _, _ in CombinedLoader([a, b], "max_size_cycle")] == [[1, "x"], [2, "y"], [3, "x"], [4, "y"], [5, "x"]]
    assert [o for o, _, _ in CombinedLoader([a, b], "max_size")] == [[1, "x"], [2, "y"], [3, None], [4, None], [5, None]]
    assert list(CombinedLoader([a, b], "sequential")) == [(1, 0, 0), (2, 1, 0), (3, 2, 0), (4, 3, 0), (5, 4, 0), ("x", 0, 1), ("y", 1, 1)]
    assert [len(CombinedLoader([a, b], m)) for m in ("min_size", "max_size_cycle", "max_size", "sequential")] == [2, 5, 5, 7]
    path, _ = tiny_hcs_zarr
    kw = dict(z_window_size=3, num_workers=0, normalizations=[NormalizeSampled(["Phase3D"], "fov_statistics")], normalize_on_device=False,
              yx_patch_size=(128, 128))
    dm = CombinedDataModule([HCSDataModule(path, "Phase3D", "Nuclei", batch_size=2, **kw), HCSDataModule(path, "Phase3D", "Nuclei", batch_size=3, **kw)],
                            train_mode="MAX_SIZE_CYCLE", val_mode=CombineMode.SEQUENTIAL)
    assert dm.train_mode == "max_size_cycle" and dm.val_mode == "sequential"
    dm.prepare_data()
    dm.setup("fit")
    tl = dm.train_dataloader()
    batch, bi, di = next(iter(tl))
    assert isinstance(batch, list) and len(batch) == 2 and batch[0]["source"].shape[0] == 2 and batch[1]["source"].shape[0] == 3
    assert len(tl) == max(len(d.train_dataloader()) for d in dm.data_modules)
    out = dm.on_after_batch_transfer(batch, 0)          # dispatched to the children, one sub-batch each
    assert isinstance(out, list) and out[1]["source"].shape[0] == 3
    idxs = [di for _, _, di in dm.val_dataloader()]
    assert idxs == sorted(idxs) and set(idxs) == {0, 1}
    dm.training = False
    assert all(d.training is False for d in dm.data_modules)


def _sharded_worker(rank, world, init_file, out):
    from viscy_amd.data import ShardedDistributedSampler

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    s = ShardedDistributedSampler(list(range(11)), shuffle=True, seed=3, drop_last=True)
    s.set_epoch(2)
    out[rank] = list(s)
    dist.destroy_process_group()


def test_sharded_sampler_and_batched_concat_datamodule(tiny_hcs_zarr):
    """viscy_data/distributed.py:16-58 (world-2 gloo): rank r permutes ITS contiguous shard; combined.py:186-378: the joint
    loader hands over per-dataset micro-batches that on_after_batch_transfer merges into one batch; CPU CenterSpatialCropd"""
    from viscy_amd.data import BatchedConcatDataModule, ConcatDataModule
    from viscy_amd.transforms import CenterSpatialCropd, RandWeightedCropd

    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sharded_worker, args=(2, tempfile.mktemp(), out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert sorted(r0) == [0, 1, 2, 3, 4] and sorted(r1) == [5, 6, 7, 8, 9]       # num_samples = 5: shards [0,5) and [5,10)
    g = torch.Generator().manual_seed(3 + 2)                                        # seed + epoch, one permutation per shard
    assert r0 == torch.randperm(5, generator=g).tolist() and r1 == (torch.randperm(5, generator=g) + 5).tolist()
    path, _ = tiny_hcs_zarr
    kw = dict(z_window_size=3, batch_size=3, num_workers=0, yx_patch_size=(32, 32), normalize_on_device=False,
              normalizations=[NormalizeSampled(["Phase3D"], "fov_statistics")])
    mk = lambda: HCSDataModule(path, "Phase3D", "Nuclei", augmentations=[  # noqa: E731
        RandWeightedCropd(["Phase3D", "Nuclei"], w_key="Nuclei", spatial_size=(-1, 48, 48), num_samples=2),
        CenterSpatialCropd(["Phase3D", "Nuclei"], roi_size=(-1, 32, 32))], **kw)
    with pytest.raises(ValueError, match="divisible by `num_samples`"):
        mk().setup("fit")                        # 3 % 2: standalone modules insist on the divisibility (hcs.py:787-800) ...
    dm = BatchedConcatDataModule([mk(), mk()])  # ... children of the batched container do not
    dm.setup("fit")
    assert dm.train_patches_per_stack == 2 and len(dm.train_dataset) == 2 * len(dm.data_modules[0].train_dataset)
    batch = next(iter(dm.train_dataloader()))
    assert isinstance(batch, list) and all("_dataset_idx" in mb for mb in batch)
    merged = dm.on_after_batch_transfer(batch, 0)
    assert merged["source"].shape == (6, 1, 3, 32, 32) and merged["target"].shape == (6, 1, 3, 32, 32)   # 3 stacks x 2 crops
    assert "norm_meta" not in merged and "_dataset_idx" not in merged
    with pytest.raises(ValueError, match="Inconsistent batch size"):
        ConcatDataModule([mk(), HCSDataModule(path, "Phase3D", "Nuclei", z_window_size=3, batch_size=4, num_workers=0)])


# ------------------------------------------------------------------------------------------------ YAML seam
def test_yaml_recipes_compose_and_instantiate(tmp_path, tiny_hcs_zarr):
    """viscy_utils/compose.py:31-140 semantics (base: lists, deep merge, lists replace, private keys stripped, cycles) and the
    class_path map: the published VSCyto3D fine-tuning recipe's tree (shrunk) builds the MI355X objects"""
    from viscy_amd import config as C
    from viscy_amd.data import CombinedDataModule
    from viscy_amd.losses import MixedLoss
    from viscy_amd.transforms import BatchedRandAffined, RandWeightedCropd
    from viscy_amd.vsunet import VSUNet

    path, _ = tiny_hcs_zarr
    (tmp_path / "recipes").mkdir()
    (tmp_path / "recipes" / "fit.yml").write_text(
        "trainer:\n  max_epochs: 7\n  precision: bf16-mixed\n  callbacks:\n    - class_path: lightning.pytorch.callbacks.LearningRateMonitor\n"
        "      init_args: {logging_interval: step}\n_anchors: &a {x: 1}\nmodel:\n  init_args:\n    lr: 0.5\n    model_config: {in_channels: 1, dims: [1, 2, 3, 4]}\n")
    (tmp_path / "leaf.yml").write_text(f"""
base:
  - recipes/fit.yml
model:
  class_path: cytoland.engine.VSUNet
  init_args:
    architecture: fcmae
    model_config:
      out_channels: 2
      encoder_blocks: [1, 1, 1, 1]
      encoder_drop_path_rate: 0.1
      dims: [16, 32, 64, 128]
      decoder_conv_blocks: 1
      in_stack_depth: 5
      pretraining: false
    loss_function:
      class_path: viscy_utils.losses.MixedLoss
      init_args: {{l1_alpha: 0.5, l2_alpha: 0.0, ms_dssim_alpha: 0.5}}
    lr: 0.0002
    schedule: WarmupCosine
trainer:
  strategy: ddp_find_unused_parameters_true
  logger:
    class_path: lightning.pytorch.loggers.TensorBoardLogger
    init_args: {{save_dir: /tmp}}
data:
  class_path: viscy_data.combined.CombinedDataModule
  init_args:
    train_mode: MAX_SIZE_CYCLE
    val_mode: SEQUENTIAL
    data_modules:
      - class_path: viscy_data.hcs.HCSDataModule
        init_args:
          data_path: {path}
          source_channel: Phase3D
          target_channel: [Nuclei]
          z_window_size: 5
          batch_size: 4
          num_workers: 0
          mmap_preload: true
          scratch_dir: {tmp_path}/scratch
          yx_patch_size: [64, 64]
          augmentations:
            - class_path: viscy_transforms.RandWeightedCropd
              init_args: {{keys: [Phase3D, Nuclei], w_key: Nuclei, spatial_size: [5, 96, 96], num_samples: 2}}
          normalizations:
            - class_path: viscy_transforms.NormalizeSampled
              init_args: {{keys: [Phase3D], level: fov_statistics, subtrahend: mean, divisor: std}}
          gpu_augmentations:
            - class_path: viscy_transforms.BatchedRandAffined
              init_args:
                keys: [source, target]
                prob: 0.8
                rotate_range: [3.14, 0, 0]
                shear_range: [0.0, 0.05, 0.05]
                scale_range: [[0.7, 1.3], [0.5, 1.5], [0.5, 1.5]]
            - class_path: viscy_transforms.BatchedCenterSpatialCropd
              init_args: {{keys: [source, target], roi_size: [5, 64, 64]}}
""")
    cfg = C.load_composed_config(tmp_path / "leaf.yml")
    assert "base" not in cfg and "_anchors" not in cfg
    assert cfg["trainer"]["max_epochs"] == 7 and cfg["trainer"]["strategy"] == "ddp_find_unused_parameters_true"
    assert cfg["model"]["init_args"]["lr"] == 0.0002                                   # leaf overrides base
    assert cfg["model"]["init_args"]["model_config"]["dims"] == [16, 32, 64, 128]      # lists replace
    assert cfg["model"]["init_args"]["model_config"]["in_channels"] == 1               # dicts merge
    (tmp_path / "a.yml").write_text("base: [b.yml]\n")
    (tmp_path / "b.yml").write_text("base: a.yml\n")
    with pytest.raises(ValueError, match="Circular"):
        C.load_composed_config(tmp_path / "a.yml")
    module, dm, trainer, skipped = C.build(cfg)
    assert isinstance(module, VSUNet) and isinstance(module.loss_function, MixedLoss) and module.schedule == "WarmupCosine"
    assert module.model.cfg["drop_path"] == [0.1] * 4
    assert isinstance(dm, CombinedDataModule) and dm.train_mode == "max_size_cycle" and dm.data_modules[0].mmap_preload
    assert isinstance(dm.data_modules[0].augmentations[0], RandWeightedCropd) and dm.data_modules[0].train_patches_per_stack == 2
    assert isinstance(dm.data_modules[0]._gpu_augmentations.transforms[0].affine, BatchedRandAffined)  # fused with the crop
    assert trainer.max_epochs == 7 and trainer.callbacks == []
    assert set(skipped) == {"lightning.pytorch.callbacks.LearningRateMonitor", "lightning.pytorch.loggers.TensorBoardLogger",
                            "trainer.strategy=ddp_find_unused_parameters_true"}
    with pytest.raises(KeyError, match="no viscy_amd counterpart"):
        C.instantiate({"class_path": "viscy_models.unet.Unet25d", "init_args": {}})


def test_mmap_preload_with_unequal_time_axes(tmp_path):
    """hcs.py:351-378: FOVs with different T occupy their own slabs of the staged buffer (cumulative offsets)"""
    rng = np.random.default_rng(5)
    pos = {"A/1/0": rng.random((3, 2, 4, 32, 32), dtype=np.float32), "A/2/0": rng.random((1, 2, 4, 32, 32), dtype=np.float32),
           "B/1/0": rng.random((2, 2, 4, 32, 32), dtype=np.float32)}
    path = str(tmp_path / "t.zarr")
    write_hcs_plate(path, pos, ["Phase3D", "Nuclei"])
    kw = dict(z_window_size=2, batch_size=1, num_workers=0, split_ratio=0.67, yx_patch_size=(32, 32))
    plain = HCSDataModule(path, "Phase3D", "Nuclei", **kw)
    plain.setup("fit")
    mm = HCSDataModule(path, "Phase3D", "Nuclei", mmap_preload=True, scratch_dir=tmp_path / "s", **kw)
    mm.prepare_data()
    mm.setup("fit")
    assert HCSDataModule._fov_t_offsets([p for _, p in open_ome_zarr(path).positions()], "0") == [0, 3, 4, 6]
    n = len(plain.train_dataset)
    assert n == len(mm.train_dataset) and n > 0
    for i in range(n):
        a, b = plain.train_dataset[i], mm.train_dataset[i]
        assert a["index"] == b["index"] and torch.equal(a["source"], b["source"]) and torch.equal(a["target"], b["target"])


# ---------------------------------------------------------------- foreground masks / non-zero rejection sampling
@pytest.fixture(scope="module")
def sampling_plate():
    from tests.conftest import build_sampling_plate

    path = os.path.join(tempfile.mkdtemp(), "sampling.zarr")
    pos, ch = build_sampling_plate(path)
    return path, pos, ch


@pytest.mark.parametrize("tag", ["plain", "two_targets_masks", "reject_intensity", "reject_mask_channel", "reject_source_channel"])
def test_sliding_window_sampling_matches_the_reference_run(sampling_plate, tag):
    """tests/golden/hcs_sampling.pt = what the REFERENCE's SlidingWindowDataset returned over this plate (G11,
    oracle/validate_against_reference.py): same (path, t, z) per draw incl. the rejection re-draws, same image / mask sums,
    per-timepoint statistics resolved (sliding_window.py:148-164, 222-252; foreground_masks.py)"""
    path, _, _ = sampling_plate
    case = load_golden("hcs_sampling.pt")["cases"][tag]
    positions = [p for _, p in open_ome_zarr(path).positions()]
    ds = SlidingWindowDataset(positions, **case["kwargs"])
    torch.manual_seed(case["seed"])
    for i, want in zip(case["order"], case["samples"]):
        s = ds[i]
        assert s["index"] == want["index"]
        assert abs(s["source"].double().sum().item() - want["source_sum"]) < 1e-9
        assert abs(s["target"].double().sum().item() - want["target_sum"]) < 1e-9
        assert ("fg_mask" in s) == (want["fg_sum"] is not None)
        if "fg_mask" in s:
            assert s["fg_mask"].shape == s["target"].shape and s["fg_mask"].double().sum().item() == want["fg_sum"]
            assert set(s["fg_mask"].unique().tolist()) <= {0.0, 1.0}
        assert float(s["norm_meta"]["Phase"]["timepoint_statistics"]["mean"]) == pytest.approx(want["tp_mean"])


def test_fg_mask_key_errors_and_mask_layouts(sampling_plate, tmp_path):
    from viscy_amd.data.hcs import ForegroundMaskSupport

    path, pos, ch = sampling_plate
    positions = [p for _, p in open_ome_zarr(path).positions()]
    assert "fg_mask" not in SlidingWindowDataset(positions, {"source": ["Phase"], "target": ["Nuclei"]}, 3)[0]
    with pytest.raises(FileNotFoundError, match="fg_mask"):
        SlidingWindowDataset(positions, {"source": ["Phase"], "target": ["Nuclei"]}, 3, fg_mask_key="no_such_array")
    with pytest.raises(ValueError, match="min_nonzero_fraction"):
        SlidingWindowDataset(positions, {"source": ["Phase"], "target": ["Nuclei"]}, 3, min_nonzero_fraction=1.5)
    with pytest.raises(ValueError, match="nonzero_channel"):
        SlidingWindowDataset(positions, {"source": ["Phase"], "target": ["Nuclei"]}, 3, min_nonzero_fraction=0.5, nonzero_channel="DAPI")
    # full-channel vs target-only mask arrays (foreground_masks.py:66-109)
    assert ForegroundMaskSupport.resolve_mask_ch_indices(3, 3, 2, [1, 2]) == [1, 2]
    assert ForegroundMaskSupport.resolve_mask_ch_indices(2, 3, 2, [1, 2]) == [0, 1]
    with pytest.raises(ValueError, match="expected 3"):
        ForegroundMaskSupport.resolve_mask_ch_indices(4, 3, 2, [1, 2])


def test_spatial_transforms_carry_the_masks_and_intensity_transforms_do_not():
    """viscy-data tests/test_hcs.py:688-758 on this package's transform classes"""
    from viscy_amd.transforms import (BatchedCenterSpatialCropd, BatchedRandAdjustContrastd, BatchedRandAffined, BatchedRandFlipd,
                                      RandWeightedCropd)

    spatial = BatchedRandAffined(keys=["Phase", "Fluorescence"], prob=0.5, rotate_range=[0.1, 0.0, 0.0])
    intensity = BatchedRandAdjustContrastd(keys=["Phase", "Fluorescence"], prob=0.5)
    for _ in range(2):  # idempotent
        HCSDataModule._inject_mask_keys([spatial, intensity], ("Fluorescence",), ("__fg_mask_Fluorescence",))
    assert list(spatial.keys).count("__fg_mask_Fluorescence") == 1 and spatial.allow_missing_keys is True
    assert "__fg_mask_Fluorescence" not in intensity.keys
    # GPU-side flip + centre crop keep the batched mask pixel-aligned with the target
    B, C, D, H, W = 2, 1, 4, 16, 16
    target = torch.zeros(B, C, D, H, W)
    target[:, :, :, : H // 2, : W // 4] = 1.0
    flip, crop = BatchedRandFlipd(keys=["target"], prob=1.0, spatial_axes=[1]), BatchedCenterSpatialCropd(keys=["target"], roi_size=(2, 12, 12))
    HCSDataModule._inject_mask_keys([flip, crop], ("target",), ("fg_mask",))
    out = crop(flip({"target": target.clone(), "fg_mask": (target > 0).float()}))
    assert out["target"].shape == (B, C, 2, 12, 12) and torch.equal((out["target"] > 0).float(), out["fg_mask"])
    assert not torch.equal(out["target"], crop({"target": target.clone()})["target"])  # the flip did move the pattern
    out = crop(flip({"target": target.clone()}))  # allow_missing_keys: batches without masks pass through
    assert "fg_mask" not in out
    # CPU-side multi-sample crop (the fit recipes' RandWeightedCropd) crops the per-channel temp key with the target
    t2 = torch.zeros(1, 8, 32, 32)
    t2[:, :, 8:24, 8:24] = torch.rand(1, 8, 16, 16) + 0.5
    wc = RandWeightedCropd(keys=["Nuclei"], w_key="Nuclei", spatial_size=(4, 8, 8), num_samples=3)
    HCSDataModule._inject_mask_keys([wc], ("Nuclei",), ("__fg_mask_Nuclei",))
    for s in wc({"Nuclei": t2, "__fg_mask_Nuclei": (t2 > 0).float()}):
        assert s["Nuclei"].shape == (1, 4, 8, 8) and torch.equal((s["Nuclei"] > 0).float(), s["__fg_mask_Nuclei"])


@pytest.mark.parametrize("mmap", [False, True], ids=["zarr", "mmap"])
def test_datamodule_serves_aligned_masks_and_rejects_empty_windows(sampling_plate, tmp_path, mmap):
    """HCSDataModule(fg_mask_key=..., min_nonzero_fraction=...) (hcs.py:124-154, 466-476, 776-783): the batch carries
    ``fg_mask`` aligned with ``target`` after CPU crop + GPU flip / crop, ``target_2d`` slices both, the rejection filter is
    a training-set setting only, the mask buffer is staged next to the data buffer under mmap_preload"""
    from viscy_amd.transforms import BatchedCenterSpatialCropd, BatchedRandFlipd, RandWeightedCropd

    path, pos, ch = sampling_plate
    dm = HCSDataModule(path, "Phase", ["Membrane", "Nuclei"], z_window_size=4, batch_size=4, num_workers=0, yx_patch_size=(8, 8),
                       split_ratio=0.67, fg_mask_key="fg_mask", min_nonzero_fraction=0.3, nonzero_channel="Nuclei",
                       max_nonzero_retries=50, target_2d=True, mmap_preload=mmap, scratch_dir=tmp_path,
                       augmentations=[RandWeightedCropd(keys=["Phase", "Membrane", "Nuclei"], w_key="Membrane", spatial_size=(4, 12, 12),
                                                        num_samples=2)],
                       gpu_augmentations=[BatchedRandFlipd(keys=["source", "target"], prob=1.0, spatial_axes=[2]),
                                          BatchedCenterSpatialCropd(keys=["source", "target"], roi_size=(4, 8, 8))])
    dm.prepare_data()
    if mmap:
        assert (dm._mmap_cache_dir / "fg_mask.mmap").exists()
        (dm._mmap_cache_dir / "fg_mask.mmap").unlink()   # marker present, buffer cleaned up: rebuilt (hcs.py:515-545)
        dm.prepare_data()
        assert (dm._mmap_cache_dir / "fg_mask.mmap").exists()
    dm.setup("fit")
    assert dm.train_dataset.min_nonzero_fraction == 0.3 and dm.val_dataset.min_nonzero_fraction == 0.0
    assert dm.train_dataset.fg_mask_support is not None and dm.val_dataset.fg_mask_support is not None
    torch.manual_seed(0)
    seen = 0
    for batch in dm.train_dataloader():
        assert batch["fg_mask"].shape == (4, 2, 4, 12, 12)
        # every stack the loader kept has >= 30 % foreground in the Nuclei mask of its full window (before the crop)
        for pth, t, z in zip(batch["index"][0], batch["index"][1].tolist(), batch["index"][2].tolist()):
            win = pos[pth.strip("/").rsplit("/", 1)[0]][t, 2, z:z + 4]
            assert (win > 0.5).mean() >= 0.3
        dm.training = True
        out = dm.on_after_batch_transfer(batch, 0)
        assert out["source"].shape == (4, 1, 4, 8, 8) and out["target"].shape == (4, 2, 1, 8, 8) == out["fg_mask"].shape
        assert torch.equal((out["target"] > 0.5).float(), out["fg_mask"])
        seen += 1
    assert seen >= 1
    dm.setup("predict")
    assert dm.predict_dataset.fg_mask_support is None  # hcs.py:669-671
    with pytest.raises(NotImplementedError, match="ground_truth_masks"):
        HCSDataModule(path, "Phase", "Nuclei", 4, ground_truth_masks=tmp_path)


def test_task_list_hazards_are_tracked_by_byte_range(monkeypatch):
    """ops._queue (the weight-space task lists, one concurrent launch per list): a job that reads or writes memory an already queued
    job writes, or writes memory an already queued job reads — through ANY view, not only the same start address — launches the
    list collected so far first; the bookkeeping is cleared also when that launch fails."""
    from viscy_amd import _lib as L
    from viscy_amd import ops

    flushed = []

    def fake_flush():
        flushed.append(len(ops._BATCH))
        del ops._BATCH[:], ops._BATCH_KEEP[:], ops._BATCH_WRITTEN[:], ops._BATCH_READ[:]

    monkeypatch.setattr(ops, "flush", fake_flush)
    monkeypatch.setattr(ops, "ptr", lambda t: None if t is None else t.data_ptr())
    monkeypatch.setattr(ops, "_BATCH", [])
    flat = torch.zeros(1024)
    a, b, c = flat[0:256], flat[128:384], flat[512:768]   # a and b overlap, with different start addresses
    src = torch.zeros(256)
    assert ops._span(flat[0:256].view(16, 16).t()) == (flat.data_ptr(), flat.data_ptr() + 256 * 4)
    assert ops._queue(L.WTASK_TRANSPOSE, 0, (16, 16, 1), src, a, None, None) and flushed == []
    assert ops._queue(L.WTASK_TRANSPOSE, 0, (16, 16, 1), src, c, None, None) and flushed == []       # disjoint output: same list
    assert ops._queue(L.WTASK_TRANSPOSE, 0, (16, 16, 1), src, b, None, None) and flushed == [2]      # write overlaps a queued write
    assert ops._queue(L.WTASK_MATVEC, 0, (16, 16), c, torch.zeros(16), None, src) and flushed == [2]  # reads c: nothing queued writes it now
    assert ops._queue(L.WTASK_MATVEC, 0, (16, 16), a, torch.zeros(16), None, src) and flushed == [2, 2]  # a overlaps b, which is being written
    assert ops._queue(L.WTASK_TRANSPOSE, 0, (16, 16, 1), b, src, None, None) and flushed == [2, 2, 1]    # writes (src) what a queued job reads
    # the real flush clears its bookkeeping when the launch raises
    monkeypatch.undo()
    monkeypatch.setattr(ops, "_BATCH", [L.VsxWTask()])
    ops._BATCH_WRITTEN.append((0, 1)), ops._BATCH_READ.append((0, 1)), ops._BATCH_KEEP.append(None)

    class Boom:
        def vsx_weight_tasks(self, *a):
            raise RuntimeError("launch failed")

    monkeypatch.setattr(ops, "lib", lambda: Boom())
    monkeypatch.setattr(ops, "stream", lambda: None)
    with pytest.raises(RuntimeError):
        ops.flush()
    assert ops._BATCH_WRITTEN == [] and ops._BATCH_READ == [] and ops._BATCH_KEEP == [] and ops._BATCH == []
