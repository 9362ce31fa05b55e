"""TEST INFRASTRUCTURE — CPU restatement (plain torch, fp32) of the DynaCLR contrastive path:

  * ``StemDepthtoChannels``  /root/reference/packages/viscy-models/src/viscy_models/components/stems.py:53-134
  * ``ContrastiveEncoder``   /root/reference/packages/viscy-models/src/viscy_models/contrastive/encoder.py:52-154
    (timm ConvNeXt classifier trunk with its patchify convolution removed, global-average-pool head with ``fc`` removed,
    Linear-BN-ReLU-Linear-BN projection)
  * ``NTXentLoss`` / ``NTXentHCL``  .../contrastive/loss.py:20-186
  * the NT-Xent branch of ``ContrastiveModule.training_step``  /root/reference/applications/dynaclr/src/dynaclr/engine.py:262-287

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.

Third-party arithmetic that is NOT under /root/reference and not installed here, restated from the published sources:
  * timm 1.0.27 ``ConvNeXt`` (``convnext_tiny``: depths (3,3,9,3), dims (96,192,384,768), ``ls_init_value=1e-6`` layer scale
    ``gamma``, plain ``Mlp``; ``convnextv2_tiny``: GRN, no layer scale), ``NormMlpClassifierHead`` (global avg pool ->
    ``LayerNorm2d`` -> flatten -> fc); key names ``stem.{0,1}``, ``stages.i.{downsample.{0,1},blocks.j.{conv_dw,norm,mlp.fc1,
    mlp.fc2[,mlp.grn]}}`` (+ ``blocks.j.gamma``), ``head.norm``, ``head.fc``.
  * pytorch-metric-learning ``NTXentLoss`` (``GenericPairLoss`` with ``CosineSimilarity``, ``MeanReducer``): every ordered
    pair of distinct samples with equal labels is a positive pair (a, p); the negatives of a are all samples with a
    different label;  loss(a,p) = -log( e^{s_ap/T} / (e^{s_ap/T} + sum_n e^{s_an/T}) + tiny ),  mean over positive pairs.
Pinning (oracle/validate_against_reference.py G10): the stem is the reference's own class (imported directly, exact);
the reference's own ``encoder.py`` runs unchanged on a stub ``timm.create_model`` that returns this file's trunk
(wiring: stem swap, ``head.fc`` removal, projection; exact incl. state-dict keys); the ConvNeXt-V1 block equals
``transformers``' independent ``ConvNextLayer`` (1e-6); the reference's own ``NTXentHCL._compute_loss`` (beta > 0) runs on a
stub ``pytorch_metric_learning`` base class built from ``PairLossBase`` below == ``NTXentHCL`` here (exact).  The pml base
semantics themselves (pair mining, reducer) are **parity unpinned** — cross-checked against the textbook SimCLR
cross-entropy form in tests/test_oracle.py.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .unext2_ref import ConvNeXtBlock as ConvNeXtV2Block
from .unext2_ref import LayerNorm2d


# ------------------------------------------------------------------------------------------------ encoder
class StemDepthtoChannels(nn.Module):
    """stems.py:53-134: Conv3d(kernel, stride) then fold the remaining depth into channels."""

    def __init__(self, in_channels: int, in_stack_depth: int, in_channels_encoder: int, stem_kernel_size=(5, 4, 4),
                 stem_stride=(5, 4, 4)):
        super().__init__()
        out_depth = (in_stack_depth - stem_kernel_size[0]) // stem_stride[0] + 1
        out_ch = in_channels_encoder // out_depth
        mismatch = in_channels_encoder - out_depth * out_ch
        if mismatch != 0:
            raise ValueError(f"Stem needs to output {mismatch} more channels to match the encoder. Adjust the in_stack_depth.")
        self.conv = nn.Conv3d(in_channels, out_ch, kernel_size=stem_kernel_size, stride=stem_stride)

    def forward(self, x: Tensor) -> Tensor:
        x = self.conv(x)
        b, c, d, h, w = x.shape
        return x.reshape(b, c * d, h, w)


class Mlp(nn.Module):
    """timm.layers.Mlp (Linear): fc1 -> GELU -> fc2"""

    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x: Tensor) -> Tensor:
        return self.fc2(self.act(self.fc1(x)))


class ConvNeXtV1Block(nn.Module):
    """timm ConvNeXtBlock(ls_init_value=1e-6, use_grn=False, conv_mlp=False): x + gamma * mlp(LN(dwconv7(x)))"""

    def __init__(self, dim: int, ls_init_value: float = 1e-6):
        super().__init__()
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim))
        self.conv_dw = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, 4 * dim)

    def forward(self, x: Tensor) -> Tensor:
        shortcut = x
        x = self.conv_dw(x).permute(0, 2, 3, 1)
        x = self.mlp(self.norm(x)).permute(0, 3, 1, 2)
        return x * self.gamma.reshape(1, -1, 1, 1) + shortcut


class _Stage(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int, depth: int, v2: bool):
        super().__init__()
        if cin != cout or stride > 1:
            k = 2 if stride > 1 else 1
            self.downsample = nn.Sequential(LayerNorm2d(cin), nn.Conv2d(cin, cout, k, stride=stride))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[(ConvNeXtV2Block(cout, conv_mlp=False) if v2 else ConvNeXtV1Block(cout)) for _ in range(depth)])

    def forward(self, x: Tensor) -> Tensor:
        return self.blocks(self.downsample(x))


class _Head(nn.Module):
    """timm NormMlpClassifierHead(hidden_size=None): global avg pool -> LayerNorm2d -> flatten -> (drop) -> fc"""

    def __init__(self, dim: int, num_classes: int):
        super().__init__()
        self.norm = LayerNorm2d(dim)
        self.fc = nn.Linear(dim, num_classes) if num_classes > 0 else nn.Identity()

    def forward(self, x: Tensor) -> Tensor:
        x = x.mean((2, 3), keepdim=True)
        return self.fc(self.norm(x).flatten(1))


class ConvNeXt(nn.Module):
    """What ``timm.create_model("convnext_tiny" | "convnextv2_tiny", features_only=False, num_classes=n)`` builds."""

    def __init__(self, backbone: str, num_classes: int, in_chans: int = 3, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768)):
        super().__init__()
        if backbone not in ("convnext_tiny", "convnextv2_tiny"):
            raise NotImplementedError(backbone)
        v2 = backbone.startswith("convnextv2")
        self.num_features = dims[-1]
        self.stem = nn.Sequential(nn.Conv2d(in_chans, dims[0], 4, stride=4), LayerNorm2d(dims[0]))
        stages, prev = [], dims[0]
        for i, (d, c) in enumerate(zip(depths, dims)):
            stages.append(_Stage(prev, c, 2 if i > 0 else 1, d, v2))
            prev = c
        self.stages = nn.Sequential(*stages)
        self.norm_pre = nn.Identity()
        self.head = _Head(dims[-1], num_classes)
        self.apply(_timm_init)

    def forward(self, x: Tensor) -> Tensor:
        return self.head(self.norm_pre(self.stages(self.stem(x))))


def _timm_init(m: nn.Module) -> None:
    """timm.models.convnext._init_weights"""
    if isinstance(m, nn.Conv2d):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        nn.init.zeros_(m.bias)


class ContrastiveEncoder(nn.Module):
    """encoder.py:52-154"""

    def __init__(self, backbone: str, in_channels: int, in_stack_depth: int, stem_kernel_size=(5, 4, 4), stem_stride=(5, 4, 4),
                 embedding_dim: int = 768, projection_dim: int = 128, drop_path_rate: float = 0.0, pretrained: bool = False,
                 depths=(3, 3, 9, 3), dims=(96, 192, 384, 768)):
        super().__init__()
        assert drop_path_rate == 0.0 and not pretrained
        self.backbone = backbone
        encoder = ConvNeXt(backbone, num_classes=embedding_dim, depths=depths, dims=dims)
        in_channels_encoder = encoder.stem[0].out_channels
        encoder.stem[0] = nn.Identity()
        projection = nn.Sequential(
            nn.Linear(encoder.num_features, embedding_dim), nn.BatchNorm1d(embedding_dim), nn.ReLU(inplace=True),
            nn.Linear(embedding_dim, projection_dim), nn.BatchNorm1d(projection_dim))
        encoder.head.fc = nn.Identity()
        self.stem = StemDepthtoChannels(in_channels, in_stack_depth, in_channels_encoder, stem_kernel_size, stem_stride)
        self.encoder = encoder
        self.projection = projection

    def forward(self, x: Tensor):
        embedding = self.encoder(self.stem(x))
        return embedding, self.projection(embedding)


# ------------------------------------------------------------------------------------------------ losses
def cosine_anneal(start: float, end: float, epoch: int, total: int) -> float:
    """viscy_models/schedule.py: cosine interpolation start -> end over ``total`` epochs, ``end`` afterwards"""
    if epoch >= total:
        return end
    return end + 0.5 * (start - end) * (1 + math.cos(math.pi * epoch / total))


class _Cosine:
    is_inverted = True


class PairLossBase(nn.Module):
    """pytorch-metric-learning ``NTXentLoss`` as used here (see module docstring): labels -> all (anchor, positive) and
    (anchor, negative) index pairs, cosine-similarity matrix, ``_compute_loss`` hook, mean over positive pairs."""

    def __init__(self, temperature: float = 0.07, **kwargs):
        super().__init__()
        self.temperature = temperature
        self.distance = _Cosine()

    def add_to_recordable_attributes(self, **kw):  # pml bookkeeping, no arithmetic
        pass

    def zero_losses(self):
        return None

    @staticmethod
    def pairs(labels: Tensor):
        same = labels.unsqueeze(1) == labels.unsqueeze(0)
        diff = ~same
        same = same.clone()
        same.fill_diagonal_(False)
        a1, p = torch.where(same)
        a2, n = torch.where(diff)
        return a1, p, a2, n

    def forward(self, embeddings: Tensor, labels: Tensor) -> Tensor:
        e = F.normalize(embeddings, p=2, dim=1)
        mat = e @ e.t()
        a1, p, a2, n = self.pairs(labels)
        out = self._compute_loss(mat[a1, p], mat[a2, n], (a1, p, a2, n))
        if out is None:
            return embeddings.sum() * 0
        return out["loss"]["losses"].mean()

    def _compute_loss(self, pos_pairs, neg_pairs, indices_tuple):
        a1, p, a2, _ = indices_tuple
        if len(a1) == 0 or len(a2) == 0:
            return self.zero_losses()
        dtype = neg_pairs.dtype
        pos = pos_pairs.unsqueeze(1) / self.temperature
        neg = neg_pairs / self.temperature
        n_per_p = (a2.unsqueeze(0) == a1.unsqueeze(1)).to(dtype)
        neg = neg * n_per_p
        neg[n_per_p == 0] = torch.finfo(dtype).min
        max_val = torch.max(pos, neg.max(dim=1, keepdim=True)[0]).detach()
        num = torch.exp(pos - max_val).squeeze(1)
        den = torch.exp(neg - max_val).sum(dim=1) + num
        return {"loss": {"losses": -torch.log(num / den + torch.finfo(dtype).tiny), "indices": (a1, p), "reduction_type": "pos_pair"}}


class NTXentLoss(PairLossBase):
    """loss.py:20-73"""

    def __init__(self, temperature: float = 0.07, temperature_schedule: str = "constant", temperature_start: float = 0.1,
                 temperature_warmup_epochs: int = 50, **kwargs):
        super().__init__(temperature=temperature, **kwargs)
        self.temperature_schedule, self.temperature_start = temperature_schedule, temperature_start
        self.temperature_end, self.temperature_warmup_epochs = temperature, temperature_warmup_epochs

    def step(self, epoch: int) -> None:
        if self.temperature_schedule == "cosine":
            self.temperature = cosine_anneal(self.temperature_start, self.temperature_end, epoch, self.temperature_warmup_epochs)


class NTXentHCL(NTXentLoss):
    """loss.py:76-186: negatives re-weighted by exp(beta * sim), weights normalised to sum to the negative count."""

    def __init__(self, temperature: float = 0.07, beta: float = 0.5, **kwargs):
        super().__init__(temperature=temperature, **kwargs)
        self.beta = beta

    def _compute_loss(self, pos_pairs, neg_pairs, indices_tuple):
        if self.beta == 0.0:
            return super()._compute_loss(pos_pairs, neg_pairs, indices_tuple)
        a1, p, a2, _ = indices_tuple
        if len(a1) == 0 or len(a2) == 0:
            return self.zero_losses()
        dtype = neg_pairs.dtype
        pos = pos_pairs.unsqueeze(1) / self.temperature
        negs = neg_pairs / self.temperature
        n_per_p = (a2.unsqueeze(0) == a1.unsqueeze(1)).to(dtype)
        w = torch.exp(self.beta * neg_pairs) * n_per_p
        w = w * n_per_p.sum(1, keepdim=True) / w.sum(1, keepdim=True).clamp(min=1e-8)
        negm = negs * n_per_p
        negm[n_per_p == 0] = torch.finfo(dtype).min
        max_val = torch.max(pos, negm.max(dim=1, keepdim=True)[0]).detach()
        num = torch.exp(pos - max_val).squeeze(1)
        den = (w * torch.exp(negm - max_val)).sum(1) + num
        return {"loss": {"losses": -torch.log(num / den + torch.finfo(dtype).tiny), "indices": (a1, p), "reduction_type": "pos_pair"}}


def contrastive_step_loss(model: ContrastiveEncoder, loss_fn: PairLossBase, anchor: Tensor, positive: Tensor) -> Tensor:
    """dynaclr/engine.py:262-275: two separate forwards (BatchNorm statistics per call), labels = arange twice"""
    _, pa = model(anchor)
    _, pp = model(positive)
    idx = torch.arange(pa.shape[0], device=pa.device)
    return loss_fn(torch.cat((pa, pp)), torch.cat((idx, idx)))


def randomize_encoder_(model: nn.Module, seed: int) -> nn.Module:
    """deterministic non-trivial state for parity tests: every parameter random (GRN included), layer scale away from its 1e-6
    init so the branch matters, BatchNorm running statistics away from (0, 1)"""
    from .unext2_ref import randomize_

    randomize_(model, seed=seed)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.endswith(".gamma"):
                p_.copy_(0.5 + 0.2 * torch.randn(p_.shape, generator=g))
        for n_, b_ in model.named_buffers():
            if n_.endswith("running_var"):
                b_.copy_(0.5 + torch.rand(b_.shape, generator=g))
            elif n_.endswith("running_mean"):
                b_.copy_(0.1 * torch.randn(b_.shape, generator=g))
    return model
