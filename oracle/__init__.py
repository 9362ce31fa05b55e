"""ORACLE — test infrastructure, not product code.

CPU restatements (plain torch, fp32) of the reference algorithms on the accelerated path, each citing the reference
file:line it follows:

  unext2_ref.py      UNeXt2 forward (viscy_models/unet/unext2.py + components/{stems,blocks,heads}.py; timm 1.0.27 ConvNeXt-V2
                     and MONAI 1.5.2 UpSample / Convolution internals restated from their published algorithms)
  fcmae_ref.py       FullyConvolutionalMAE dense and masked paths (viscy_models/unet/fcmae.py), MaskedMSELoss (cytoland/engine.py)
  contrastive_ref.py DynaCLR: StemDepthtoChannels, ContrastiveEncoder (viscy_models/contrastive/encoder.py), NTXentLoss /
                     NTXentHCL (contrastive/loss.py); timm ConvNeXt-V1 / classifier head and pytorch-metric-learning's pair
                     semantics restated from their published algorithms
  loss_ref.py        MixedLoss / ms_ssim_25d (viscy_utils/losses/mixed_loss.py, evaluation/metrics.py)
  transforms_ref.py  viscy_transforms normalisation / augmentation arithmetic with injected random parameters

Pinning: ``validate_against_reference.py`` (runs only where /root/reference exists) checks every restatement against the
reference's own code — imported directly or executed on stub third-party modules — and writes the golden fixtures under
``tests/golden/`` (data only: inputs, seeds, expected outputs).  Parity status: pinned (G1-G3, G6, G6b, G8, G9, G10) except the
internals of timm / MONAI / kornia / pytorch-metric-learning, which are absent from /root/reference ("parity unpinned" for exactly those pieces; they
are cross-checked against `transformers`' ConvNeXt / ConvNeXt-V2, the SimCLR cross-entropy form and the in-repo restatement fcmae.py:174-221).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; nothing under
``viscy_amd/`` does (tests/test_abi_cpu.py::test_product_never_imports_oracle).
"""
