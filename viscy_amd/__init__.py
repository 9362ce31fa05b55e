"""viscy_amd — MI355X-native UNeXt2 virtual-staining hot path (drop-in for the VisCy / Cytoland classes).

    from viscy_amd import UNeXt2, MixedLoss, VSUNet, HCSDataModule

See DESIGN.md (what is built and why) and INTEGRATION.md (how it binds to the reference).
"""

__all__ = ["UNeXt2", "MixedLoss", "VSUNet", "HCSDataModule", "HCSPredictionWriter", "FcmaeUNet", "FullyConvolutionalMAE", "MaskedMSELoss", "ContrastiveEncoder", "ContrastiveModule", "NTXentLoss", "NTXentHCL", "FlatAdamW", "FlatDataParallel", "TrainStep"]


def __getattr__(name):
    if name == "UNeXt2":
        from .unext2 import UNeXt2 as v
    elif name == "MixedLoss":
        from .losses import MixedLoss as v
    elif name == "FcmaeUNet":
        from .vsunet import FcmaeUNet as v
    elif name == "FullyConvolutionalMAE":
        from .fcmae import FullyConvolutionalMAE as v
    elif name == "MaskedMSELoss":
        from .losses import MaskedMSELoss as v
    elif name in ("ContrastiveEncoder", "ContrastiveModule", "NTXentLoss", "NTXentHCL"):
        from . import contrastive as _c

        v = getattr(_c, name)
    elif name == "VSUNet":
        from .vsunet import VSUNet as v
    elif name == "HCSDataModule":
        from .data import HCSDataModule as v
    elif name == "HCSPredictionWriter":
        from .prediction_writer import HCSPredictionWriter as v
    elif name == "FlatAdamW":
        from .optim import FlatAdamW as v
    elif name == "FlatDataParallel":
        from .parallel import FlatDataParallel as v
    elif name == "TrainStep":
        from .step import TrainStep as v
    else:
        raise AttributeError(name)
    return v
