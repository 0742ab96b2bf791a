"""MI355X-native drop-in for the conv-autoencoder + MLP training path of the reference notebook.

Public surface mirrors the notebook: ``Encoder``, ``Decoder``,
``SupervisedAutoencoder``, ``MLP``, ``extract_features`` plus the ``fit``/``evaluate`` entry points
that restate its inline loops.  All arithmetic runs in hand-written gfx950 HIP kernels behind the
C ABI declared in ``include/eae.h``.
"""
from .modules import Encoder, Decoder, SupervisedAutoencoder, MLP  # noqa: F401
from .augment import augment_batch  # noqa: F401
from .train import (fit_autoencoder, grid_search_autoencoder, extract_features, fit_mlp, grid_search_mlp,  # noqa: F401
                    evaluate)
from .report import confusion_matrix, classification_report, loss_heatmap  # noqa: F401
from .probe import ce_mse_ratio_probe  # noqa: F401

__all__ = ["Encoder", "Decoder", "SupervisedAutoencoder", "MLP", "fit_autoencoder", "grid_search_autoencoder",
           "extract_features", "fit_mlp", "grid_search_mlp", "evaluate", "augment_batch",
           "confusion_matrix", "classification_report", "loss_heatmap", "ce_mse_ratio_probe"]
