"""MI355X-native DeepLabV3+ forward/backward path behind the reference's get_deeplabv3p_model() API.

The directory name carries the reference repo's name (with hyphens), so import it through
importlib:  importlib.import_module('tf-keras-deeplabv3p-model-set_amd')  -- or use the
`deeplabv3p` shim package at the repo root, which mirrors the reference's import paths
(`from deeplabv3p.model import get_deeplabv3p_model`).
"""
from .model import (get_deeplabv3p_model, deeplab_model_map, DeeplabModel, SGD, Adam, RMSprop, get_optimizer,  # noqa: F401
                    SparseCategoricalCrossEntropy, WeightedSparseCategoricalCrossEntropy, SparseSoftmaxFocalLoss,
                    miou_from_confusion, Jaccard, jaccard_from_counts, EvalCallBack)

from . import mixed_precision  # noqa: F401,E402
from . import watchdog  # noqa: F401,E402

__all__ = ['get_deeplabv3p_model', 'deeplab_model_map', 'DeeplabModel', 'SGD', 'Adam', 'RMSprop', 'get_optimizer',
           'SparseCategoricalCrossEntropy', 'WeightedSparseCategoricalCrossEntropy', 'SparseSoftmaxFocalLoss',
           'miou_from_confusion', 'Jaccard', 'jaccard_from_counts', 'EvalCallBack', 'mixed_precision']
