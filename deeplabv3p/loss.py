"""`from deeplabv3p.loss import ...` as train.py:15 does"""
import importlib as _il

_m = _il.import_module('tf-keras-deeplabv3p-model-set_amd.model')
SparseCategoricalCrossEntropy = _m.SparseCategoricalCrossEntropy
WeightedSparseCategoricalCrossEntropy = _m.WeightedSparseCategoricalCrossEntropy
SparseSoftmaxFocalLoss = _m.SparseSoftmaxFocalLoss
