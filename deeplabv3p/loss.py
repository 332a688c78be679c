import importlib as _il

SparseCategoricalCrossEntropy = _il.import_module('tf-keras-deeplabv3p-model-set_amd.model').SparseCategoricalCrossEntropy
