import importlib as _il

_m = _il.import_module('tf-keras-deeplabv3p-model-set_amd.model')
get_deeplabv3p_model = _m.get_deeplabv3p_model
deeplab_model_map = _m.deeplab_model_map
