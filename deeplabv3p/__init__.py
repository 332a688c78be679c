"""Import-path shim: `from deeplabv3p.model import get_deeplabv3p_model` works as in the reference
(train.py:13, deeplab.py:17); everything lives in tf-keras-deeplabv3p-model-set_amd/."""
