"""Mixed-precision policy, the counterpart of what the reference's train.py:37-46 does with
`tensorflow.keras.mixed_precision`: set a global policy BEFORE the model is built and every layer computes in the
policy's compute dtype while the variables stay float32.

    from mixed_precision import Policy, set_policy        # train.py:41-44
    set_policy(Policy('mixed_bfloat16'))
    model = get_deeplabv3p_model(...)

On MI355X the 16-bit compute type is bfloat16 (BASELINE.json configs[4]: "mixed-precision bf16"): activations and their
gradients are stored as bf16, the matrix cores take bf16 operands and accumulate in fp32, BatchNorm statistics, softmax /
loss, parameter gradients, optimiser state and the master weights are fp32 -- so no loss scaling is needed and
'mixed_float16' (which the reference names, with its dynamic loss scale) is accepted as an alias of the same bf16 path.
"""
_GLOBAL = 'float32'
_NAMES = {'float32': 'float32', 'mixed_bfloat16': 'mixed_bfloat16', 'mixed_float16': 'mixed_bfloat16'}


class Policy:
    def __init__(self, name):
        if name not in _NAMES:
            raise ValueError('Unknown mixed-precision policy %r (float32, mixed_bfloat16, mixed_float16)' % (name,))
        self.name = name

    @property
    def compute_dtype(self):
        return 'float32' if _NAMES[self.name] == 'float32' else 'bfloat16'

    @property
    def variable_dtype(self):
        return 'float32'


def set_global_policy(policy):
    global _GLOBAL
    name = policy.name if isinstance(policy, Policy) else policy
    if name not in _NAMES:
        raise ValueError('Unknown mixed-precision policy %r' % (name,))
    if name == 'mixed_float16':
        import warnings
        warnings.warn("policy 'mixed_float16' (train.py:41) runs as mixed_bfloat16 on MI355X: bf16 storage, fp32 accumulation, "
                      "no loss scaling", stacklevel=2)
    _GLOBAL = name


set_policy = set_global_policy          # the `experimental` spelling train.py:44 uses


def global_policy():
    return Policy(_GLOBAL)


def is_bf16(policy=None):
    name = (policy.name if isinstance(policy, Policy) else policy) if policy is not None else _GLOBAL
    return _NAMES[name] == 'mixed_bfloat16'
