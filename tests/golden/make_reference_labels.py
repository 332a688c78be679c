"""Label fixtures from the reference's own example data (/root/reference/example/*.png, the two VOC2012
annotations its README demos use): the decoded class-id maps, as data, with the label tail of
SegmentationGenerator.__getitem__ applied (deeplabv3p/data.py:116-121: int32, flatten, label > num_classes-1 -> 255).
Decoded with PIL exactly as the reference reads them (deeplabv3p/data.py:74 `Image.open(label_path)` -> np.array).

The known answers that go with them (tests/test_golden.py) are ANALYTIC consequences of deeplabv3p/loss.py:121-156,
not outputs of this repo's oracle:
  * all-equal logits: p = 1/C for every class, so loss = ln(C) * (#labelled pixels) / (#all pixels) -- the mean runs
    over ALL entries, ignored ones included (Keras reduction), and 255 rows contribute 0;
  * logits = 20 * onehot(label): p_y = 1/(1 + (C-1) e^-20) is above 1 - 1e-7, so it is CLIPPED to 1 - 1e-7
    (loss.py:150 via K.categorical_crossentropy's epsilon) and loss = -ln(1 - 1e-7) * (#labelled) / (#all).

    python tests/golden/make_reference_labels.py
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/example'

if __name__ == '__main__':
    out = {}
    for name in ('2007_000039', '2007_000346'):
        lab = np.array(Image.open(os.path.join(REF, name + '.png')))
        assert lab.dtype == np.uint8 and lab.ndim == 2
        lab = lab.astype('int32')
        lab[lab > 20] = 255                       # data.py:121 with num_classes = 21
        out[name] = lab.astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, 'voc_example_labels.npz'), **out)
    for k, v in out.items():
        print(k, v.shape, dict(zip(*np.unique(v, return_counts=True))))
