"""golden vectors for the label tail of the generator (deeplabv3p/data.py:116-145), made with the REAL
sklearn.utils.class_weight.compute_class_weight the reference calls (scikit-learn is importable in this image);
run once: python tests/golden/make_label_weights.py -> tests/golden/label_weights.npz"""
import os
import numpy as np
from sklearn.utils import class_weight

rng = np.random.default_rng(7)
num_classes, ignore_index = 21, 255
cases = []
# ragged class mixes: few classes, all classes, labels above num_classes-1 (-> ignore), one single class, odd sizes
for P, values, probs in [(1021, [0, 12, 15, 255], [.6, .1, .25, .05]),
                         (4099, list(range(21)) + [255], None),
                         (777, [0, 3, 21, 40, 254, 255], None),
                         (513, [7], None),
                         (65 * 65, [0, 1, 2, 20, 200], [.9, .05, .03, .015, .005])]:
    lab = rng.choice(values, size=P, p=probs).astype(np.uint8)
    # the reference's own lines (data.py:116-121, 134-145)
    label = lab.astype('int32').flatten()
    label[label > (num_classes - 1)] = ignore_index
    weights = np.zeros(P, dtype='float32')
    class_list = np.unique(label)
    cw = class_weight.compute_class_weight(class_weight='balanced', classes=class_list, y=label)
    for class_id, w in zip(class_list, cw):
        np.putmask(weights, label == class_id, w)
    cases.append((lab, label.astype(np.float32), weights))
out = {}
for i, (lab, lf, w) in enumerate(cases):
    out['u8_%d' % i], out['labels_%d' % i], out['weights_%d' % i] = lab, lf, w
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'label_weights.npz'),
                    num_classes=num_classes, ignore_index=ignore_index, n=len(cases), **out)
print('wrote', len(cases), 'cases')
