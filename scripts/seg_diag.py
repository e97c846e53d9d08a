import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import semantic_slam_mapping_amd as ssm, segnet_ref
from oracle.binding import Oracle
o = Oracle()
ctx = ssm.Context(0, orb_features=1000, max_batch=4, voxel_capacity_log2=16)
w = segnet_ref.make_weights(1234)
for l, (wt, sc, sh) in enumerate(w): ctx.segnet_set_layer(l, wt, sc, sh)
bgr = o.synth_frame(0x5EED0000, 0)[0]
labels, sem = ctx.classify(bgr)
logits = ctx.segnet_logits()
x = np.stack([o.resize(np.ascontiguousarray(bgr[:, :, c]), 480, 360) for c in range(3)])
import time
t=time.time(); ref16 = segnet_ref.forward(x, w, True).transpose(1, 2, 0); print('ref time', time.time()-t)
ref32 = segnet_ref.forward(x, w, False).transpose(1, 2, 0)
d = np.abs(logits - ref16)
print('max|ref16|', np.abs(ref16).max(), 'mean|ref16|', np.abs(ref16).mean())
print('abs err: max', d.max(), 'mean', d.mean(), 'p99', np.percentile(d, 99), 'p99.9', np.percentile(d, 99.9))
print('agree16', (labels == ref16.argmax(2)).mean(), 'agree32', (labels == ref32.argmax(2)).mean(), 'ref16 vs ref32 agree', (ref16.argmax(2) == ref32.argmax(2)).mean())
d32 = np.abs(ref16 - ref32); print('ref16 vs ref32 abs err: max', d32.max(), 'mean', d32.mean())
print('label hist', np.bincount(labels.ravel(), minlength=12))
