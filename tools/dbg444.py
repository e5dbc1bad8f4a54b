import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np
import synth, oracle_lib as O
from pixo_amd import jpeg
w, h = 512, 16
px = synth.noise(w, h, 1)
oy, ocb, ocr = O.coeffs(px, w, h, 2, 0, 80)
y, cb, cr = jpeg.coefficients(px, jpeg.JpegOptions.builder(w, h).quality(80).build())
for name, a, b in (("y", y, oy), ("cb", cb, ocb), ("cr", cr, ocr)):
    bad = np.nonzero((a != b).any(1))[0]
    print(name, "blocks", a.shape[0], "bad", len(bad), bad[:20])
    if len(bad):
        k = bad[0]
        print(" got ", a[k].reshape(8, 8)[:3]); print(" want", b[k].reshape(8, 8)[:3])
        # is it some other block of the oracle?
        for pl, arr in (("y", oy), ("cb", ocb), ("cr", ocr)):
            m = np.nonzero((arr == a[k]).all(1))[0]
            if len(m): print("  matches oracle", pl, m[:5])
