"""GPU: the trellis search on eight lanes per block against the one-lane form (both against the oracle for small images):
preset-2 files of random images, byte for byte.   python tools/trellis_forms_check.py [seconds] [seed]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import synth, oracle_lib as O
from pixo_amd import jpeg
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); n = bad = checked = 0
gens = [lambda w, h, sd: synth.noise(w, h, sd), lambda w, h, sd: synth.photo(w, h, sd), lambda w, h, sd: synth.gradient_rgb(w, h),
        lambda w, h, sd: synth.flat_blocks(w, h), lambda w, h, sd: synth.checkerboard(w, h, 1 + sd % 9), lambda w, h, sd: synth.constant(w, h, sd % 256),
        lambda w, h, sd: (synth.noise(w, h, sd) >> (sd % 7)).astype(np.uint8), lambda w, h, sd: np.where(synth.noise(w, h, sd) > 127, 255, 0).astype(np.uint8)]
while time.time() - t0 < budget:
    w, h = int(rng.randint(1, 700)), int(rng.randint(1, 500))
    q = int(rng.choice([1, 5, 20, 50, 75, 80, 90, 95, 100, rng.randint(1, 101)]))
    ss = int(rng.randint(0, 2)); sd = int(rng.randint(0, 1 << 20))
    px = gens[int(rng.randint(0, len(gens)))](w, h, sd)
    files = []
    for form in ("lane", "group"):
        jpeg.debug_configure("trellis_form=" + form)
        files.append(bytes(jpeg.encode_jpeg(np.ascontiguousarray(px).reshape(-1), w, h, 2, q, 2, bool(ss))))
    n += 1
    if files[0] != files[1]:
        bad += 1; print("FORMS DIFFER", w, h, q, ss, sd, len(files[0]), len(files[1]), flush=True)
    if w * h <= 200 * 200:
        checked += 1
        if files[1] != bytes(O.encode_flat(np.ascontiguousarray(px).reshape(-1), w, h, 2, q, 2, bool(ss))):
            bad += 1; print("ORACLE DIFFERS", w, h, q, ss, sd, flush=True)
jpeg.debug_configure(None)
print("trellis forms: %d random preset-2 files, one lane per block = eight lanes per block; %d of them also = the oracle; mismatches %d; %.0f s" % (n, checked, bad, time.time() - t0))
sys.exit(1 if bad else 0)
