"""GPU box: cases FIRST..FIRST+COUNT-1 of tests/fresh_cases.py through the HIP library (C ABI, `encode_jpeg`) and through the
oracle, whole files byte for byte (the oracle itself was held to the reference's wasm on cases 0..19999 in the build container).
   python tools/fresh_gpu_check.py [first] [count] [max side of the JPEG cases]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import fresh_cases as F
import oracle_lib as O
from pixo_amd import jpeg
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
max_side = int(sys.argv[3]) if len(sys.argv) > 3 else 320
t0 = time.time(); bad = 0
for i in range(first, first + count):
    o, px = F.case_of(i, max_side)
    a = bytes(jpeg.encode_jpeg(px, o["w"], o["h"], o["color_type"], o["quality"], o["preset"], o["s420"]))
    b = bytes(O.encode_flat(px, o["w"], o["h"], o["color_type"], o["quality"], o["preset"], o["s420"]))
    if a != b:
        bad += 1; print("MISMATCH", o, len(a), len(b), flush=True)
pbad = 0
from pixo_amd import png
import numpy as np
for i in range(first, first + count):
    o, px = F.png_case_of(i)
    bpp = F.PNG_BPP[o["color_type"]]
    strategy, flags, ostrat, stateful = {0: (png.FilterStrategy.ADAPTIVE_FAST, png.NO_RAYON, O.S_ADAPTIVE_FAST, True),
                                         1: (png.FilterStrategy.ADAPTIVE, 0, O.S_ADAPTIVE, False),
                                         2: (png.FilterStrategy.BIGRAMS, 0, O.S_BIGRAMS, False)}[o["preset"]]
    got, gad = png.apply_filters(px, o["w"], o["h"], bpp, strategy, flags)
    want, wad = O.png_filter(px, o["w"], o["h"], bpp, ostrat, stateful)
    if not np.array_equal(got, want) or gad != wad:
        pbad += 1; print("PNG MISMATCH", o, flush=True)
print("fresh png cases %d..%d: %d compared, %d mismatches" % (first, first + count - 1, count, pbad))
bad += pbad
print("fresh cases %d..%d (max side %d): %d compared, %d mismatches, single-pass fallbacks %d, %.0f s" %
      (first, first + count - 1, max_side, count, bad, jpeg.lookback_fallbacks(), time.time() - t0))
sys.exit(1 if bad else 0)
