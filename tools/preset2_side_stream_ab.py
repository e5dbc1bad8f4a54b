import os, sys, time
root = "/root/repo"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import synth, statistics
from pixo_amd import jpeg
import oracle_lib as O
for (w, h) in ((64, 64), (512, 512), (1024, 1024), (1920, 1080)):
    for kind in ("noise", "gradient"):
        px = synth.noise(w, h, 42) if kind == "noise" else synth.gradient_rgb(w, h)
        row = []
        ref = None
        for form, sw in (("now", None), ("no direct stores", "no_direct_small"), ("one stream", "no_side_stats")):
            jpeg.debug_configure(sw)
            fn = lambda: jpeg.encode_jpeg(px, w, h, 2, 80, 2, True)
            for _ in range(20): r = fn()
            if ref is None: ref = bytes(r)
            assert bytes(r) == ref
            ts = []
            for _ in range(200):
                t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
            row.append("%s %7.1f us" % (form, statistics.median(ts) * 1e6))
        jpeg.debug_configure(None)
        if w <= 512:
            assert ref == bytes(O.encode_flat(px, w, h, 2, 80, 2, True)), "differs from the oracle"
        print("%4dx%-4d %-8s preset 2: %s   (%d B)" % (w, h, kind, "   ".join(row), len(ref)), flush=True)
