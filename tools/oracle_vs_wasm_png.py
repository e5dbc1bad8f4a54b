#!/usr/bin/env python3
"""Differential campaign for the PNG row filters (C5), build container only: randomised cases of tests/fresh_cases.py
(`png_case_of`) through the REFERENCE's wasm `encode_png` (src/wasm.rs:79) — the PNG is parsed, its IDAT stream inflated — and
through the oracle's `po_png_filter`: filtered stream and Adler-32 compared.  Cases whose pixel format the reference reduced
(presets 1/2: palette, gray, dropped alpha) are counted and skipped: the filters then ran on other bytes.

    python tools/oracle_vs_wasm_png.py FIRST COUNT [--record N]      # --record: tests/golden/png_fresh_cases.json"""
import hashlib, json, os, subprocess, sys, tempfile, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fresh_cases as F  # noqa: E402
import oracle_lib as O  # noqa: E402
import make_golden_png as MG  # noqa: E402  (parse_png)

STRATEGY = {0: (O.S_ADAPTIVE_FAST, True), 1: (O.S_ADAPTIVE, False), 2: (O.S_BIGRAMS, False)}


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    record = int(sys.argv[sys.argv.index("--record") + 1]) if "--record" in sys.argv else 0
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    tmp = tempfile.mkdtemp(prefix="pixo_fresh_png_")
    t0 = time.time(); bad = 0; reduced = 0; compared = 0; recs = []; by = {}
    for lo in range(first, first + count, 64):
        ids = range(lo, min(lo + 64, first + count))
        man = {"cases": []}; keep = []
        for k, i in enumerate(ids):
            o, px = F.png_case_of(i)
            inp = os.path.join(tmp, "in%d.bin" % k); px.tofile(inp)
            man["cases"].append(dict(kind="png", input=inp, w=o["w"], h=o["h"], color_type=o["color_type"], preset=o["preset"],
                                     lossy=False, output=os.path.join(tmp, "out%d.png" % k)))
            keep.append((o, px))
        mp = os.path.join(tmp, "manifest.json"); json.dump(man, open(mp, "w"))
        lines = subprocess.run(["node", "--max-old-space-size=4096", os.path.join(ROOT, "oracle", "ref_wasm.js"), mp],
                               check=True, capture_output=True, text=True).stdout.strip().splitlines()
        for k, ((o, px), line) in enumerate(zip(keep, lines)):
            r = json.loads(line); assert r["ok"], (o, r)
            (w, h, depth, ctype, _, _, _), flt, trailer = MG.parse_png(open(man["cases"][k]["output"], "rb").read())
            if (w, h, depth, {0: 0, 4: 1, 2: 2, 6: 3}.get(ctype, -1)) != (o["w"], o["h"], 8, o["color_type"]):
                reduced += 1; continue
            strategy, stateful = STRATEGY[o["preset"]]
            mine, adler = O.png_filter(px, o["w"], o["h"], F.PNG_BPP[o["color_type"]], strategy, stateful)
            compared += 1; key = (o["color_type"], o["preset"]); by[key] = by.get(key, 0) + 1
            if mine.tobytes() != flt or adler != trailer:
                bad += 1; print("MISMATCH", o, flush=True)
            if len(recs) < record:
                recs.append(dict(o, filtered_len=len(flt), filtered_sha256=hashlib.sha256(flt).hexdigest(), adler32=trailer))
    print("png cases %d..%d: %d compared, %d skipped (pixel format reduced by the reference), %d mismatches, %.0f s" %
          (first, first + count - 1, compared, reduced, bad, time.time() - t0))
    print("by (colour type, preset):", " ".join("c%d/p%d:%d" % (k[0], k[1], v) for k, v in sorted(by.items())))
    if record:
        assert not bad
        json.dump({"generator": "tests/fresh_cases.py png_case_of(id)", "cases": recs},
                  open(os.path.join(ROOT, "tests", "golden", "png_fresh_cases.json"), "w"), indent=0)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
