#!/usr/bin/env python3
"""Generate the JPEG golden vectors with the REFERENCE's own code.

Runs the reference's compiled WebAssembly build (oracle/_ref/pixo_bg.wasm, staged
by `make -C oracle ref`) under node via oracle/ref_wasm.js, on deterministic
synthetic inputs (tests/synth.py), and records for every case the output length
and sha256; small outputs are also stored verbatim under tests/golden/jpeg/.

This script needs /root/reference (or an already staged oracle/_ref) and node; it
is run in the build container only.  Tests read the committed results and never
touch /root/reference.

    python tests/golden/make_golden.py            # everything up to 4096x4096
    python tests/golden/make_golden.py --huge     # also 16384x16384 (31 s, 1.5 GB)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

STORE_LIMIT = 6000  # bytes: outputs up to this size are committed verbatim


def primaries(w, h):
    """Saturated colours incl. the two inputs whose Cb/Cr hit the 255 clamp
    (pure blue -> Cb=256 before clamp, pure red -> Cr=256; color.rs:73-76)."""
    import numpy as np
    pal = np.array([[0, 0, 255], [255, 0, 0], [0, 255, 0], [255, 255, 255], [0, 0, 0],
                    [255, 0, 255], [0, 255, 255], [255, 255, 0], [1, 0, 254], [254, 1, 0]],
                   np.uint8)
    idx = (np.arange(w)[None, :] // 3 + np.arange(h)[:, None] // 2) % len(pal)
    return pal[idx].reshape(-1)


GEN = {
    "noise": lambda w, h, seed=42: synth.noise(w, h, seed),
    "noise_gray": lambda w, h, seed=42: synth.noise_gray(w, h, seed),
    "gradient": lambda w, h, seed=0: synth.gradient_rgb(w, h),
    "flat_blocks": lambda w, h, seed=0: synth.flat_blocks(w, h),
    "checker": lambda w, h, seed=0: synth.checkerboard(w, h, 5),
    "const128": lambda w, h, seed=0: synth.constant(w, h, 128),
    "const255": lambda w, h, seed=0: synth.constant(w, h, 255),
    "const0_gray": lambda w, h, seed=0: synth.constant(w, h, 0, 1),
    "primaries": lambda w, h, seed=0: primaries(w, h),
}


def case(gen, w, h, q=80, s420=True, preset=0, color=2, seed=42):
    return dict(gen=gen, w=w, h=h, quality=q, s420=s420, preset=preset, color_type=color, seed=seed)


def matrix(huge=False):
    cs = []
    # SURVEY §8c small parity set
    cs += [case("noise", 16, 16), case("noise", 17, 13), case("noise", 17, 13, s420=False),
           case("noise", 40, 24, q=35), case("noise", 33, 50, q=95, s420=False),
           case("noise", 64, 64)]
    # reference EDGE_CASE_DIMENSIONS (tests/support/synthetic.rs:276), both subsamplings
    for (w, h) in [(1, 1), (2, 2), (7, 7), (8, 8), (9, 9), (16, 16), (15, 17), (1, 100), (100, 1),
                   (31, 33), (48, 16), (16, 48)]:
        for s in (True, False):
            cs.append(case("noise", w, h, s420=s, seed=7))
    for g, w, h in [("gradient", 72, 40), ("flat_blocks", 48, 48), ("const128", 32, 32),
                    ("const255", 24, 24), ("checker", 37, 29), ("primaries", 45, 35)]:
        for q, s in [(80, True), (85, False), (100, True), (1, False), (50, True)]:
            cs.append(case(g, w, h, q=q, s420=s))
    # every quality scale branch a few times on noise
    for q in (1, 2, 10, 25, 49, 50, 51, 75, 90, 99, 100):
        cs.append(case("noise", 24, 24, q=q, s420=(q % 2 == 0), seed=q))
    # gray (s420 flag must be ignored for gray: jpeg/mod.rs:1449)
    cs += [case("noise_gray", 31, 17, color=0), case("noise_gray", 64, 64, q=50, color=0, s420=False),
           case("noise_gray", 8, 8, q=90, color=0), case("const0_gray", 20, 20, color=0),
           case("noise_gray", 1, 1, color=0), case("noise_gray", 100, 3, q=30, color=0)]
    # optimised Huffman tables (preset 1)
    cs += [case("noise", 64, 64, preset=1), case("noise", 40, 24, q=85, s420=False, preset=1),
           case("noise", 33, 50, q=60, preset=1), case("noise_gray", 31, 17, color=0, preset=1),
           case("gradient", 72, 40, preset=1), case("const128", 32, 32, preset=1),
           case("flat_blocks", 48, 48, q=95, s420=False, preset=1),
           case("noise", 8, 8, q=100, preset=1, s420=False)]
    # BASELINE.json configs (hash only)
    cs += [case("noise", 512, 512), case("noise", 1920, 1080), case("noise", 1000, 1000, s420=True),
           case("noise", 1000, 1000, s420=False), case("gradient", 1024, 1024, q=85),
           case("noise", 4096, 4096), case("noise", 4096, 4096, s420=False),
           case("noise", 512, 512, preset=1)]
    if huge:
        cs.append(case("noise", 16384, 16384))
    return cs


ERROR_CASES = [
    dict(w=4, h=4, color_type=2, quality=0, preset=0, s420=True, nbytes=48),
    dict(w=4, h=4, color_type=2, quality=101, preset=0, s420=True, nbytes=48),
    dict(w=0, h=4, color_type=2, quality=80, preset=0, s420=True, nbytes=0),
    dict(w=4, h=0, color_type=2, quality=80, preset=0, s420=False, nbytes=0),
    dict(w=4, h=4, color_type=3, quality=80, preset=0, s420=True, nbytes=64),
    dict(w=4, h=4, color_type=1, quality=80, preset=0, s420=True, nbytes=32),
    dict(w=4, h=4, color_type=2, quality=80, preset=0, s420=True, nbytes=47),
    dict(w=4, h=4, color_type=0, quality=80, preset=0, s420=True, nbytes=48),
    dict(w=70000, h=1, color_type=0, quality=80, preset=0, s420=True, nbytes=70000),
    dict(w=4, h=4, color_type=2, quality=0, preset=0, s420=True, nbytes=0),  # quality checked first
]


def name_of(c):
    return "%s_%dx%d_q%d_%s_p%d_c%d_s%d" % (c["gen"], c["w"], c["h"], c["quality"],
                                             "420" if c["s420"] else "444", c["preset"],
                                             c["color_type"], c["seed"])


def main():
    huge = "--huge" in sys.argv
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"],
                          stdout=subprocess.DEVNULL)
    cases = matrix(huge)
    out_dir = os.path.join(HERE, "jpeg")
    os.makedirs(out_dir, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="pixo_golden_")
    man = {"cases": []}
    for i, c in enumerate(cases):
        data = GEN[c["gen"]](c["w"], c["h"], c["seed"])
        inp = os.path.join(tmp, "in%d.bin" % i)
        data.tofile(inp)
        man["cases"].append(dict(kind="jpeg", input=inp, w=c["w"], h=c["h"],
                                 color_type=c["color_type"], quality=c["quality"],
                                 preset=c["preset"], s420=c["s420"],
                                 output=os.path.join(tmp, "out%d.jpg" % i)))
    for i, e in enumerate(ERROR_CASES):
        inp = os.path.join(tmp, "err%d.bin" % i)
        synth.lcg_bytes(e["nbytes"], 3).tofile(inp)
        man["cases"].append(dict(kind="jpeg", input=inp, w=e["w"], h=e["h"],
                                 color_type=e["color_type"], quality=e["quality"],
                                 preset=e["preset"], s420=e["s420"]))
    mpath = os.path.join(tmp, "manifest.json")
    json.dump(man, open(mpath, "w"))
    res = subprocess.run(["node", "--max-old-space-size=4096",
                          os.path.join(ROOT, "oracle", "ref_wasm.js"), mpath],
                         check=True, capture_output=True, text=True).stdout.strip().splitlines()
    records = []
    for i, c in enumerate(cases):
        r = json.loads(res[i])
        assert r["ok"], (c, r)
        blob = open(man["cases"][i]["output"], "rb").read()
        rec = dict(c)
        rec["name"] = name_of(c)
        rec["len"] = len(blob)
        rec["sha256"] = hashlib.sha256(blob).hexdigest()
        if len(blob) <= STORE_LIMIT:
            rec["file"] = "jpeg/" + rec["name"] + ".jpg"
            open(os.path.join(HERE, rec["file"]), "wb").write(blob)
        records.append(rec)
    errs = []
    for i, e in enumerate(ERROR_CASES):
        r = json.loads(res[len(cases) + i])
        assert not r["ok"], (e, r)
        rec = dict(e)
        rec["error"] = r["error"]
        errs.append(rec)
    wasm_sha = hashlib.sha256(open(os.path.join(ROOT, "oracle", "_ref", "pixo_bg.wasm"), "rb").read()).hexdigest()
    if not huge:
        # keep a previously recorded 16384^2 entry (expensive to regenerate)
        old = os.path.join(HERE, "jpeg_cases.json")
        if os.path.exists(old):
            for r in json.load(open(old))["cases"]:
                if r["w"] == 16384:
                    records.append(r)
    json.dump({"reference_wasm_sha256": wasm_sha, "cases": records, "errors": errs},
              open(os.path.join(HERE, "jpeg_cases.json"), "w"), indent=1)
    print("wrote %d cases (%d stored verbatim), %d error cases" %
          (len(records), sum(1 for r in records if "file" in r), len(errs)))


if __name__ == "__main__":
    main()
