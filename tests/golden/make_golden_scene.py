#!/usr/bin/env python3
"""Reference-made golden vectors on PHOTOGRAPH-LIKE content (round 6; VERDICT r5 #9: the reference benchmarks on photographs —
benches/BENCHMARKS.md:92-93 — which cannot travel; two structurally different synthetics stand in): `synth.photo` (sums of
blurred noise) and `synth.scene` (flat regions with hard edges, oriented texture, saturated details) through the reference's
own wasm build (oracle/_ref/pixo_bg.wasm under node, oracle/ref_wasm.js), presets 0 / 1 / 2, both subsamplings, gray; records
length + sha256 per case in tests/golden/jpeg_scene_cases.json.  Build container only (needs node and /root/reference or a
staged oracle/_ref); the tests read the committed json.        python tests/golden/make_golden_scene.py"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as G  # noqa: E402


def matrix():
    cs = []
    for gen in ("scene", "photo"):
        for (w, h) in [(64, 64), (200, 120), (333, 211), (512, 512), (1000, 700), (1920, 1080)]:
            for (q, s420, preset) in [(80, True, 0), (85, False, 0), (80, True, 1), (75, False, 1), (80, True, 2), (90, False, 2), (35, True, 0), (97, True, 0)]:
                if preset == 2 and w * h > 600 * 600:
                    continue  # (the wasm's trellis search is slow; large preset-2 files are covered by jpeg_p2_cases.json)
                cs.append(dict(gen=gen, w=w, h=h, quality=q, s420=s420, preset=preset, color_type=2, seed=42 + w % 7))
        cs.append(dict(gen=gen, w=4096, h=4096, quality=80, s420=True, preset=0, color_type=2, seed=42))
        cs.append(dict(gen=gen, w=4096, h=4096, quality=80, s420=False, preset=0, color_type=2, seed=42))
        cs.append(dict(gen=gen + "_gray", w=640, h=480, quality=80, s420=False, preset=0, color_type=0, seed=42))
        cs.append(dict(gen=gen + "_gray", w=1001, h=333, quality=60, s420=False, preset=1, color_type=0, seed=43))
    return cs


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    cases = matrix()
    tmp = tempfile.mkdtemp(prefix="pixo_golden_scene_")
    man = {"cases": []}
    for i, c in enumerate(cases):
        inp = os.path.join(tmp, "in%d.bin" % i)
        G.make_input(c).tofile(inp)
        man["cases"].append(dict(kind="jpeg", input=inp, w=c["w"], h=c["h"], color_type=c["color_type"], quality=c["quality"],
                                 preset=c["preset"], s420=c["s420"], output=os.path.join(tmp, "out%d.jpg" % i)))
    mpath = os.path.join(tmp, "manifest.json")
    json.dump(man, open(mpath, "w"))
    res = subprocess.run(["node", "--max-old-space-size=4096", os.path.join(ROOT, "oracle", "ref_wasm.js"), mpath],
                         check=True, capture_output=True, text=True).stdout.strip().splitlines()
    records = []
    for i, c in enumerate(cases):
        r = json.loads(res[i])
        assert r["ok"], (c, r)
        blob = open(man["cases"][i]["output"], "rb").read()
        rec = dict(c)
        rec["name"] = "%s_%dx%d_q%d_%s_p%d_c%d_s%d" % (c["gen"], c["w"], c["h"], c["quality"], "420" if c["s420"] else "444", c["preset"], c["color_type"], c["seed"])
        rec["len"] = len(blob)
        rec["sha256"] = hashlib.sha256(blob).hexdigest()
        rec["bits_per_pixel"] = round(len(blob) * 8 / (c["w"] * c["h"]), 3)
        records.append(rec)
        os.remove(man["cases"][i]["output"]); os.remove(man["cases"][i]["input"])
    wasm = os.path.join(ROOT, "oracle", "_ref", "pixo_bg.wasm")
    out = {"reference_wasm_sha256": hashlib.sha256(open(wasm, "rb").read()).hexdigest(), "cases": records}
    json.dump(out, open(os.path.join(HERE, "jpeg_scene_cases.json"), "w"), indent=0)
    print("%d cases -> tests/golden/jpeg_scene_cases.json" % len(records))


if __name__ == "__main__":
    main()
