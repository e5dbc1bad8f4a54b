#!/usr/bin/env python3
"""Golden vectors for preset 2 ("max": optimised Huffman + progressive scans + trellis quantisation,
src/jpeg/mod.rs:197-216) made by the REFERENCE's own wasm build — the only way its API exposes the
progressive/trellis path.  Same harness as make_golden.py; results in jpeg_p2_cases.json, small
files verbatim under tests/golden/jpeg_p2/.  Build container only."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

STORE_LIMIT = 6000


def matrix():
    c = MG.case
    cs = [c("noise", 16, 16, preset=2), c("noise", 17, 13, preset=2), c("noise", 17, 13, s420=False, preset=2),
          c("noise", 40, 24, q=35, preset=2), c("noise", 33, 50, q=95, s420=False, preset=2), c("noise", 64, 64, preset=2),
          c("gradient", 72, 40, preset=2), c("gradient", 72, 40, q=85, s420=False, preset=2),
          c("flat_blocks", 48, 48, preset=2), c("flat_blocks", 48, 48, q=100, preset=2), c("const128", 32, 32, preset=2),
          c("checker", 37, 29, q=50, preset=2), c("primaries", 45, 35, q=1, s420=False, preset=2),
          c("noise_gray", 31, 17, color=0, preset=2), c("noise_gray", 64, 64, q=50, color=0, s420=False, preset=2),
          c("const0_gray", 20, 20, color=0, preset=2), c("noise", 1, 1, preset=2), c("noise", 100, 1, s420=False, preset=2),
          c("noise", 256, 200, q=75, preset=2), c("gradient", 320, 240, q=90, preset=2), c("noise", 512, 512, preset=2)]
    for q in (1, 10, 49, 50, 51, 99, 100):
        cs.append(c("noise", 24, 24, q=q, s420=(q % 2 == 0), seed=q, preset=2))
    return cs


def main():
    cases = matrix()
    out_dir = os.path.join(HERE, "jpeg_p2")
    os.makedirs(out_dir, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="pixo_golden_p2_")
    man = {"cases": []}
    for i, c in enumerate(cases):
        inp = os.path.join(tmp, "in%d.bin" % i)
        MG.GEN[c["gen"]](c["w"], c["h"], c["seed"]).tofile(inp)
        man["cases"].append(dict(kind="jpeg", input=inp, w=c["w"], h=c["h"], color_type=c["color_type"], quality=c["quality"],
                                 preset=2, s420=c["s420"], output=os.path.join(tmp, "out%d.jpg" % i)))
    mp = os.path.join(tmp, "manifest.json")
    json.dump(man, open(mp, "w"))
    lines = subprocess.run(["node", os.path.join(ROOT, "oracle", "ref_wasm.js"), mp], stdout=subprocess.PIPE, check=True).stdout.decode().strip().splitlines()
    out = []
    for i, (c, line) in enumerate(zip(cases, lines)):
        r = json.loads(line)
        assert r["ok"], (c, r)
        blob = open(man["cases"][i]["output"], "rb").read()
        rec = dict(c, name=MG.name_of(c), len=len(blob), sha256=hashlib.sha256(blob).hexdigest())
        if len(blob) <= STORE_LIMIT:
            open(os.path.join(out_dir, rec["name"] + ".jpg"), "wb").write(blob)
            rec["stored"] = True
        out.append(rec)
        print(rec["name"], len(blob))
    json.dump({"cases": out}, open(os.path.join(HERE, "jpeg_p2_cases.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
