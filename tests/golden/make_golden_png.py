#!/usr/bin/env python3
"""Generate the PNG row-filter golden vectors with the REFERENCE's own code.

Runs `encodePng` of the reference's compiled WebAssembly build (oracle/_ref/pixo_bg.wasm) under
node via oracle/ref_wasm.js on deterministic inputs, parses the PNG, inflates the IDAT stream
(zlib) and records per case: the filter byte of every row, sha256 and length of the filtered
stream, the zlib trailer (Adler-32 of the filtered stream); small streams are stored verbatim
under tests/golden/png/.  Build container only (needs node + the staged wasm).

Presets (png/mod.rs:129-183): 0 = AdaptiveFast (sequential, stateful in this build), no
reductions; 1 = Adaptive + optimize_alpha + colour/palette reduction — inputs are chosen so that
those leave the pixels alone (alpha never 0, > 256 colours, not gray, not opaque); 2 = Bigrams with the
same reductions (and the slow optimal DEFLATE, so only small images).

    python tests/golden/make_golden_png.py [--huge]     # --huge adds 4096x4096 RGBA (10 s, 0.7 GB)
"""
import hashlib
import json
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

STORE_LIMIT = 24000
BPP = {0: 1, 1: 2, 2: 3, 3: 4}


def rgba_gradient(w, h):
    rgb = synth.gradient_rgb(w, h).reshape(h, w, 3)
    a = (1 + (np.arange(w)[None, :] * 3 + np.arange(h)[:, None] * 5) % 254).astype(np.uint8)
    return np.concatenate([rgb, a[:, :, None]], axis=2).reshape(-1)


def make_input(c):
    w, h, ct, gen, seed = c["w"], c["h"], c["color_type"], c["gen"], c.get("seed", 42)
    if gen == "noise":
        b = synth.lcg_bytes(w * h * BPP[ct], seed)
        if ct in (1, 3):
            b[BPP[ct] - 1::BPP[ct]] |= 1
        return b
    if gen == "gradient":
        if ct == 3:
            return rgba_gradient(w, h)
        if ct == 2:
            return synth.gradient_rgb(w, h)
    if gen == "flat":
        return synth.flat_blocks(w, h)
    raise ValueError(c)


def cases(huge=False):
    cs = []
    def add(gen, w, h, ct, preset, seed=42):
        cs.append(dict(gen=gen, w=w, h=h, color_type=ct, preset=preset, seed=seed,
                       name="%s_%dx%d_c%d_p%d_s%d" % (gen, w, h, ct, preset, seed)))
    for preset in (0, 1):
        add("noise", 128, 96, 3, preset); add("noise", 131, 67, 3, preset, 7); add("noise", 256, 128, 3, preset)
        add("noise", 200, 100, 2, preset); add("noise", 97, 83, 2, preset, 3)
        add("gradient", 160, 80, 3, preset); add("gradient", 320, 200, 2, preset)
        add("noise", 70, 70, 3, preset, 5)            # area > 4096
        add("noise", 64, 64, 3, preset, 5)            # area == 4096: forced Sub (filter.rs:76-86)
        if preset == 0:
            add("noise", 17, 9, 3, preset, 5)        # (preset 1 would palettise 153 pixels)
        add("noise", 1024, 40, 3, preset, 11)
    # preset 2: FilterStrategy::Bigrams (png/mod.rs:179)
    add("noise", 128, 96, 3, 2); add("noise", 131, 67, 3, 2, 7); add("gradient", 160, 80, 3, 2); add("gradient", 320, 200, 2, 2)
    add("noise", 97, 83, 2, 2, 3); add("noise", 64, 64, 3, 2, 5); add("noise", 70, 70, 3, 2, 5); add("noise", 1024, 40, 3, 2, 11)
    add("noise", 300, 200, 0, 0, 9); add("noise", 300, 200, 1, 0, 9)   # gray, gray+alpha through preset 0
    add("flat", 256, 256, 2, 0)
    add("noise", 1920, 1080, 3, 1)
    if huge:
        add("noise", 4096, 4096, 3, 1)                # C5
    return cs


def parse_png(png):
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    i, idat, ihdr = 8, [], None
    while i < len(png):
        n, typ = struct.unpack(">I4s", png[i:i + 8])
        body = png[i + 8:i + 8 + n]
        if typ == b"IHDR": ihdr = struct.unpack(">IIBBBBB", body)
        if typ == b"IDAT": idat.append(body)
        i += 12 + n
    z = b"".join(idat)
    return ihdr, zlib.decompress(z), struct.unpack(">I", z[-4:])[0]


def main():
    huge = "--huge" in sys.argv
    cs = cases(huge)
    os.makedirs(os.path.join(HERE, "png"), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        man = {"cases": []}
        for k, c in enumerate(cs):
            inp = os.path.join(tmp, "in%d.bin" % k)
            make_input(c).tofile(inp)
            man["cases"].append(dict(kind="png", input=inp, w=c["w"], h=c["h"], color_type=c["color_type"], preset=c["preset"],
                                     lossy=False, output=os.path.join(tmp, "out%d.png" % k)))
        mp = os.path.join(tmp, "manifest.json")
        json.dump(man, open(mp, "w"))
        res = subprocess.run(["node", "--max-old-space-size=4096", os.path.join(ROOT, "oracle", "ref_wasm.js"), mp],
                             stdout=subprocess.PIPE, check=True).stdout.decode().strip().splitlines()
        out = []
        for k, (c, line) in enumerate(zip(cs, res)):
            r = json.loads(line)
            assert r["ok"], (c, r)
            png = open(man["cases"][k]["output"], "rb").read()
            (w, h, depth, ctype, _, _, _), flt, trailer = parse_png(png)
            png_ct = {0: 0, 4: 1, 2: 2, 6: 3}.get(ctype, -1)
            row = c["w"] * BPP[c["color_type"]] + 1
            assert (w, h, depth, png_ct) == (c["w"], c["h"], 8, c["color_type"]), ("reduction changed the pixel format", c, ctype)
            assert len(flt) == row * c["h"] and trailer == zlib.adler32(flt)
            filters = bytes(flt[y * row] for y in range(c["h"]))
            rec = dict(c, png_len=len(png), filtered_len=len(flt), filtered_sha256=hashlib.sha256(flt).hexdigest(),
                       adler32=trailer, filters="".join(str(f) for f in filters))
            if len(flt) <= STORE_LIMIT:
                open(os.path.join(HERE, "png", c["name"] + ".flt"), "wb").write(flt)
                rec["stored"] = True
            out.append(rec)
            print(c["name"], len(flt), "%08x" % trailer, {f: rec["filters"].count(str(f)) for f in range(5)})
    dst = os.path.join(HERE, "png_cases.json")
    if not huge and os.path.exists(dst):  # keep a previously generated huge case
        old = [c for c in json.load(open(dst))["cases"] if c["w"] * c["h"] > 4000 * 4000]
        out += old
    json.dump({"wasm_sha256": hashlib.sha256(open(os.path.join(ROOT, "oracle", "_ref", "pixo_bg.wasm"), "rb").read()).hexdigest(),
               "cases": out}, open(dst, "w"), indent=0)


if __name__ == "__main__":
    main()
