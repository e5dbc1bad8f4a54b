"""Loads the committed golden vectors (made by tests/golden/make_golden.py with the
reference's own wasm build) and regenerates their deterministic inputs."""
import hashlib
import json
import os

import numpy as np

import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _primaries(w, h):
    pal = np.array([[0, 0, 255], [255, 0, 0], [0, 255, 0], [255, 255, 255], [0, 0, 0],
                    [255, 0, 255], [0, 255, 255], [255, 255, 0], [1, 0, 254], [254, 1, 0]],
                   np.uint8)
    idx = (np.arange(w)[None, :] // 3 + np.arange(h)[:, None] // 2) % len(pal)
    return pal[idx].reshape(-1)


_GEN = {
    "noise": lambda w, h, seed: synth.noise(w, h, seed),
    "noise_gray": lambda w, h, seed: synth.noise_gray(w, h, seed),
    "gradient": lambda w, h, seed: synth.gradient_rgb(w, h),
    "flat_blocks": lambda w, h, seed: synth.flat_blocks(w, h),
    "checker": lambda w, h, seed: synth.checkerboard(w, h, 5),
    "const128": lambda w, h, seed: synth.constant(w, h, 128),
    "const255": lambda w, h, seed: synth.constant(w, h, 255),
    "const0_gray": lambda w, h, seed: synth.constant(w, h, 0, 1),
    "primaries": lambda w, h, seed: _primaries(w, h),
    # photograph-like content (round 6, tests/golden/make_golden_scene.py); gray = the luminance-like mean of the three channels
    "photo": lambda w, h, seed: synth.photo(w, h, seed),
    "scene": lambda w, h, seed: synth.scene(w, h, seed),
    "photo_gray": lambda w, h, seed: (synth.photo(w, h, seed).reshape(-1, 3).astype(np.uint16).sum(axis=1) // 3).astype(np.uint8),
    "scene_gray": lambda w, h, seed: (synth.scene(w, h, seed).reshape(-1, 3).astype(np.uint16).sum(axis=1) // 3).astype(np.uint8),
}


def load():
    return json.load(open(os.path.join(GOLDEN_DIR, "jpeg_cases.json")))


def scene_cases(max_pixels=None, min_pixels=0):
    """the reference-made vectors on photograph-like content (jpeg_scene_cases.json)"""
    d = json.load(open(os.path.join(GOLDEN_DIR, "jpeg_scene_cases.json")))
    return [c for c in d["cases"] if min_pixels <= c["w"] * c["h"] and (max_pixels is None or c["w"] * c["h"] <= max_pixels)]


def cases(max_pixels=None, min_pixels=0):
    out = []
    for c in load()["cases"]:
        n = c["w"] * c["h"]
        if n < min_pixels or (max_pixels is not None and n > max_pixels):
            continue
        out.append(c)
    return out


def make_input(c):
    return _GEN[c["gen"]](c["w"], c["h"], c["seed"])


def golden_bytes(c):
    if "file" in c:
        return open(os.path.join(GOLDEN_DIR, c["file"]), "rb").read()
    return None


def check(c, blob):
    """Byte-exact check of an encoder output against golden case c."""
    assert len(blob) == c["len"], (c["name"], len(blob), c["len"])
    assert hashlib.sha256(blob).hexdigest() == c["sha256"], c["name"]
    g = golden_bytes(c)
    if g is not None:
        assert blob == g, c["name"]
