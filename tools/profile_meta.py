#!/usr/bin/env python3
"""What a committed counter profile (profiles/issue_<name>.json, profiles/traffic_<name>.json) was measured ON, so that bench.py can
tell when it no longer describes the loaded library (the counters in the bench line are read from profiles/, not
measured in the run): the library's version string and a hash of the SOURCES the profiled kernel is compiled from (the files below
+ the Makefile, comments and layout stripped).  A profile without these, or with other values than the tree's, is reported as `counters_stale`.

    python tools/profile_meta.py <name>            prints the meta of profile <name> for the tree as it stands
    python tools/profile_meta.py --stamp f.json    writes it into an existing profile (name = the file's `label`)"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pixo_amd", "csrc")
TILE = ["jpeg_kernels.hip", "jpeg_kernels.hpp", "jpeg_tile.h"]
SCAN = ["jpeg_scan_block.h", "jpeg_scan_dev.h", "dispatch_gate.hpp"]
SOURCES = {  # profile name prefix -> the files its kernel is compiled from
    "pixels_code": TILE + SCAN + ["jpeg_pixels_code.hip", "jpeg_pixels_code.hpp"],
    "scan_code": SCAN + ["jpeg_scan_fused.hip", "jpeg_entropy.hpp"],
    "prog_code": SCAN + ["jpeg_scan_fused.hip", "jpeg_entropy.hpp"],
    "stuff": SCAN + ["jpeg_scan_fused.hip", "jpeg_entropy.hpp"],
    "trellis": ["jpeg_trellis.hip", "jpeg_trellis.h", "jpeg_trellis.hpp"],
    "c5": ["png_filter.hip", "png_filter.hpp", "png_filter_math.h"],
    "c": TILE,  # c2, c2_444, c2_unaligned, c3
}


def sources_of(name):
    for prefix in sorted(SOURCES, key=len, reverse=True):
        if name.startswith(prefix):
            return SOURCES[prefix]
    return None


def _code_only(text):
    """The text without comments and without layout: a reworded comment must not make a profile stale, a changed token must."""
    import re
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*$", lambda m: m.group(0) if not m.group(0).lstrip().startswith("# ") else " ", text, flags=re.M)  # (Makefile comments)
    return " ".join(text.split())


def source_hash(name, root=None):
    files = sources_of(name)
    if files is None:
        return None
    h = hashlib.sha256()
    base = os.path.join(root, "pixo_amd", "csrc") if root else CSRC
    for f in sorted(files) + ["Makefile"]:
        with open(os.path.join(base, f), "r", encoding="utf-8", errors="replace") as fh:
            h.update(f.encode() + b"\0" + _code_only(fh.read()).encode() + b"\0")
    return h.hexdigest()[:16]


def library_version():
    sys.path.insert(0, ROOT)
    from pixo_amd import _lib
    return _lib.load().pixo_hip_version().decode()


def meta(name):
    return {"library_version": library_version(), "kernel_source_hash": source_hash(name),
            "kernel_sources": sources_of(name)}


def stale_reason(profile, name):
    """None when the committed profile `profile` (a dict) describes the tree's kernel; otherwise why not."""
    have = profile.get("kernel_source_hash")
    if not have:
        return "profile carries no kernel_source_hash (measured before round 6)"
    want = source_hash(name)
    if want != have:
        return "kernel sources changed since the profile was measured (%s -> %s)" % (have, want)
    try:
        v = library_version()
    except Exception:
        return None
    if profile.get("library_version") != v:
        return "library version %r, profile measured on %r" % (v, profile.get("library_version"))
    return None


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--stamp":
        d = json.load(open(sys.argv[2]))
        d.update(meta(d.get("label") or os.path.basename(sys.argv[2]).split("_", 1)[1].rsplit(".", 1)[0]))
        json.dump(d, open(sys.argv[2], "w"), indent=1)
        print(json.dumps(d))
    else:
        print(json.dumps(meta(sys.argv[1])))
