#!/usr/bin/env python3
"""Differential campaign, build container only (needs node + oracle/_ref/pixo_bg.wasm = the reference's own compiled build):
randomised cases of tests/fresh_cases.py through the REFERENCE (wasm `encode_jpeg`, src/wasm.rs:113-142) and through the
oracle's restatement (`po_encode_jpeg_flat`), whole files compared byte for byte.

    python tools/oracle_vs_wasm.py FIRST COUNT [--record N] [--max-side S]

--record N writes length + sha256 of the first N cases to tests/golden/jpeg_fresh_cases.json (the CPU and GPU tests then hold the
oracle and the HIP library to the reference's answers without node)."""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fresh_cases as F  # noqa: E402
import oracle_lib as O  # noqa: E402


MAX_SIDE = int(sys.argv[sys.argv.index("--max-side") + 1]) if "--max-side" in sys.argv else 320  # (larger images: another case set)


def run_chunk(ids, tmp):
    man = {"cases": []}
    opts, pxs = [], []
    for k, i in enumerate(ids):
        o, px = F.case_of(i, MAX_SIDE)
        inp = os.path.join(tmp, "in%d.bin" % k)
        px.tofile(inp)
        man["cases"].append(dict(kind="jpeg", input=inp, w=o["w"], h=o["h"], color_type=o["color_type"], quality=o["quality"],
                                 preset=o["preset"], s420=o["s420"], output=os.path.join(tmp, "out%d.jpg" % k)))
        opts.append(o); pxs.append(px)
    mp = os.path.join(tmp, "manifest.json")
    json.dump(man, open(mp, "w"))
    lines = subprocess.run(["node", "--max-old-space-size=4096", os.path.join(ROOT, "oracle", "ref_wasm.js"), mp],
                           check=True, capture_output=True, text=True).stdout.strip().splitlines()
    out = []
    for k, (o, px, line) in enumerate(zip(opts, pxs, lines)):
        r = json.loads(line)
        assert r["ok"], (o, r)
        ref = open(man["cases"][k]["output"], "rb").read()
        mine = O.encode_flat(px, o["w"], o["h"], o["color_type"], o["quality"], o["preset"], o["s420"])
        out.append((o, ref, bytes(mine)))
    return out


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    record = int(sys.argv[sys.argv.index("--record") + 1]) if "--record" in sys.argv else 0
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    tmp = tempfile.mkdtemp(prefix="pixo_fresh_")
    t0 = time.time()
    bad, recs, by = [], [], {}
    for lo in range(first, first + count, 64):
        for o, ref, mine in run_chunk(range(lo, min(lo + 64, first + count)), tmp):
            key = (o["kind"], o["preset"])
            by[key] = by.get(key, 0) + 1
            if ref != mine:
                bad.append(o)
                print("MISMATCH", o, len(ref), len(mine), flush=True)
            if o["id"] < first + record:
                recs.append(dict(o, len=len(ref), sha256=hashlib.sha256(ref).hexdigest()))
    print("cases %d..%d (max side %d): %d compared, %d mismatches, %.0f s" % (first, first + count - 1, MAX_SIDE, count, len(bad), time.time() - t0))
    print("by (kind, preset):", " ".join("%s/p%d:%d" % (k[0], k[1], v) for k, v in sorted(by.items())))
    if record:
        assert not bad
        json.dump({"generator": "tests/fresh_cases.py case_of(id)", "made_by": "tools/oracle_vs_wasm.py %d %d --record %d" % (first, count, record),
                   "cases": recs}, open(os.path.join(ROOT, "tests", "golden", "jpeg_fresh_cases.json"), "w"), indent=0)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
