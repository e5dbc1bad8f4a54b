#!/usr/bin/env python3
"""Same-box, same-process A/B of whole LIBRARIES on the metric's workload (configs[1]: one 4096x4096 RGB8 4:2:0 launch).

    python tools/ab_binaries.py [--rounds 9] [--steps 200] name=path.so ...      (default: r03, r04, r05 = the in-tree library)

Every library is dlopen'ed into this one process (RTLD_LOCAL), the variants take turns block by block — settle, then per round
one block of --steps launches per variant in rotating order, HIP events on the launch stream — so that clocks, box and moment are
shared.  The plain copy of the same bytes (pixo_hip_debug_stream_copy of the in-tree library) runs as one more variant.
Each variant's tuple for buffer 0 is compared with the first variant's (bit-exact or the run aborts).
Prints per variant: median / min / max us per launch, fraction of 8 TB/s, ratio to the copy."""
import ctypes as C
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import synth
    args = sys.argv[1:]
    rounds, steps, parity = 9, 200, True
    libs = []
    while args:
        a = args.pop(0)
        if a == "--rounds":
            rounds = int(args.pop(0))
        elif a == "--steps":
            steps = int(args.pop(0))
        elif a == "--no-parity":  # timing-only experiment builds that compute garbage on purpose
            parity = False
        else:
            n, p = a.split("=", 1)
            p, _, sw = p.partition("@")  # name=path.so@debug switches (a COPY of the library per set of switches: one state per dlopen'ed file)
            libs.append((n, os.path.join(ROOT, p) if not os.path.isabs(p) else p, sw))
    if not libs:
        libs = [("r03", os.path.join(ROOT, "tools/ab/ab_r03.so"), ""), ("r04", os.path.join(ROOT, "tools/ab/ab_r04.so"), ""),
                ("r05", os.path.join(ROOT, "pixo_amd/libpixo_hip.so"), "")]
    from pixo_amd import _lib
    _lib._preload_process_hip_runtime()
    W = H = 4096
    q = 80
    dev = torch.device("cuda", 0)
    yb, cbn = (W // 16) * (H // 16) * 4, (W // 16) * (H // 16)
    nbuf = 7
    base = torch.from_numpy(np.ascontiguousarray(synth.noise(W, H, 42))).to(dev)
    ins = [(base ^ torch.tensor(i, dtype=torch.uint8, device=dev)).contiguous() if i else base for i in range(nbuf)]
    outs = [(torch.empty((yb, 64), dtype=torch.int16, device=dev), torch.empty((cbn, 64), dtype=torch.int16, device=dev),
             torch.empty((cbn, 64), dtype=torch.int16, device=dev)) for _ in range(nbuf)]
    couts = [torch.empty(W * H * 3, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    stream = torch.cuda.current_stream().cuda_stream
    sp = C.c_void_p(stream) if stream else None
    variants = []
    for name, path, sw in libs:
        L = C.CDLL(path)
        if sw:
            L.pixo_hip_debug_configure.argtypes = [C.c_char_p]
            assert L.pixo_hip_debug_configure(sw.encode()) == 0
        L.pixo_hip_jpeg_coeffs_device.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

        def step(i, L=L):
            k = i % nbuf
            y, cb, cr = outs[k]
            rc = L.pixo_hip_jpeg_coeffs_device(ins[k].data_ptr(), W, H, 2, 1, q, 1, y.data_ptr(), cb.data_ptr(), cr.data_ptr(), sp)
            assert rc == 0, rc
        variants.append((name, step))
    Lc = C.CDLL(os.path.join(ROOT, "pixo_amd/libpixo_hip.so"))
    Lc.pixo_hip_debug_stream_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]

    def cstep(i):
        k = i % nbuf
        rc = Lc.pixo_hip_debug_stream_copy(ins[k].data_ptr(), couts[k].data_ptr(), W * H * 3, sp)
        assert rc == 0, rc
    # parity between the variants (buffer 0)
    ref = None
    for name, step in variants:
        for t in outs[0]:
            t.zero_()
        step(0)
        torch.cuda.synchronize()
        got = [t.clone() for t in outs[0]]
        if ref is None:
            ref = got
        else:
            assert not parity or all(torch.equal(a, b) for a, b in zip(ref, got)), "variant %s gives a different tuple" % name
    variants.append(("copy", cstep))
    print("box:", torch.cuda.get_device_name(0), "| variants:", [n for n, _ in variants], "| rounds", rounds, "x", steps, "launches per block")
    # settle
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 0.3:
        for _ in range(16):
            variants[n % len(variants)][1](n)
            n += 1
        torch.cuda.synchronize()
    res = {name: [] for name, _ in variants}
    for r in range(rounds):
        order = variants[r % len(variants):] + variants[:r % len(variants)]
        for name, step in order:
            for i in range(20):
                step(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(steps):
                step(i)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / steps * 1e3)
    copy_med = statistics.median(res["copy"])
    alg = W * H * 6
    for name, _ in variants:
        v = sorted(res[name])
        med = statistics.median(v)
        print("%-6s median %7.3f us  min %7.3f  max %7.3f  frac of 8 TB/s %.4f  over copy %.4f" % (name, med, v[0], v[-1], alg / (med * 1e-6) / 8e12, med / copy_med))


if __name__ == "__main__":
    main()
