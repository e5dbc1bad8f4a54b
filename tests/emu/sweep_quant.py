#!/usr/bin/env python3
"""Exhaustive proof-by-enumeration of the quantiser fast path (jpeg_tile.h quant_row8):
for EVERY divisor q in 1..255 and EVERY f32 x with |x| <= 4096 (all 2*1,166,016,513 bit
patterns incl. zeros and denormals) check that lanes the safety test does not flag already
equal roundf(x / q), i.e. (x / q).round() of the reference (quantize.rs:102).
|x| <= 4096 covers the DCT output range (|coef| <= 1024 * 1.4 for inputs in [-128, 128)).

    python tests/emu/sweep_quant.py [nproc]      # ~12 min on 8 cores
Result is written to profiles/quant_fastpath_sweep.txt."""
import ctypes as C
import multiprocessing as mp
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def one(q):
    import emu_lib as E
    # PIXO_EMU_LIB: an emulation library built with another -DPIXO_QUANT_EPS (experiments)
    L = C.CDLL(os.environ["PIXO_EMU_LIB"]) if os.environ.get("PIXO_EMU_LIB") else E.lib()
    L.emu_quant_exhaustive.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    hi = struct.unpack("<I", struct.pack("<f", 4096.0))[0]
    f, w, wf = C.c_long(), C.c_long(), C.c_long()
    L.emu_quant_exhaustive(q, 0, hi, C.byref(f), C.byref(w), C.byref(wf))
    return q, f.value, w.value, wf.value


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, (os.cpu_count() or 2) - 2)
    import emu_lib as E
    E.lib()  # build once before forking
    with mp.Pool(n) as pool:
        rows = pool.map(one, range(1, 256), chunksize=1)
    out = os.environ.get("PIXO_SWEEP_OUT") or os.path.join(os.path.dirname(os.path.dirname(HERE)), "profiles", "quant_fastpath_sweep.txt")
    with open(out, "w") as fh:
        fh.write("# q flagged(took exact divide) wrong_unflagged wrong_final ; x = every f32 with |x|<=4096\n")
        for r in rows:
            fh.write("%d %d %d %d\n" % r)
        fh.write("# TOTAL wrong_unflagged=%d wrong_final=%d\n" % (sum(r[2] for r in rows), sum(r[3] for r in rows)))
    print("wrong_unflagged", sum(r[2] for r in rows), "wrong_final", sum(r[3] for r in rows))
