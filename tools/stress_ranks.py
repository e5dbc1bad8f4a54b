"""Randomised multi-rank stress on ONE GPU: WORLD processes (gloo between them, all on GPU 0) run `sharded.encode_banded`
(device bands; plain / shared arena / optimised tables), `sharded.encode_batch` and `encode_gathered_device` on random shapes and
content (tests/fresh_cases.py) for SECONDS; rank dst compares every file with the oracle.
    python tools/stress_ranks.py WORLD SECONDS        (starts its own ranks)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main(rank, world, port, seconds):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np, torch, torch.distributed as dist
    import fresh_cases as F, oracle_lib as O
    from pixo_amd import ColorType, jpeg, sharded
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    raw = F._Raw(12345)  # the same stream on every rank: every rank draws the same case
    t_end = time.time() + seconds
    cases = bad = 0
    while True:
        go = torch.tensor([1 if time.time() < t_end else 0])
        dist.broadcast(go, 0)
        if not int(go[0]):
            break
        kind = F.KINDS[raw.below(len(F.KINDS))]
        ct = 0 if raw.below(4) == 0 else 2
        ch = 1 if ct == 0 else 3
        ss = raw.below(2) if ct else 0
        w, h = 1 + raw.below(900), 1 + raw.below(900)
        q = 1 + raw.below(100)
        opt = raw.below(3) == 0
        mode = raw.below(6)  # 0-2 banded, 3 banded + shared, 4 batch, 5 gathered (progressive)
        dst = raw.below(world)
        if mode == 4:
            w, h = 1 + raw.below(300), 1 + raw.below(300)
            n = 1 + raw.below(2 * world + 3)
            src = raw.below(world)
            imgs = [F.content(kind, w, h, ch, raw) for _ in range(n)]
            b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)).optimize_huffman(opt)
            d = torch.from_numpy(np.concatenate(imgs)).to(dev) if rank == src else None
            got = sharded.encode_batch(d, b.build(), n, src=src, dst=dst, device=0)
            if rank == dst:
                arena, offs, lens = got
                oo = O.make_options(w, h, ct, q, ss, optimize_huffman=opt)
                for i in range(n):
                    bad += arena[offs[i]: offs[i] + lens[i]].numpy().tobytes() != O.encode(imgs[i], oo)
            cases += 1
            continue
        px = F.content(kind, w, h, ch, raw)
        b = jpeg.JpegOptions.builder(w, h).color_type(ColorType(ct)).quality(q).subsampling(jpeg.Subsampling(ss)).optimize_huffman(opt)
        if mode == 5:
            b = b.progressive(True)
        opts = b.build()
        band = jpeg.band(w, h, ct, ss, world, rank)
        mine = torch.from_numpy(px[band["row_begin"] * w * ch: band["row_end"] * w * ch].copy()).to(dev)
        want = O.encode(px, O.make_options(w, h, ct, q, ss, optimize_huffman=opt, progressive=(mode == 5))) if rank == dst or mode == 3 else None
        if mode == 5:
            got = sharded.encode_gathered_device(mine, opts, dst=dst)
        elif mode == 3:
            size = len(want) + 64
            name = "pixo_stress_%s_%d" % (port, cases)
            shared = sharded.SharedFile(name, size, create=True) if rank == 0 else None
            dist.barrier()
            if rank != 0:
                shared = sharded.SharedFile(name, size, create=False)
            shared.register()
            n = sharded.encode_banded(mine, opts, dst=dst, device=0, shared=shared)
            got = shared.array()[:n].tobytes() if rank == dst else None
            dist.barrier()
            shared.close(unlink=rank == 0)
        else:
            got = sharded.encode_banded(mine, opts, dst=dst, device=0)
        if rank == dst and got != want:
            bad += 1
            print("MISMATCH", dict(kind=kind, w=w, h=h, ct=ct, ss=ss, q=q, opt=opt, mode=mode, dst=dst), flush=True)
        cases += 1
    t = torch.tensor([bad])
    dist.all_reduce(t)
    if rank == 0:
        print("world %d on one GPU: %d cases in %.0f s, %d mismatches, single-pass fallbacks (rank 0) %d" % (world, cases, seconds, int(t[0]), jpeg.lookback_fallbacks()), flush=True)
    dist.destroy_process_group()
    sys.exit(1 if int(t[0]) else 0)


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--rank":
        rank_main(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], float(sys.argv[5]))
    else:
        world, seconds = int(sys.argv[1]), float(sys.argv[2])
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(world), port, str(seconds)], env=env) for r in range(world)]
        rcs = [p.wait(timeout=seconds + 600) for p in ps]
        sys.exit(max(rcs))
