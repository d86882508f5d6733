"""Soak test of the peer-store all-gather (include/fp8mi_peer.h): `world` processes on ONE GPU, thousands of back-to-back gathers of random slab sizes,
every gathered buffer verified on the device (mismatches accumulate in a counter; the host only looks at the end).  A lost or reordered flag, a slab
landing before its consumers are done, a stale epoch - all show as mismatches or as timeout bits.  Usage: python tools/stress_peer_gather.py [world=3] [iters=3000]"""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, iters, q):
    sys.path.insert(0, os.path.join(ROOT, "fp8-mps-metal_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fp8_peer_gather
    dev = torch.device("cuda:0")
    slab = 1 << 20
    pg = fp8_peer_gather.PeerGather(world * slab, dev, timeout_us=20_000_000)
    buf = pg.tensor(torch.uint8)
    view = buf.view(world, slab)
    rng = torch.Generator().manual_seed(4242)                 # the same sizes on every rank
    sizes = (torch.randint(1, slab // 16 + 1, (iters,), generator=rng) * 16).tolist()
    ar = torch.arange(slab, device=dev, dtype=torch.int32)
    ranks = torch.arange(world, device=dev, dtype=torch.int32)[:, None]
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    for it, n in enumerate(sizes):
        view[rank, :n] = ((rank * 37 + it * 11 + ar[:n]) % 251).to(torch.uint8)
        if it % 3 == 2:                                       # every third gather from a side stream, ordered by events as the sharded linear does
            side.wait_stream(torch.cuda.current_stream(dev))
            pg.allgather(rank * slab, n, side.cuda_stream)
            torch.cuda.current_stream(dev).wait_stream(side)
        else:
            pg.allgather(rank * slab, n)
        want = ((ranks * 37 + it * 11 + ar[None, :n]) % 251).to(torch.uint8)
        bad += (view[:, :n] != want).sum()
    status = pg.status()
    q.put((rank, int(bad.item()), status))
    pg.close()
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, iters, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=1100) for _ in range(world))
    for p in ps:
        p.join(60)
    print(f"# peer-store all-gather soak: {world} processes on one GPU, {iters} gathers of random sizes (16 B .. 1 MiB), every gathered buffer checked on the device")
    for rank, bad, status in res:
        print(f"rank {rank}: mismatching bytes {bad}, timeout status {status}")
    print("OK" if all(b == 0 and s == 0 for _, b, s in res) else "FAILED")
    return 0 if all(b == 0 and s == 0 for _, b, s in res) else 1


if __name__ == "__main__":
    sys.exit(main())
