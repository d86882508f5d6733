"""Time the peer-store all-gather alone (include/fp8mi_peer.h) with `world` processes on ONE GPU, by blocks per peer.
Same-device copies, not xGMI: what this shows is whether the push kernel's own structure (few deep blocks) can feed a link
(153 GB/s per peer) - not what the link does.  Usage: python tools/time_peer_gather.py [world=2] [slab_MiB=6]"""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, slab, q):
    sys.path.insert(0, os.path.join(ROOT, "fp8-mps-metal_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fp8_peer_gather
    dev = torch.device("cuda:0")
    rows = []
    for blocks in (0, 1, 2, 4, 8, 16, 32, 64):
        if blocks:
            os.environ["FP8MI_PEER_BLOCKS"] = str(blocks)
        else:
            os.environ.pop("FP8MI_PEER_BLOCKS", None)
        pg = fp8_peer_gather.PeerGather(world * slab, dev, timeout_us=20_000_000)
        for _ in range(5):
            pg.allgather(rank * slab, slab)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            pg.allgather(rank * slab, slab)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        assert pg.status() == 0
        pg.close()
        rows.append((blocks, us))
    q.put((rank, rows))
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    slab = (int(sys.argv[2]) if len(sys.argv) > 2 else 6) << 20
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, slab, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in ps:
        p.join(60)
    print(f"# peer-store all-gather alone: {world} processes on one GPU, slab {slab >> 20} MiB per rank; us per call (2 launches) on rank 0 / max over ranks,")
    print("# GB/s per peer = slab / time (each rank pushes its slab to world-1 peers at once)")
    for i, (blocks, _) in enumerate(res[0]):
        worst = max(res[r][i][1] for r in range(world))
        print(f"blocks/peer {'auto' if not blocks else blocks:>4}: {res[0][i][1]:8.2f} / {worst:8.2f} us   {slab / worst / 1e3:7.1f} GB/s per peer, "
              f"{slab * (world - 1) * world / worst / 1e3:7.1f} GB/s copied on the card")


if __name__ == "__main__":
    main()
