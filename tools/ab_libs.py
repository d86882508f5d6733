#!/usr/bin/env python3
"""A/B of whole library builds (e.g. -DFP8MI_CSTORE=n variants) on the bench's GEMM shapes: one child process per (library, round),
libraries interleaved, so box drift hits all of them alike.  Two figures per shape: the median per-dispatch kernel time (start/stop
events of each launch) and the wall time per launch of a back-to-back run (what a caller sees: includes whatever the end of one
kernel leaves for the start of the next, e.g. the L2 write-back of the release).
    python tools/ab_libs.py [--rounds R] lib1.so lib2.so ...      (paths relative to fp8-mps-metal_amd/ or absolute)"""
import os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fp8-mps-metal_amd")
SHAPES = [("c3", 512, 4096, 4096, "f32"), ("c3_bf16", 512, 4096, 4096, "bf16"), ("m1024", 1024, 4096, 4096, "f32"),
          ("mid", 2048, 4096, 4096, "bf16"), ("flux", 4096, 3072, 12288, "bf16"), ("shard", 4096, 3072, 1536, "bf16"),
          ("decode", 64, 14336, 4096, "bf16"), ("sq8k_f32", 8192, 8192, 8192, "f32"),
          ("flux_f32", 4096, 3072, 12288, "f32"), ("mid_f32", 2048, 4096, 4096, "f32"), ("sq8k_bf16", 8192, 8192, 8192, "bf16")]


def child():
    sys.path[:0] = [PKG]
    import torch, fp8_mi355x_lib as L
    dev = torch.device("cuda:0"); lib = L.load()
    g = torch.Generator(device=dev).manual_seed(1)
    ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
    s1 = torch.full((1,), 0.01, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for name, M, K, N, out in SHAPES:
        A = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g)
        nb = min(24, max(2, (320 << 20) // (N * K)))
        Bs = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
        C = torch.empty(M, N, dtype=torch.float32 if out == "f32" else torch.bfloat16, device=dev)

        def run(i, st=st):
            L.check(lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i % nb].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                           M, N, K, K, K, N, 0, 0, 0 if out == "f32" else 2, 0, 0, 0, 0, ws.data_ptr(), ws.numel(), st), "mm")
        reps = 48 if M * N * K < (1 << 36) else 12
        for i in range(6): run(i)
        torch.cuda.synchronize()
        with L.kernel_timer(reps) as kt:
            for i in range(reps): run(i)
        torch.cuda.synchronize()
        disp = statistics.median(kt.ms) * 1e3
        # back-to-back as ONE graph (no host in the way)
        # (a graph launch itself costs ~0.1-0.2 ms on this stack: the graph holds >= 20 ms of kernels so that it does not show)
        per_graph = max(reps, int(20e3 / disp))
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            cap = torch.cuda.current_stream().cuda_stream   # the capture stream, not the one the handle above was taken from
            for i in range(per_graph): run(i, cap)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): gr.replay()
        e1.record(); torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) * 1e3 / (3 * per_graph)
        print(f"R {name} {disp:.3f} {wall:.3f}", flush=True)
        del Bs, A, C, gr


def main():
    args = sys.argv[1:]
    rounds = 3
    if args and args[0] == "--rounds":
        rounds = int(args[1]); args = args[2:]
    libs = [a if os.path.isabs(a) else os.path.join(PKG, a) for a in args]
    res = {lib: {} for lib in libs}
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, FP8MI_LIB_PATH=lib, HIP_FORCE_DEV_KERNARG="1")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
            if out.returncode:
                print(f"# {lib}: child failed\n{out.stderr[-2000:]}"); continue
            for line in out.stdout.splitlines():
                if line.startswith("R "):
                    _, name, d, w = line.split()
                    res[lib].setdefault(name, []).append((float(d), float(w)))
    names = [s[0] for s in SHAPES]
    print(f"# medians over {rounds} rounds: per-dispatch us / back-to-back wall us per launch")
    print(f"{'library':28s}" + "".join(f"{n:>16s}" for n in names))
    for lib in libs:
        row = f"{os.path.basename(lib):28s}"
        for n in names:
            v = res[lib].get(n)
            row += f"{statistics.median(x[0] for x in v):8.2f}/{statistics.median(x[1] for x in v):7.2f}" if v else f"{'-':>16s}"
        print(row)


if __name__ == "__main__":
    child() if sys.argv[1:2] == ["--child"] else main()
