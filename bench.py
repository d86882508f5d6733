#!/usr/bin/env python3
"""
bench.py - measurement harness for the FP8 e4m3fn hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|gemm|gemv|flux|quantize|dequant]

Prints ONE JSON line (rank 0).  A "step" is one pass of the hot path over one
batch of synthetic inputs that are already resident in HBM: `inner` back-to-back
launches of the C-ABI entry point (fp8mi_scaled_mm / fp8mi_encode /
fp8mi_dequant), each on a DIFFERENT weight buffer out of a rotating set larger
than the 256 MiB Infinity Cache, so weights stream from HBM as they do in a
model whose layers are visited once per forward.  The batch is replayed as a HIP
graph, so the number is not bounded by Python launch overhead.

Workloads (BASELINE.json configs):
  gemm     M=512, K=N=4096, fp32 out   (configs[2]; the N=1 default, TFLOP/s)
  gemv     M=1, K=14336, N=4096        (configs[1]; GB/s of algorithmic bytes)
  gemv_c1  M=1, K=N=4096               (configs[0] - the reference's CPU-runnable case - on the GPU: one dispatch floor + 16.8 MB)
  flux     M=4096, K=3072, N=12288, bf16 out (configs[3]); with --gpus N > 1 the
           N dimension is column-sharded over the ranks and the output is
           all-gathered over RCCL/xGMI ("strong" scaling: fixed total work)
  mid      M=2048, K=N=4096, bf16 out  (a mid-size GEMM: one round of 256x128 tiles)
  wide     M=1024, K=N=4096, fp32 out  (twice C3's rows: one 128x128 tile per CU - the deep-ring class)
  skinny   M=4, K=N=4096               (the reference's batch-4 shape, README.md:77; GB/s)
  gemv_sq  M=1, K=N=14336              (the reference's published big GEMV shape, README.md:77-82)
  quantize / quantize_rne / dequant   2^30 elements   (configs[4]; _rne = torch/OCP rounding through the hardware convert)

Every step lasts >= 16 ms (as many passes over the rotating weight buffers as that takes), so the driver's 20 steps
time >= 0.3 s of sustained work per workload.  `measured_ceilings` are probes run in the same process (tools/
ceiling_probe.hip: register-only fp8 MFMA loop, streaming reads); `roofline.traffic` comes from the PMC passes of
tools/profile_round.sh and is null unless profiles/pmc_traffic.json was measured on the kernel sources running now.

Two objects ride on the line: `roofline` for the dominant kernel - algorithmic
flops or bytes per launch / the kernel's average DEVICE duration, the latter
measured live with per-dispatch HIP events (fp8mi_profile_begin/_end, the
timestamps rocprofv3 --kernel-trace reads) - and `cpu_baseline`, the C oracle
(oracle/fp8_oracle.c, OpenMP) timed on this box's host cores on a bounded
sample of the same workload.  The oracle is only ever the baseline / checker.
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

# Kernel arguments in device memory instead of host memory: a runtime knob of the HIP runtime that must be set before
# libamdhip64 loads (i.e. before `import torch`).  Measured on MI355X (profiles/r02_kernarg.txt): every launch of a
# 5-20 us kernel is 1.2-1.7 us shorter (a scalar load from a host-memory kernarg segment is a PCIe round trip).  The
# kernels also load all their arguments in one clause, so the setting is a deployment recommendation,
# not a requirement; the line reports which way it ran (`config.dev_kernarg`).
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "fp8-mps-metal_amd")
ORACLE = os.path.join(ROOT, "oracle")
for _p in (PKG, ORACLE):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import fp8_mi355x_lib as L  # noqa: E402

METRIC = "FP8 _scaled_mm TFLOPS (M>=4) + GEMV HBM GB/s (M=1) vs roofline, 1/8 GPU"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP8_FLOP_PER_CLK_CU = 8192     # dense fp8 MFMA (f8f6f4 K=128 path): 5.03 PFLOP/s at 256 CU x 2.4 GHz
CACHE_BYTES = 256 << 20        # Infinity Cache


def log(*a):
    print(*a, file=sys.stderr, flush=True)


DATA_MODE = "gauss"


def clean_bytes(shape, dev, gen):
    """Synthetic e4m3 operand bytes (SURVEY 8d offers both):
      gauss   - seeded N(0,1) values amax-quantised to e4m3 (scale 448/amax) with
                the product's own encode kernel: the bit statistics of real
                weights / activations (default);
      uniform - uniformly random bytes with the two NaN patterns remapped
                (0x7F->0x7E, 0xFF->0xFE): every exponent equally likely, the
                worst case for switching power (the chip clocks lower on it);
      zeros   - all-zero bytes (upper bound on clocks; never a reported number)."""
    if DATA_MODE == "zeros":
        return torch.zeros(shape, dtype=torch.uint8, device=dev)
    if DATA_MODE == "uniform":
        b = torch.randint(0, 256, shape, dtype=torch.uint8, device=dev, generator=gen)
        return torch.where((b & 0x7F) == 0x7F, b ^ 1, b)
    import fp8_mi355x_native as native
    n = 1
    for d in shape:
        n *= d
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    chunk = 1 << 26
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        x = torch.randn(m, device=dev, generator=gen)
        out[i:i + m] = native.fp8_quantize(x)[0]
    return out.reshape(shape)


MAX_STREAMS = 4

# name -> (M, K, N); "decode" = a small batch against the C2 weight shape (split-K fills the chip)
# "gemv_sq" = the reference's one published big GEMV shape (test_fp8_metal.py:233-235, README.md:77-82: 2.38 ms on M4 Pro)
MM_WORKLOADS = {"gemm": (512, 4096, 4096), "gemv": (1, 14336, 4096), "gemv_c1": (1, 4096, 4096), "flux": (4096, 3072, 12288),   # gemv_c1 = BASELINE config C1 on the GPU
                "skinny": (4, 4096, 4096), "decode": (64, 14336, 4096), "gemv_sq": (1, 14336, 14336),
                "mid": (2048, 4096, 4096),    # "mid" = one round of 256x128 tiles (the mid-size class between C3 and FLUX)
                "wide": (1024, 4096, 4096)}   # "wide" = twice C3's rows, fp32 out: one 128x128 tile per CU (FP8MI_KERNEL_GEMM_128D)
WEIGHT_STREAMING = ("gemv", "gemv_c1", "skinny", "decode", "gemv_sq")   # quoted in GB/s of algorithmic bytes
TARGET_STEP_S = 0.016   # a step lasts >= 16 ms: the driver's 20 steps time >= 0.3 s of sustained work per workload


class Workload:
    """One named workload: buffers + a `launch(i)` closure calling the C ABI."""

    def __init__(self, name, dev, world=1, rank=0, kernel=L.KERNEL_AUTO, nbuf=None, sharded=False, gather="rccl"):
        self.name, self.dev, self.world, self.rank, self.kernel = name, dev, world, rank, kernel
        self.peer = None
        self.sharded = sharded or world > 1
        self.lib = L.load()
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        self.collective = None
        if name in MM_WORKLOADS:
            M, K, N = MM_WORKLOADS[name]
            self.M, self.K, self.N_total = M, K, N
            Nl = N // world
            assert N % world == 0
            self.N = Nl
            self.out_dtype = torch.bfloat16 if name in ("flux", "decode", "mid") else torch.float32
            esz = 2 if name in ("flux", "decode", "mid") else 4
            # (>= 4 buffers for the 205 MB matrix: with 2 part of every pass was still in the 256 MiB Infinity Cache and the
            #  "HBM" rate read above the streaming ceiling of the same run)
            nbuf = nbuf or max(4 if Nl * K > (128 << 20) else 2, -(-int(1.25 * CACHE_BYTES) // (Nl * K)))
            self.A = clean_bytes((M, K), dev, torch.Generator(device=dev).manual_seed(99))  # replicated activations
            self.Bs = [clean_bytes((Nl, K), dev, gen) for _ in range(nbuf)]
            self.sa = torch.full((1,), 0.01, dtype=torch.float32, device=dev)
            self.sb = torch.full((1,), 0.01, dtype=torch.float32, device=dev)
            self.flops = 2.0 * M * Nl * K
            self.bytes = float(M * K + Nl * K + esz * M * Nl)
            self.unit_flops = name not in WEIGHT_STREAMING
            # split-K workspaces (include/fp8mi.h): owned by the caller, counter block zeroed once; one per
            # concurrent launch chain (--streams)
            self.wss = [torch.zeros(int(self.lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
                        for _ in range(MAX_STREAMS)]
            self.inner = nbuf
            if self.sharded:
                # the shipped N-column-sharded linear (fp8_sharded_linear.py): transposed blocks,
                # chunk-cyclic rows, all-gather of chunk j on a side stream under the GEMM of chunk j+1
                from fp8_sharded_linear import ColumnShardedFP8Linear
                self.chunks = 2   # per-call RCCL latency vs overlap: 2 row chunks per rank
                if gather == "peer":   # ONE IPC-mapped gather buffer for every weight buffer's linear (include/fp8mi_peer.h)
                    import fp8_peer_gather
                    self.peer = fp8_peer_gather.PeerGather(N * M * esz, dev, timeout_us=5_000_000)   # (a wait is microseconds; 5 s bounds a run that went wrong)
                self.linears = [ColumnShardedFP8Linear(B, self.sb, None, N=N, chunks=self.chunks, out_dtype=self.out_dtype, peer=self.peer)
                                for B in self.Bs]
                self.Cs = [torch.empty(Nl, M, dtype=self.out_dtype, device=dev) for _ in range(2)]
            else:
                self.Cs = [torch.empty(M, Nl, dtype=self.out_dtype, device=dev) for _ in range(2)]
            self.code = L.BF16 if name in ("flux", "decode", "mid") else L.F32
            self.desc = {"workload": f"{name}: M={M} K={K} N={N} e4m3fn, per-tensor scales, "
                                     f"{'bf16' if name in ('flux', 'decode', 'mid') else 'fp32'} out, {nbuf} rotating weight buffers",
                         "M": M, "K": K, "N": N}
        elif name == "linear":
            # end-to-end dynamic-quant linear on the FLUX shape: bf16 activations -> amax -> scaled encode -> scaled_mm
            # with a bf16 result: three launches per linear, no host sync in between
            M, K, N = MM_WORKLOADS["flux"]
            self.M, self.K, self.N, self.N_total = M, K, N, N
            nbuf = nbuf or max(2, -(-int(1.25 * CACHE_BYTES) // (N * K)))
            self.x = (torch.randn(M, K, device=dev, generator=gen) * 3).to(torch.bfloat16)
            self.xq = torch.empty(M, K, dtype=torch.uint8, device=dev)
            self.scales = torch.empty(2, dtype=torch.float32, device=dev)
            self.Bs = [clean_bytes((N, K), dev, gen) for _ in range(nbuf)]
            self.sb = torch.full((1,), 0.01, dtype=torch.float32, device=dev)
            self.Cs = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(2)]
            self.wss = [torch.zeros(int(self.lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
                        for _ in range(MAX_STREAMS)]
            self.flops = 2.0 * M * N * K
            self.bytes = float(2 * M * K * 2 + M * K + M * K + N * K + 2 * M * N)
            self.unit_flops = True
            self.inner = nbuf
            self.desc = {"workload": f"linear: x bf16 ({M},{K}) -> amax + encode -> scaled_mm with W e4m3fn ({N},{K}) -> bf16; "
                                     f"3 launches per linear, {nbuf} rotating weight buffers", "M": M, "K": K, "N": N}
        elif name in ("quantize", "quantize_rne", "dequant"):
            n = 1 << 30
            self.count = n
            self.enc_mode = L.ENC_RNE if name == "quantize_rne" else L.ENC_REFERENCE
            if name != "dequant":
                self.src = [torch.randn(n, device=dev, generator=gen) * 16 for _ in range(1)]
                self.dst = [torch.empty(n, dtype=torch.uint8, device=dev)]
                self.bytes = 5.0 * n
            else:
                self.src = [clean_bytes((n,), dev, gen)]
                self.dst = [torch.empty(n, dtype=torch.float16, device=dev)]
                self.bytes = 3.0 * n
            self.flops = 0.0
            self.unit_flops = False
            self.inner = 2
            what = {"quantize": "fp32 -> e4m3fn, reference rounding rules", "dequant": "e4m3fn -> fp16",
                    "quantize_rne": "fp32 -> e4m3fn, torch / OCP round-to-nearest-even (hardware convert)"}[name]
            self.desc = {"workload": f"{name}: 2^30 elements ({what})", "elements": n}
        else:
            raise ValueError(name)

    def launch(self, i, stream, chain=0):
        lib = self.lib
        ws = self.wss[chain] if hasattr(self, "wss") else None
        if self.name in MM_WORKLOADS:
            B = self.Bs[i % len(self.Bs)]
            C = self.Cs[i % 2]
            if self.world > 1:  # transposed product: "A" operand = weight shard, "B_nk" operand = activations
                rc = lib.fp8mi_scaled_mm_ws(B.data_ptr(), self.A.data_ptr(), C.data_ptr(), self.sb.data_ptr(),
                                            self.sa.data_ptr(), None, None, self.N, self.M, self.K, self.K, self.K,
                                            self.M, 0, 0, self.code, 0, L.NAN_ZERO, self.kernel, 0,
                                            ws.data_ptr(), ws.numel(), stream)
            else:
                rc = lib.fp8mi_scaled_mm_ws(self.A.data_ptr(), B.data_ptr(), C.data_ptr(), self.sa.data_ptr(),
                                            self.sb.data_ptr(), None, None, self.M, self.N, self.K, self.K, self.K,
                                            self.N, 0, 0, self.code, 0, L.NAN_ZERO, self.kernel, 0,
                                            ws.data_ptr(), ws.numel(), stream)
        elif self.name == "linear":
            rc = lib.fp8mi_quantize(self.x.data_ptr(), L.BF16, self.xq.data_ptr(), self.scales.data_ptr(), self.x.numel(),
                                    L.ENC_REFERENCE, stream)
            L.check(rc, "bench launch linear/quantize")
            rc = lib.fp8mi_scaled_mm_ws(self.xq.data_ptr(), self.Bs[i % len(self.Bs)].data_ptr(), self.Cs[i % 2].data_ptr(),
                                        self.scales.data_ptr() + 4, self.sb.data_ptr(), None, None, self.M, self.N, self.K,
                                        self.K, self.K, self.N, 0, 0, L.BF16, 0, L.NAN_ZERO, self.kernel, 0,
                                        ws.data_ptr(), ws.numel(), stream)
        elif self.name in ("quantize", "quantize_rne"):
            rc = lib.fp8mi_encode(self.src[0].data_ptr(), L.F32, self.dst[0].data_ptr(), None, self.count,
                                  self.enc_mode, stream)
        else:
            rc = lib.fp8mi_dequant(self.src[0].data_ptr(), self.dst[0].data_ptr(), None, self.count, L.F16, stream)
        L.check(rc, f"bench launch {self.name}")

    def step(self, streams=None):
        """One step, eagerly on the current stream (also what gets captured)."""
        if self.sharded:
            for i in range(self.inner):
                self.out = self.linears[i % len(self.linears)](self.A, self.sa)  # (M, N) view of the gathered C^T
            return
        if streams:  # independent launches round-robin over side streams (inside a capture: parallel graph branches)
            cur = torch.cuda.current_stream(self.dev)
            ev = torch.cuda.Event()
            ev.record(cur)
            for st in streams:
                st.wait_event(ev)
            for i in range(self.inner):
                self.launch(i, streams[i % len(streams)].cuda_stream, chain=i % len(streams))
            for st in streams:
                cur.wait_stream(st)
            return
        s = torch.cuda.current_stream(self.dev).cuda_stream
        for i in range(self.inner):
            self.launch(i, s)


class stdout_to_stderr:
    """RCCL prints a version banner to STDOUT when a communicator is created (librccl, NCCL_DEBUG unset); the driver wants ONE JSON line there.
    File-descriptor level, because the writer is a C library."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def agree_on_graph(ok, graph, replay, flag_device):
    """Every rank must take the same branch: a rank whose capture failed must not sit in an all-reduce while the others replay a graph
    full of all-gathers (a deadlock, not an error).  So NOTHING captured is replayed before the ranks have agreed: all-reduce(MIN) of the
    local verdict first, then the first replay on every rank or on none.  `replay` runs (and synchronises) the local graph once; a failure
    of that first replay is a second, equally agreed, verdict.  Returns the graph or None - the same on every rank."""
    def all_ok(v):
        t = torch.tensor([1 if v else 0], dtype=torch.int32, device=flag_device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1
    if not all_ok(ok and graph is not None):
        return None
    ok2 = True
    try:
        replay(graph)
    except Exception as e:
        ok2 = False
        log(f"[bench] first replay of the captured sharded step failed ({e!r}); running it eagerly")
    return graph if all_ok(ok2) else None


def capture_sharded(w, dev, inject_failure=False):
    """The sharded step (per linear: `chunks` GEMM launches + `chunks` in-place all-gathers on a side stream) as ONE HIP graph: eagerly the host
    needs ~80 us per linear for the launches, events and collectives (1-rank rehearsal: 197 us per FLUX linear against 120 us of kernel), which at
    N = 8 - shard GEMMs of ~12 us per chunk - would be the critical path.  RCCL collectives capture into HIP graphs (tested with one rank:
    tests/test_gpu_patch.py).  The warm-up steps run real collectives on every rank (a failure there is fatal for the job, as for any eager step);
    the capture only RECORDS; the captured collectives first run after agree_on_graph()."""
    cur = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        w.step(None)   # warm: communicator, the module's side stream and events, allocator
        w.step(None)
    cur.wait_stream(side)
    torch.cuda.synchronize(dev)
    ok, graph = True, None
    try:
        if inject_failure:
            raise RuntimeError("injected capture failure")
        graph = torch.cuda.CUDAGraph()
        # thread_local: with more than one rank the process group's watchdog thread polls events while this thread captures; under the default
        # (global) capture mode such a call from ANOTHER thread invalidates the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            w.step(None)
    except Exception as e:   # capture not possible here: every rank falls back to eager steps
        ok, graph = False, None
        log(f"[bench] sharded step not captured ({e!r}); running it eagerly")

    def first_replay(g):
        g.replay()
        torch.cuda.synchronize(dev)
    return agree_on_graph(ok, graph, first_replay, dev)


def time_steps(w, steps, warmup, use_graph, world, n_streams=1):
    dev = w.dev
    graph = None
    # (capturable: RCCL collectives, and the peer-store gather - plain kernel launches whose epoch lives in device memory; not gloo's host staging)
    if (use_graph and w.sharded and dist.is_initialized() and (dist.get_backend() == "nccl" or w.peer is not None)
            and os.environ.get("FP8MI_BENCH_SHARDED_GRAPH", "1") == "1"):
        graph = capture_sharded(w, dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 and not w.sharded else None
    if w.name in ("quantize", "quantize_rne", "dequant", "linear"):
        streams = None  # these reuse one output buffer per launch
    if use_graph and world == 1 and not w.sharded:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            w.step(streams)  # warm every code path before capture
            # size the step: enough launches that one step lasts >= TARGET_STEP_S (a whole number of passes over the
            # rotating weight buffers), so the timed region is sustained work, not a 7 ms burst
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            w.step(streams)
            e1.record(side)
            e1.synchronize()
            per_pass = max(e0.elapsed_time(e1) * 1e-3, 1e-6)
            base = w.inner
            w.inner = base * max(1, min(4096 // max(base, 1), int(TARGET_STEP_S / per_pass + 0.999)))
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            w.step(streams)
    run = graph.replay if graph is not None else (lambda: w.step(streams))
    for _ in range(warmup):
        run()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, graph is not None


def kernel_durations(w, launches):
    """Average pure device duration (s) of the workload's kernel, per-dispatch events.  The launches are issued eagerly
    from Python; after an idle gap (or while the host is still busy with the CPU baseline's threads) the device clocks
    down between them and a 6 us kernel reads 9 us.  So: an untimed burst first, then two timed passes, and the pass
    with the lower AVERAGE is reported (every figure is the mean over all dispatches of one pass); `pass_avgs_us` keeps
    BOTH averages on the line, so the selection is visible (a min-of-2, not a mean-of-2)."""
    s = torch.cuda.current_stream(w.dev).cuda_stream
    best, avgs = None, []
    for _ in range(2):
        for i in range(min(launches, 64)):
            w.launch(i, s)
        torch.cuda.synchronize(w.dev)
        with L.kernel_timer(launches) as kt:
            for i in range(launches):
                w.launch(i, s)
        torch.cuda.synchronize(w.dev)
        ms = sorted(kt.ms)
        if not ms:
            return None
        r = {"avg_s": sum(ms) / len(ms) * 1e-3, "min_s": ms[0] * 1e-3, "median_s": ms[len(ms) // 2] * 1e-3, "n": len(ms)}
        avgs.append(round(r["avg_s"] * 1e6, 3))
        if best is None or r["avg_s"] < best["avg_s"]:
            best = r
    best["pass_avgs_us"] = avgs
    return best


_PROBE = None
PROBE_ABI = 2   # tools/ceiling_probe.hip probe_abi_version(): 2 = probe_read takes (buf, sink, bytes, blocks, in_flight, stream)


def _probe_lib():
    """tools/libceiling_probe.so (not part of the product): register-only fp8 MFMA loop + streaming-read kernel."""
    global _PROBE
    if _PROBE is None:
        so = os.path.join(ROOT, "tools", "libceiling_probe.so")
        # Never compiled from here: by now the GPU is initialised (and under rocprofv3 the profiler's preload is in the
        # environment), and hipcc is a launcher that execs clang - an exec hop this pool forbids.  __graft_entry__.build()
        # and tools/profile_round.sh build it; a missing probe is reported, not repaired.
        if not os.path.exists(so):
            raise RuntimeError("tools/libceiling_probe.so not built (python -c 'import __graft_entry__ as g; g.build()')")
        src = os.path.join(ROOT, "tools", "ceiling_probe.hip")
        if os.path.exists(src) and os.path.getmtime(so) < os.path.getmtime(src):
            # a probe left over from an older source has another C signature: calling it would hand it an int as its stream
            raise RuntimeError("tools/libceiling_probe.so is older than tools/ceiling_probe.hip: rebuild it (__graft_entry__.build())")
        lib = ctypes.CDLL(so)
        try:   # the probe states its own ABI revision; bench.py and the probe must agree (a stale .so with a fresh mtime is refused too)
            lib.probe_abi_version.restype = ctypes.c_int
            abi = int(lib.probe_abi_version())
        except AttributeError:
            abi = -1
        if abi != PROBE_ABI:
            raise RuntimeError(f"tools/libceiling_probe.so has probe ABI {abi}, bench.py expects {PROBE_ABI}: rebuild it")
        vp = ctypes.c_void_p
        lib.probe_mfma.restype = ctypes.c_int
        lib.probe_mfma.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        lib.probe_mfma_flops.restype = ctypes.c_double
        lib.probe_mfma_flops.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.probe_read.restype = ctypes.c_int
        lib.probe_read.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp]
        _PROBE = lib
    return _PROBE


def measure_ceilings(dev, info):
    """What THIS device sustains, measured in this run (each >= 50 ms of work):
      mfma_TFLOPs        fp8 MFMA (the GEMM's instruction) back to back from registers on weight-like bytes, every CU busy
      read_large_GBs     streaming read of a 2 GiB buffer (cold: larger than the 256 MiB Infinity Cache)
      read_56MiB_GBs     a 56 MiB buffer read once per launch, 6 rotating buffers (the size of config C2's weights):
                         includes the launch ramp a 10 us kernel cannot amortise.
    Both in the launch shape the round-3 sweep found fastest (tools/probes/read_sweep.hip)."""
    try:
        lib = _probe_lib()
        st = torch.cuda.current_stream(dev)
        gen = torch.Generator(device=dev).manual_seed(7)
        ops = clean_bytes((32 * 1024,), dev, gen)
        blocks, waves, iters = 2 * info["compute_units"], 8, 20000
        out = torch.empty(blocks * waves * 64 * 4, dtype=torch.float32, device=dev)

        def timed(fn, reps):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st)
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e-3 / reps

        t = timed(lambda: lib.probe_mfma(ops.data_ptr(), out.data_ptr(), blocks, waves, iters, st.cuda_stream), 12)
        mfma = lib.probe_mfma_flops(blocks, waves, iters) / t / 1e12
        sink = torch.zeros(1, dtype=torch.int32, device=dev)
        big = torch.ones(2 << 30, dtype=torch.uint8, device=dev)
        # (grid, loads in flight per lane) = the fastest of the round-3 sweep per buffer size (profiles/r03_read_sweep.txt); round 2's single
        # point (2048 / 4096 workgroups, 4 in flight) read 10-20 % below these
        t = timed(lambda: lib.probe_read(big.data_ptr(), sink.data_ptr(), big.numel(), 256, 8, st.cuda_stream), 8)
        read_large = big.numel() / t / 1e9
        del big
        n56 = 56 << 20
        bufs = [torch.ones(n56, dtype=torch.uint8, device=dev) for _ in range(6)]
        it = [0]

        def rot():
            lib.probe_read(bufs[it[0] % 6].data_ptr(), sink.data_ptr(), n56, 2048, 1, st.cuda_stream)
            it[0] += 1
        t = timed(rot, 600)
        read_small = n56 / t / 1e9
        del bufs
        torch.cuda.empty_cache()
        return {"mfma_TFLOPs": round(mfma, 1), "read_large_GBs": round(read_large, 1), "read_56MiB_GBs": round(read_small, 1)}
    except Exception as e:  # the ceilings are context; never let them take the bench line down
        return {"error": repr(e)}


def source_fingerprint():
    """sha256 over the kernel sources: profiles/pmc_traffic.json is only valid for the kernels it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def roofline_of(w, kd, info, traffic_and_source, ceilings=None):
    if kd is None:
        return None
    if w.unit_flops:
        peak = info["compute_units"] * (info["clock_khz"] / 1e6) * FP8_FLOP_PER_CLK_CU / 1e3  # TFLOP/s
        ach = w.flops / kd["avg_s"] / 1e12
        r = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s"}
    else:
        ach = w.bytes / kd["avg_s"] / 1e9
        r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    r["frac"] = round(r["achieved"] / r["peak"], 4)
    if w.unit_flops:  # C3 sits at the ridge: also say how far the same launch is from the HBM roof
        r["hbm_achieved_GBs"] = round(w.bytes / kd["avg_s"] / 1e9, 1)
        r["hbm_frac"] = round(r["hbm_achieved_GBs"] / HBM_PEAK_GBS, 4)
    # informational: what this device sustained in isolation in THIS run (measure_ceilings)
    c = ceilings or {}
    r["measured_ceiling"] = c.get("mfma_TFLOPs") if w.unit_flops else c.get("read_large_GBs")
    traffic, traffic_source = traffic_and_source if traffic_and_source else (None, None)
    r["traffic"] = traffic
    # where the figure comes from: NOT measured in this run - the PMC passes of tools/profile_round.sh (separate rocprofv3
    # --pmc FETCH_SIZE / WRITE_SIZE runs, x2 gfx950 FETCH correction) on the builder's box, stamped with the fingerprint of
    # the kernel sources they were taken on; null when the sources running now differ
    r["traffic_source"] = traffic_source
    r["kernel_avg_us"] = round(kd["avg_s"] * 1e6, 3)
    if "source" in kd:
        r["kernel_avg_source"] = kd["source"]
    r["kernel_min_us"] = round(kd["min_s"] * 1e6, 3)
    r["kernel_pass_avgs_us"] = kd.get("pass_avgs_us")   # kernel_avg_us is the LOWER of these two passes
    r["kernel_launches_timed"] = kd["n"]
    r["algorithmic_per_launch"] = w.flops if w.unit_flops else w.bytes
    return r


FLOOR_IDS = {"gemm": (901, 902, 903), "wide": (911, 912, 913)}   # tools/floor_probe.hip: (DMA stream only, launch + C store only, launch only)
FLOOR_PROBE_ABI = 1


def measure_floor(w, launches=128):
    """`roofline.floor`: what ONE dispatch of this workload's kernel costs on this box before it multiplies anything - timing-only forms of the
    SAME kernel source (tools/floor_probe.hip compiles fp8mi_gemm.hip with FP8MI_FLOOR_PROBE), timed like the real kernel: per-dispatch
    start/stop events, rotating weight buffers, untimed burst first, the lower of two pass averages.
      empty_launch_us   the kernel returns behind its argument loads (same grid, block and LDS allocation)
      c_store_only_us   no K loop: launch, arguments, tile map, the fused epilogue's store of the whole C, kernel end
      dma_only_us       the K loop's LDS-DMA stream with its waits and barriers, no fragment reads, no MFMAs (+ everything above)
    The real kernel hides its MFMAs and fragment reads under that stream, so dma_only_us is this design's attainable floor."""
    if w.name not in FLOOR_IDS or w.sharded:
        return None
    try:
        so = os.path.join(ROOT, "tools", "libfloor_probe.so")
        src = os.path.join(ROOT, "tools", "floor_probe.hip")
        if not os.path.exists(so):
            return {"error": "tools/libfloor_probe.so not built (python -c 'import __graft_entry__ as g; g.build()')"}
        newest = max(os.path.getmtime(f) for f in [src] + [os.path.join(PKG, "csrc", n) for n in ("fp8mi_gemm.hip", "fp8mi_gemm_epi.h", "fp8mi_common.h")])
        if os.path.getmtime(so) < newest:
            return {"error": "tools/libfloor_probe.so is older than the kernel sources it compiles: rebuild it (__graft_entry__.build())"}
        lib = ctypes.CDLL(so)
        lib.floor_probe_abi_version.restype = ctypes.c_int
        if int(lib.floor_probe_abi_version()) != FLOOR_PROBE_ABI:
            return {"error": "tools/libfloor_probe.so has another probe ABI: rebuild it"}
        vp = ctypes.c_void_p
        lib.floor_probe_run.restype = ctypes.c_int
        lib.floor_probe_run.argtypes = [ctypes.c_int, vp, ctypes.POINTER(vp), ctypes.c_int, vp, vp, vp, ctypes.c_longlong, ctypes.c_longlong,
                                        ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float)]
        nb = len(w.Bs)
        Bs = (vp * nb)(*[b.data_ptr() for b in w.Bs])
        st = torch.cuda.current_stream(w.dev).cuda_stream
        out = (ctypes.c_float * launches)()

        def avg_us(kid):
            best = None
            for _ in range(2):
                rc = lib.floor_probe_run(kid, w.A.data_ptr(), Bs, nb, w.Cs[0].data_ptr(), w.sa.data_ptr(), w.sb.data_ptr(), w.M, w.N, w.K,
                                         w.code, min(launches, 64), st, out)   # untimed burst (clocks up)
                if rc:
                    raise RuntimeError(f"floor_probe_run({kid}) -> {rc}")
                rc = lib.floor_probe_run(kid, w.A.data_ptr(), Bs, nb, w.Cs[0].data_ptr(), w.sa.data_ptr(), w.sb.data_ptr(), w.M, w.N, w.K,
                                         w.code, launches, st, out)
                if rc:
                    raise RuntimeError(f"floor_probe_run({kid}) -> {rc}")
                a = sum(out[i] for i in range(launches)) / launches * 1e3
                best = a if best is None or a < best else best
            return round(best, 3)
        dma, store, empty = FLOOR_IDS[w.name]
        res = {"empty_launch_us": avg_us(empty), "c_store_only_us": avg_us(store), "dma_only_us": avg_us(dma),
               "method": "timing-only forms of the same kernel source (tools/floor_probe.hip, Cfg::FLOOR), per-dispatch events, "
                         f"{launches} launches over the workload's rotating weight buffers, lower of two pass averages",
               "launches_timed": launches}
        torch.cuda.synchronize(w.dev)
        return res
    except Exception as e:   # context, never a reason to lose the bench line
        return {"error": repr(e)}


def cpu_baseline(w, budget_s=12.0):
    """The C oracle (OpenMP) on the host cores, bounded sample of the same workload."""
    so = os.path.join(ORACLE, "libfp8_oracle.so")
    if not os.path.exists(so):   # built by __graft_entry__.build(); never from a process that holds the GPU
        return {"error": "oracle/libfp8_oracle.so not built (python -c 'import __graft_entry__ as g; g.build()')"}
    o = ctypes.CDLL(so)
    o.fp8o_num_threads.restype = ctypes.c_int
    cores = int(o.fp8o_num_threads())
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    import numpy as np
    rng = np.random.default_rng(1234)
    if w.name in MM_WORKLOADS:
        K, N = w.K, min(w.N_total, 4096)
        B = rng.integers(0, 127, size=(N, K), dtype=np.uint8)
        sa = np.array([0.01], np.float32)

        def run(rows):
            A = rng.integers(0, 127, size=(rows, K), dtype=np.uint8)
            C = np.empty((rows, N), np.float32)
            t0 = time.perf_counter()
            o.fp8o_scaled_mm(A.ctypes.data_as(vp), B.ctypes.data_as(vp), C.ctypes.data_as(vp), sa.ctypes.data_as(vp),
                             sa.ctypes.data_as(vp), sz(rows), sz(N), sz(K), sz(1), sz(1))
            return time.perf_counter() - t0

        rows = w.M if w.M <= 512 else 512   # a 512-row slab of larger problems
        run(rows)                            # warm-up (page faults, OpenMP pool)
        t = run(rows)
        reps = int(min(400, max(1, round(budget_s / max(t, 1e-6)))))
        tt = sum(run(rows) for _ in range(reps))
        if w.unit_flops:
            val, unit = 2.0 * rows * N * K * reps / tt / 1e12, "TFLOP/s"
        else:
            val, unit = (N * K + K + 4 * N) * reps / tt / 1e9, "GB/s"
        sample = f"{reps} x fp8o_scaled_mm (oracle/fp8_oracle.c, OpenMP) on M={rows} of {w.M} rows, K={K}, N={N}; {tt:.1f} s"
        # the torch-CPU variant of the same restatement - LUT decode of both operands + float32 matmul,
        # every call (the reference's CPU path re-dequantises the weights per call), all host threads
        try:
            import fp8_oracle
            torch.set_num_threads(os.cpu_count() or 1)
            lut = torch.from_numpy(fp8_oracle.decode_lut())
            At = torch.from_numpy(rng.integers(0, 127, size=(rows, K), dtype=np.uint8)).long()
            Bt = torch.from_numpy(B).long()

            def run_t():
                t0 = time.perf_counter()
                (lut[At] @ lut[Bt].t()) * 0.01 * 0.01
                return time.perf_counter() - t0
            run_t()
            t1 = run_t()
            reps_t = int(min(200, max(1, round(budget_s / 3 / max(t1, 1e-6)))))
            tt_t = sum(run_t() for _ in range(reps_t))
            v_t = (2.0 * rows * N * K * reps_t / tt_t / 1e12) if w.unit_flops else ((N * K + K + 4 * N) * reps_t / tt_t / 1e9)
            torch_lut = {"value": float(f"{v_t:.4g}"), "unit": unit, "cores": torch.get_num_threads(),
                         "sample": f"{reps_t} x (LUT[A] @ LUT[B].T) * sa * sb in torch float32 on the same shape; {tt_t:.1f} s"}
        except Exception as e:
            torch_lut = {"error": repr(e)}
    else:
        n = 1 << 26
        if w.name == "quantize":
            x = (rng.standard_normal(n) * 16).astype(np.float32)
            out = np.empty(n, np.uint8)
            fn = lambda: o.fp8o_encode(x.ctypes.data_as(vp), out.ctypes.data_as(vp), sz(n))
            per = 5.0
        else:
            x = rng.integers(0, 127, size=n, dtype=np.uint8)
            out = np.empty(n, np.uint16)
            fn = lambda: o.fp8o_dequant_f16_bits(x.ctypes.data_as(vp), out.ctypes.data_as(vp), sz(n))
            per = 3.0
        fn()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < budget_s / 2 and reps < 50:
            fn()
            reps += 1
        tt = time.perf_counter() - t0
        val, unit = per * n * reps / tt / 1e9, "GB/s"
        sample = f"{reps} x 2^26 elements of the 2^30 (oracle/fp8_oracle.c, OpenMP); {tt:.1f} s"
    res = {"value": float(f"{val:.4g}"), "unit": unit, "cores": cores, "kind": "port", "sample": sample,
           "host_cpu": _cpu_model(), "host_logical_cpus": os.cpu_count()}
    if w.name in MM_WORKLOADS:
        res["torch_lut"] = torch_lut
    return res


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def load_traffic(name):
    """HBM-side bytes per launch from the PMC passes (tools/profile_round.sh -> profiles/pmc_traffic.json), or None when
    that file was measured on other kernel sources than the ones running now (it carries their fingerprint)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(p))
        sha = source_fingerprint()
        if d.get("source_sha") != sha:
            return None, {"file": "profiles/pmc_traffic.json", "status": "stale: measured on kernel sources %s, running %s" % (d.get("source_sha"), sha)}
        return d.get("traffic", {}).get(name), {"file": "profiles/pmc_traffic.json", "source_sha": sha, "measured_in_this_run": False,
                                                 "method": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, tools/profile_round.sh", "box": d.get("box")}
    except Exception as e:
        return None, {"file": "profiles/pmc_traffic.json", "status": "unreadable: %r" % (e,)}


def allgather_only(w, reps=10):
    """The collective of the sharded linear alone (same chunk shapes, same process group): per-rank ingress rate."""
    try:
        dev, world = w.dev, dist.get_world_size()
        nc = w.N // w.chunks
        part = torch.zeros(nc, w.M, dtype=w.out_dtype, device=dev)
        block = torch.empty(world * nc, w.M, dtype=w.out_dtype, device=dev)
        for _ in range(3):
            dist.all_gather_into_tensor(block, part)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dist.all_gather_into_tensor(block, part)
        e1.record()
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) * 1e3 / reps
        t = torch.tensor([us], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        us = float(t.item())
        recv = (world - 1) * part.numel() * part.element_size()
        return {"calls_per_linear": w.chunks, "bytes_received_per_call": recv, "avg_us_per_call": round(us, 2),
                "ingress_GBs_per_rank": round(recv / (us * 1e-6) / 1e9, 1) if recv else 0.0}
    except Exception as e:  # never let the extra measurement take the bench line down
        return {"error": repr(e)}


def dtype_of(name):
    """What the workload's kernel computes in (not a precision claim)."""
    if name in ("quantize", "quantize_rne", "dequant"):
        return "u8"
    if name == "gemv_c1":
        return "fp8_e4m3fn (M = 1, K <= 4096: decoded products summed in IEEE fp32 FMA, as fp8_matmul.metal:177-199)"
    if name in ("gemv", "gemv_sq"):
        # since round 2 the vec-mat hands K > 4096 to the matrix core too (MFMA-diagonal form); FP8MI_KERNEL_GEMV_FP32 keeps IEEE fp32
        return "fp8_e4m3fn (M = 1, K > 4096: products summed by the fp8 matrix core into fp32, not IEEE fp32 FMA as fp8_matmul.metal:177-199)"
    return "fp8_e4m3fn (fp8 MFMA into fp32 accumulators)"


def callsite_eager(dev, info):
    """The reference's own performance test, restated (test_fp8_metal.py:221-315, README.md:75-86): perf_counter around
    synchronize, warmup 5, 20 iterations, EAGER calls (no graph) at its three shapes + C3 - through the surface a ComfyUI
    call site uses: the PATCHED torch._scaled_mm on float8_e4m3fn tensors with device scales.  Per shape: host wall per
    call, the kernel's device time over the same calls (per-dispatch events), their difference (what the Python / ctypes /
    launch path adds on top of the kernel when calls are issued back to back), and the pure issue cost per call (host
    time until the call returns, GPU left to run behind).  `reference_form` is the call exactly as the reference's
    test makes it - op-level fp8_scaled_mm on uint8 tensors with CPU scale tensors (moved to the device per call)."""
    import fp8_mps_patch
    import fp8_mi355x_native as native
    gen = torch.Generator(device=dev).manual_seed(4321)
    out = {"method": "time.perf_counter around torch.cuda.synchronize; warmup 5; 20 iterations as the reference and 200 for a stable mean; "
                     "eager (no HIP graph); one weight buffer per shape as in the reference's test"}
    shapes = [("single_token_4096", 1, 4096, 4096), ("single_token_14336", 1, 14336, 14336), ("batch4_4096", 4, 4096, 4096),
              ("c3_512x4096x4096", 512, 4096, 4096)]
    fp8_mps_patch.install()
    try:
        for label, M, K, N in shapes:
            a8 = clean_bytes((M, K), dev, gen)
            b8 = clean_bytes((N, K), dev, gen)
            A = a8.view(torch.float8_e4m3fn)
            Bt = b8.view(torch.float8_e4m3fn).t()          # (K, N) column-major, as torch._scaled_mm requires
            sa = torch.full((1,), 0.01, dtype=torch.float32, device=dev)
            sb = torch.full((1,), 0.01, dtype=torch.float32, device=dev)
            sa_cpu, sb_cpu = torch.tensor([0.01]), torch.tensor([0.01])

            def call_patch():
                return torch._scaled_mm(A, Bt, scale_a=sa, scale_b=sb, out_dtype=torch.float32)

            def call_ref_form():
                return native.fp8_scaled_mm(a8, b8, sa_cpu, sb_cpu)

            def wall(fn, iters):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                t1 = time.perf_counter()
                torch.cuda.synchronize(dev)
                t2 = time.perf_counter()
                return (t2 - t0) / iters * 1e6, (t1 - t0) / iters * 1e6

            h20, _ = wall(call_patch, 20)
            h200, issue = wall(call_patch, 200)
            with L.kernel_timer(200) as kt:
                for _ in range(200):
                    call_patch()
            torch.cuda.synchronize(dev)
            k_us = sum(kt.ms) / max(len(kt.ms), 1) * 1e3
            r20, _ = wall(call_ref_form, 20)
            out[label] = {"M": M, "K": K, "N": N, "host_us_per_call_20": round(h20, 2), "host_us_per_call_200": round(h200, 2),
                          "issue_us_per_call": round(issue, 2), "kernel_us": round(k_us, 2),
                          "overhead_us_over_kernel": round(h200 - k_us, 2),
                          "reference_form_host_us_per_call_20": round(r20, 2)}
            del a8, b8, A, Bt
    finally:
        fp8_mps_patch.uninstall()
    out["reference_published_ms"] = {"hardware": "Apple M4 Pro", "source": "README.md:75-86", "single_token_4096": 0.66,
                                     "single_token_14336": 2.38, "batch4_4096": 1.03}
    torch.cuda.empty_cache()
    return out


def measure(name, dev, steps, warmup, world, rank, kernel, with_cpu, info, nbuf=None, sharded=False, n_streams=1, ceilings=None, gather="rccl"):
    w = Workload(name, dev, world, rank, kernel, nbuf, sharded, gather=gather)
    dt, graphed = time_steps(w, steps, warmup, True, world, n_streams)
    launches = steps * w.inner
    if w.unit_flops:
        value, unit = w.flops * world * launches / dt / 1e12, "TFLOP/s"
    else:
        value, unit = w.bytes * launches / dt / 1e9, "GB/s"
    kd = kernel_durations(w, min(4 * w.inner, 256)) if name != "linear" else None   # the chain is three kernels
    # One launch's share of the replayed graph (kernel + the gap to the next one) is an UPPER bound of the kernel's device time.  Where the eager
    # per-dispatch pass reads above it, that pass was host-bound - a ctypes call takes ~5 us to issue, so behind a 4 us kernel the GPU idles and
    # clocks down between launches (config C1: 8.1 us per dispatch eagerly, 5.3 us per launch in the graph) - and the graph's figure is used.
    graph_us = dt / launches * 1e6
    if kd is not None and not w.sharded and graphed and kd["avg_s"] * 1e6 > graph_us:
        kd = dict(kd, eager_avg_s=kd["avg_s"], avg_s=graph_us * 1e-6, source="graph replay: ms_per_step / launches (kernel + inter-kernel gap); the eager per-dispatch "
                                                                               "pass was host-bound and read %.3f us" % (kd["avg_s"] * 1e6))
    res = {"value": round(value, 3), "unit": unit, "ms_per_step": round(dt / steps * 1e3, 5),
           "launches_per_step": w.inner, "hip_graph": graphed, "streams": n_streams, "dtype": dtype_of(name), "config": w.desc,
           "roofline": roofline_of(w, kd, info, load_traffic(name) if world == 1 and not w.sharded else None, ceilings)}
    if res["roofline"] is not None:
        res["roofline"]["graph_us_per_launch"] = round(graph_us, 3)
    if res["roofline"] is not None and world == 1 and kernel == L.KERNEL_AUTO:
        fl = measure_floor(w)
        if fl is not None:
            if "dma_only_us" in fl and res["roofline"].get("kernel_avg_us"):
                fl["kernel_over_dma_only"] = round(res["roofline"]["kernel_avg_us"] / fl["dma_only_us"], 3)
                fl["frac_at_dma_only"] = round(res["roofline"]["frac"] * res["roofline"]["kernel_avg_us"] / fl["dma_only_us"], 4)   # the roofline fraction this design would read if the kernel ran in dma_only_us
            res["roofline"]["floor"] = fl
    if w.sharded and dist.is_initialized():
        # GEMM-only rate is roofline.achieved, this is the gather alone - the form the step used
        try:
            res["allgather_only"] = dict(peer_gather_only(w), form="peer-store (include/fp8mi_peer.h)") if w.peer is not None else allgather_only(w)
        except Exception as e:
            res["allgather_only"] = {"error": repr(e)}
    if w.peer is not None:
        res["peer_timeout_status"] = w.peer.status()
        w.peer.close()
    if with_cpu:
        res["cpu_baseline"] = cpu_baseline(w)
    del w
    torch.cuda.empty_cache()
    return res


def self_launch(n, argv, dry=False):
    """Start `python -m torch.distributed.run --nproc-per-node n bench.py <argv>` as a child process: what the driver's N > 1 command line
    is, for callers that run plain `python bench.py --gpus N`.  Rendezvous on 127.0.0.1 (the container hostname may not resolve), a free
    port, the environment unchanged apart from HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: RCCL's peer mappings need it on this pool).
    stdout of the children is filtered down to rank 0's ONE JSON line; everything else they print goes to stderr."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if dry:
        print(json.dumps({"launch": cmd, "ranks": n, "env": {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        log("[bench] the ranks exited cleanly but printed no bench line")
        rc = 1
    return rc


def rccl_report(dev, world, backend, debug_file):
    """What the communicator actually is (N > 1): ranks the backend sees after the warm-up all-reduce, the devices behind them, and the
    transport / algorithm lines of RCCL's own bring-up log (NCCL_DEBUG=INFO into a per-process file, parsed by rank 0)."""
    rep = {"backend": backend, "rccl_ranks": dist.get_world_size()}
    try:
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": dist.get_rank(), "device_index": dev.index, "uuid": str(getattr(props, "uuid", "n/a")),
                "pci": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:{getattr(props, 'pci_device_id', 0):02x}"}
        allp = [None] * world
        dist.all_gather_object(allp, mine)
        rep["devices"] = allp
        rep["device_uuids"] = [d["uuid"] for d in allp]
        rep["distinct_devices"] = len({(d["uuid"], d["pci"]) for d in allp})
    except Exception as e:
        rep["devices_error"] = repr(e)
    try:
        if debug_file and os.path.exists(debug_file):
            rep["nccl_debug"] = parse_nccl_debug(open(debug_file, errors="replace").read())
    except Exception as e:
        rep["nccl_debug_error"] = repr(e)
    return rep


def parse_nccl_debug(text):
    """Summary of an NCCL_DEBUG=INFO log: version, transports by kind (`via P2P/...`, `via SHM`, `via NET/...`), channel counts,
    the ring / tree lines of the topology search and any algorithm / protocol choice RCCL printed."""
    import re
    out = {"transports": {}, "lines": 0}
    keep = []
    for ln in text.splitlines():
        out["lines"] += 1
        m = re.search(r"via (P2P/[A-Za-z_/]+|SHM[/A-Za-z_]*|NET/[A-Za-z0-9_/]+|direct[ A-Za-z]*)", ln)
        if m:
            k = m.group(1).strip()
            out["transports"][k] = out["transports"].get(k, 0) + 1
        if re.search(r"(RCCL version|NCCL version|nranks|[0-9]+ coll channels|Connected all (rings|trees)|Ring [0-9]+ :|Trees? \[|"
                     r"Algo|Proto|xgmi|XGMI|nNodes|comm 0x[0-9a-f]+ rank .* Init COMPLETE)", ln) and len(keep) < 24:
            keep.append(ln.strip()[-220:])
    out["selected_lines"] = keep
    return out


def under_profiler():
    """rocprofv3 preloads a library that initialises the GPU before Python starts; a child process started from here would be an exec behind a
    GPU-initialising preload, which this pool forbids."""
    env = os.environ
    return any("rocprof" in env.get(k, "").lower() for k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIB")) or any(k.startswith("ROCPROF") for k in env)


def host_kernarg_child(steps, warmup):
    """`secondary.gemm_host_kernarg`: the headline workload (C3) as a process that does NOT set HIP_FORCE_DEV_KERNARG runs it - what a ComfyUI
    plugin gets, since the knob has to be in the environment before libamdhip64 loads and a plugin is imported long after.
    A fresh child process, started and finished BEFORE this process touches the GPU (no exec behind an initialised GPU, no two benches at once)."""
    import subprocess
    env = dict(os.environ, HIP_FORCE_DEV_KERNARG="0", FP8MI_BENCH_CHILD="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "gemm", "--steps", str(max(3, steps // 2)), "--warmup", str(max(1, warmup // 2)),
           "--no-secondary", "--no-cpu-baseline", "--no-ceilings"]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        for ln in reversed(out.stdout.splitlines()):
            if ln.lstrip().startswith('{"metric"'):
                d = json.loads(ln)
                r = d.get("roofline") or {}
                return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dev_kernarg": d["config"].get("dev_kernarg"),
                        "kernel_avg_us": r.get("kernel_avg_us"), "kernel_min_us": r.get("kernel_min_us"), "frac": r.get("frac"),
                        "launches_per_step": d["config"].get("launches_per_step"), "config": {"workload": d["config"].get("workload")},
                        "note": "same workload and binary as the headline, kernel arguments in HOST memory (HIP_FORCE_DEV_KERNARG=0: the HIP "
                                "runtime's default); run in a child process before this one initialised the GPU"}
        return {"error": f"child printed no bench line (rc {out.returncode}): {out.stderr[-300:]}"}
    except Exception as e:
        return {"error": repr(e)}


LAUNCHER_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                "MASTER_ADDR", "MASTER_PORT", "NCCL_DEBUG_FILE", "NCCL_ASYNC_ERROR_HANDLING", "TORCH_NCCL_ASYNC_ERROR_HANDLING")


PEER_CHILD_TIMEOUT_S = 300   # the other ranks of the run wait in their rendezvous meanwhile, whose own limit is 10 minutes


def peer_store_child(n, steps, warmup, dry=False):
    """`peer_allgather` on the N > 1 line: the sharded FLUX step gathered by the direct all-gather of include/fp8mi_peer.h (every rank stores its
    slab into all peers at once) next to the same step gathered by RCCL, measured by a SEPARATE group of N ranks that rank 0 starts and waits for
    BEFORE this process has touched the GPU (the other ranks of this run wait in their rendezvous meanwhile).  Separate processes on purpose:
    the peer-store path has only ever run with several ranks on ONE GPU (tests/test_gpu_patch.py) - no multi-GPU node was available to the build -
    so whatever it does on a real xGMI fabric (a fault included) must not be able to take the headline measurement down with it."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), "--peer-child", "--gpus", str(n), "--steps", str(max(3, steps // 2)), "--warmup", str(max(1, warmup // 2))]
    if dry:
        return {"launch": cmd}
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV and not k.startswith("TORCHELASTIC")}
    env.update(FP8MI_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        # its own session = its own process group: on a timeout the launcher AND its ranks are killed (exactly the group started here), so that
        # nothing of it is left on the GPUs when the headline measurement begins
        import signal
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            stdout, stderr = proc.communicate(timeout=PEER_CHILD_TIMEOUT_S)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.communicate()
            return {"error": f"the child group did not finish within {PEER_CHILD_TIMEOUT_S} s and was killed"}
        for ln in reversed(stdout.splitlines()):
            if ln.lstrip().startswith('{"peer_allgather"'):
                return json.loads(ln)["peer_allgather"]
        return {"error": f"the child group printed no line (rc {proc.returncode}): {stderr[-400:]}"}
    except Exception as e:
        return {"error": repr(e)}


def choose_gather(report, margin=1.05):
    """-> (use the peer-store gather for the headline step?, why).  Only on evidence from this very node: both forms measured by the child group,
    the peer-store step at least `margin` x the collective's throughput, the same bits on every rank, no bounded wait missed."""
    c, p = report.get("collective") or {}, report.get("peer_store") or {}
    if "value" not in c or "value" not in p:
        return False, "the child group did not measure both forms: " + str(report.get("error") or c.get("error") or p.get("error") or "no values")
    if not p.get("bit_equal_to_collective_on_every_rank"):
        return False, "the peer-store result differed from the collective's on some rank"
    if p.get("timeout_status", 1) != 0:
        return False, f"a bounded wait of the peer-store gather timed out (status {p.get('timeout_status')})"
    if report.get("backend") != "rccl":
        return False, f"rehearsal backend {report.get('backend')}: not a measurement"
    if p["value"] < margin * c["value"]:
        return False, f"peer-store {p['value']} vs collective {c['value']} {c.get('unit', '')}: not faster by {margin}x"
    return True, f"peer-store {p['value']} vs collective {c['value']} {c.get('unit', '')} in the child group on this node, bit-equal, no timeouts"


def peer_canary(dev, world, rank):
    """Before the headline step is put on the peer-store gather IN THIS group of ranks: three small gathers with short bounded waits, the data
    checked, the status word read; the ranks agree (MIN).  False sends every rank back to the collective."""
    ok = True
    try:
        import fp8_peer_gather
        slab = 4096
        pg = fp8_peer_gather.PeerGather(world * slab, dev, timeout_us=2_000_000)
        buf = pg.tensor(torch.uint8)
        for it in range(3):
            buf[rank * slab:(rank + 1) * slab] = (rank * 17 + it * 5 + 1) % 251
            pg.allgather(rank * slab, slab)
            got = buf.view(world, slab)[:, ::512].clone()
            want = torch.tensor([[(r * 17 + it * 5 + 1) % 251] for r in range(world)], dtype=torch.uint8, device=dev).expand(world, got.shape[1])
            ok = ok and bool(torch.equal(got, want))
        ok = ok and pg.status() == 0
        pg.close()
    except Exception as e:
        log(f"[bench] peer-store canary failed on rank {rank}: {e!r}")
        ok = False
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def peer_gather_only(w, reps=10):
    """The peer-store gather alone, slabs of the sharded linear's chunk size (the counterpart of allgather_only)."""
    dev, world = w.dev, w.world
    nc = w.N // w.chunks
    nbytes = nc * w.M * torch.empty(0, dtype=w.out_dtype).element_size()
    off = w.rank * nbytes
    for _ in range(3):
        w.peer.allgather(off, nbytes)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        w.peer.allgather(off, nbytes)
    e1.record()
    torch.cuda.synchronize(dev)
    t = torch.tensor([e0.elapsed_time(e1) * 1e3 / reps], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    us = float(t.item())
    recv = (world - 1) * nbytes
    return {"calls_per_linear": w.chunks, "bytes_received_per_call": recv, "avg_us_per_call": round(us, 2), "launches_per_call": 2,
            "ingress_GBs_per_rank": round(recv / (us * 1e-6) / 1e9, 1), "GBs_per_link": round(nbytes / (us * 1e-6) / 1e9, 1)}


def peer_child_main(args):
    """One rank of the group peer_store_child() starts: the sharded FLUX step with gather = the process group's all-gather, then with the
    peer-store all-gather - same weights, same chunks, same timing harness - and whether the two return the same bits."""
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("FP8MI_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    with stdout_to_stderr():
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
        t0 = torch.zeros(1, device=dev)
        dist.all_reduce(t0)
        torch.cuda.synchronize(dev)
    L.load()
    out = {"ranks": world, "backend": "rccl" if backend == "nccl" else backend, "workload": "flux, N-column-sharded, 2 chunks per rank (as the headline of this line)"}
    ref = None
    for key, gather in (("collective", "rccl"), ("peer_store", "peer")):
        entry = {}
        try:
            w = Workload("flux", dev, world, rank, gather=gather)
            dt, graphed = time_steps(w, args.steps, args.warmup, True, world)
            launches = args.steps * w.inner
            entry = {"value": round(w.flops * world * launches / dt / 1e12, 3), "unit": "TFLOP/s", "ms_per_step": round(dt / args.steps * 1e3, 5),
                     "us_per_linear": round(dt / launches * 1e6, 2), "linears_per_step": w.inner, "hip_graph": graphed}
            y = w.linears[0](w.A, w.sa)
            torch.cuda.synchronize(dev)
            if gather == "rccl":
                ref = y.clone()
                entry["gather_only"] = allgather_only(w)
            else:
                same = torch.tensor([1 if (ref is not None and torch.equal(y, ref)) else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(same, op=dist.ReduceOp.MIN)
                entry["bit_equal_to_collective_on_every_rank"] = bool(same.item())
                entry["gather_only"] = peer_gather_only(w)
                st = torch.tensor([w.peer.status()], dtype=torch.int32, device=dev)
                dist.all_reduce(st, op=dist.ReduceOp.MAX)
                entry["timeout_status"] = int(st.item())   # 0: every bounded wait was met
                w.peer.close()
            del w
            torch.cuda.empty_cache()
        except Exception as e:
            entry["error"] = repr(e)
        out[key] = entry
    if rank == 0:
        print(json.dumps({"peer_allgather": out}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", choices=["auto", "gemm", "gemv", "gemv_c1", "gemv_sq", "flux", "mid", "wide", "skinny", "decode", "linear",
                                                           "quantize", "quantize_rne", "dequant", "callsite"])
    ap.add_argument("--kernel", type=int, default=L.KERNEL_AUTO, help="force an FP8MI_KERNEL_* id")
    ap.add_argument("--nbuf", type=int, default=None, help="override the number of rotating weight buffers "
                    "(1 = weights stay cache-resident; for sensitivity experiments only)")
    ap.add_argument("--data", default="gauss", choices=["gauss", "uniform", "zeros"],
                    help="operand byte distribution (see clean_bytes)")
    ap.add_argument("--streams", type=int, default=1, choices=range(1, MAX_STREAMS + 1),
                    help="issue the step's independent launches round-robin over this many streams "
                         "(parallel branches of the captured graph); 1 = one serial chain (default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the in-run MFMA / streaming-read ceiling probes")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-host-kernarg", action="store_true", help="skip secondary.gemm_host_kernarg (a child process that runs C3 with kernel arguments in host memory)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="rehearse the multi-GPU code path (sharded linear + RCCL all-gather) with a 1-rank group")
    ap.add_argument("--peer-child", action="store_true", help="internal: one rank of the group that measures the peer-store all-gather (peer_store_child)")
    ap.add_argument("--no-peer-store", action="store_true", help="N > 1: skip the separate group that measures the peer-store all-gather")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE: print the torch.distributed.run command bench.py would start, as JSON, and exit")
    args = ap.parse_args()
    global DATA_MODE
    DATA_MODE = args.data

    # `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks HERE (one process per GPU over RCCL), before
    # anything has touched the GPU (a child process, never an exec), pass rank 0's line through and exit with the child's code.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, [a for a in sys.argv[1:] if a != "--dry-launch"], dry=args.dry_launch))
    if args.dry_launch:
        print(json.dumps({"launch": None, "note": "nothing to launch: --gpus 1, or WORLD_SIZE is set (a launcher is already around bench.py)"}))
        return

    if args.peer_child:
        return peer_child_main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    peer_store = None
    if (world > 1 and rank == 0 and args.workload in ("auto", "flux") and not args.no_secondary and not args.no_peer_store
            and os.environ.get("FP8MI_BENCH_CHILD") != "1"):
        peer_store = {"skipped": "under a profiler preload: no child processes"} if under_profiler() else peer_store_child(world, args.steps, args.warmup)
    host_kernarg = None
    if (world == 1 and args.workload == "auto" and not args.no_secondary and not args.force_sharded and not args.no_host_kernarg
            and os.environ.get("FP8MI_BENCH_CHILD") != "1" and os.environ.get("HIP_FORCE_DEV_KERNARG") == "1"):
        host_kernarg = {"skipped": "under a profiler preload: no child processes"} if under_profiler() else host_kernarg_child(args.steps, args.warmup)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product has no CPU path")
    # FP8MI_BENCH_BACKEND=gloo lets several ranks share ONE GPU to rehearse the N > 1 code path (not a measurement)
    backend = os.environ.get("FP8MI_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        nccl_debug_file = None
        if backend == "nccl" and "NCCL_DEBUG" not in os.environ:
            # RCCL's own account of the communicator (transports, channels, rings) into a per-process file that rank 0 parses onto the line
            import tempfile
            nccl_debug_file = os.path.join(tempfile.gettempdir(), f"fp8mi_bench_nccl_{os.getpid()}.log")
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,ENV,TUNING")
            os.environ["NCCL_DEBUG_FILE"] = nccl_debug_file
        with stdout_to_stderr():   # (RCCL's version banner goes to stdout when the communicator comes up)
            dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
            t0 = torch.zeros(1, device=dev)
            dist.all_reduce(t0)    # forces the communicator (and its banner) now
            torch.cuda.synchronize(dev)
            comm_report = rccl_report(dev, world, backend, nccl_debug_file)
            comm_report["warmup_allreduce_sum"] = float(t0.item())   # = 0: the collective ran over every rank and returned
    else:
        comm_report = None
    if args.gpus != world and rank == 0:
        log(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world} (set by the launcher around bench.py); using WORLD_SIZE")
    L.load()
    info = L.device_info(dev.index)

    primary = args.workload
    if primary == "auto":
        primary = "gemm" if world == 1 and not args.force_sharded else "flux"
    if world > 1 and primary != "flux":
        raise SystemExit("multi-GPU runs shard the FLUX linear (configs[3]); use --workload flux or auto")

    if primary == "callsite":   # only the eager call-site timing (tools / quick checks; not a driver configuration)
        print(json.dumps({"callsite_eager": callsite_eager(dev, info)}), flush=True)
        return
    ceilings = measure_ceilings(dev, info) if (world == 1 and not args.force_sharded and not args.no_ceilings) else None
    # Which gather the headline step uses at N > 1: RCCL's collective unless the separate group of ranks (peer_store_child) has just measured the
    # peer-store gather on THIS node as clearly faster, bit-equal on every rank and with every bounded wait met.  Rank 0 knows; the others are told.
    gather, gather_why = "rccl", None
    if world > 1:
        choice = torch.zeros(1, dtype=torch.int32, device=dev)
        if rank == 0 and peer_store is not None:
            use, gather_why = choose_gather(peer_store)
            choice[0] = 1 if use else 0
        dist.broadcast(choice, src=0)
        gather = "peer" if int(choice.item()) == 1 else "rccl"
        forced = os.environ.get("FP8MI_BENCH_GATHER")   # "peer" / "rccl": rehearsals and A/B runs (the same on every rank: it is the environment)
        if forced in ("peer", "rccl"):
            gather, gather_why = forced, "forced by FP8MI_BENCH_GATHER"
        if gather == "peer" and not peer_canary(dev, world, rank):
            gather, gather_why = "rccl", (gather_why or "") + "; but the canary gathers of THIS group of ranks failed: back to the collective"
    res = measure(primary, dev, args.steps, args.warmup, world, rank, args.kernel,
                  with_cpu=(world == 1 and rank == 0 and not args.no_cpu_baseline and not args.force_sharded and primary != "linear"),
                  info=info, nbuf=args.nbuf, sharded=args.force_sharded, n_streams=args.streams, ceilings=ceilings, gather=gather)
    same_workload_1gpu = None
    if world > 1:
        # the N = 1 bench line is C3 (BASELINE.json's single-GPU config); for a like-for-like strong-scaling
        # reference every rank also times the WHOLE FLUX linear on its own GPU (no sharding, no collective)
        try:
            ref = measure("flux", dev, max(3, args.steps // 2), max(1, args.warmup // 2), 1, 0, args.kernel,
                          with_cpu=False, info=info)
            same_workload_1gpu = {"value": ref["value"], "unit": ref["unit"], "ms_per_step": ref["ms_per_step"],
                                  "launches_per_step": ref["launches_per_step"], "hip_graph": ref["hip_graph"],
                                  "kernel_avg_us": (ref["roofline"] or {}).get("kernel_avg_us")}
        except Exception as e:
            same_workload_1gpu = {"error": repr(e)}
        dist.barrier()
    line = {
        "metric": METRIC, "value": res["value"], "unit": res["unit"], "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
        "scaling": "strong" if primary == "flux" else "weak", "vs_baseline": None,
        # what the path computes in: e4m3fn products summed by the gfx950 fp8 matrix core into fp32 accumulators (its
        # in-group alignment is not IEEE fp32 addition: DESIGN.md 2) / the fp32 VALU for M = 1; bytes for the casts
        "dtype": dtype_of(primary),
        "data": {"gauss": "synthetic (seeded N(0,1) amax-quantised to e4m3fn; weights rotate through > 256 MiB)",
                 "uniform": "synthetic (seeded uniform e4m3 bytes, NaN patterns remapped; weights rotate through > 256 MiB)",
                 "zeros": "synthetic (all-zero bytes; clock upper bound, not a reportable number)"}[args.data],
        "config": dict(res["config"], launches_per_step=res["launches_per_step"], hip_graph=res["hip_graph"],
                       parallelism=("N-column-sharded x%d, 2 chunk-cyclic row chunks per rank, %s "
                                    "pipelined under the GEMM (fp8_sharded_linear.py)" % (world, "peer-store all-gather (include/fp8mi_peer.h)" if gather == "peer" else "RCCL all-gather")) if world > 1 else "single GPU",
                       device=info["name"], arch=info["arch"], compute_units=info["compute_units"],
                       dev_kernarg=os.environ.get("HIP_FORCE_DEV_KERNARG", "0") == "1"),
        "roofline": res["roofline"],
    }
    if "allgather_only" in res:
        line["allgather_only"] = res["allgather_only"]
    if comm_report is not None:
        line["communicator"] = comm_report
        line["rccl_ranks"] = comm_report.get("rccl_ranks")
        line["device_uuids"] = comm_report.get("device_uuids")
    if same_workload_1gpu is not None:
        line["same_workload_on_one_gpu"] = same_workload_1gpu
    if peer_store is not None:
        line["peer_allgather"] = dict(peer_store, headline_gather=gather, headline_gather_reason=gather_why)
    if "peer_timeout_status" in res:
        line["peer_timeout_status"] = res["peer_timeout_status"]
    if "cpu_baseline" in res:
        line["cpu_baseline"] = res["cpu_baseline"]
    if ceilings is not None:
        line["measured_ceilings"] = ceilings

    if world == 1 and args.workload == "auto" and not args.no_secondary and not args.force_sharded:
        sec = {}
        for name in ("gemv", "gemv_c1", "gemv_sq", "flux", "mid", "wide", "skinny", "decode", "linear", "quantize", "quantize_rne", "dequant"):
            try:
                r = measure(name, dev, max(3, args.steps // 2), max(1, args.warmup // 2), 1, 0, L.KERNEL_AUTO,
                            with_cpu=(name == "gemv" and not args.no_cpu_baseline), info=info, ceilings=ceilings)
                if name == "gemv_sq":   # the reference's own number for this shape, on its own hardware (not a baseline for MI355X)
                    r["reference_published"] = {"ms": 2.38, "hardware": "Apple M4 Pro", "source": "README.md:80 / test_fp8_metal.py:233-235"}
                sec[name] = r
            except Exception as e:  # a secondary failure must not hide the primary number
                sec[name] = {"error": repr(e)}
        try:
            sec["callsite_eager"] = callsite_eager(dev, info)
        except Exception as e:
            sec["callsite_eager"] = {"error": repr(e)}
        if host_kernarg is not None:
            sec["gemm_host_kernarg"] = host_kernarg
        line["secondary"] = sec

    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
