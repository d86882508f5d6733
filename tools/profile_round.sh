#!/bin/bash
# Round profile on the GPU box: kernel trace of the default bench + HBM traffic counters per workload.
#   tools/profile_round.sh <tag>        (writes gpurun_out/prof_<tag>/...)
set -uo pipefail
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}; O="$R/gpurun_out/prof_$tag"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1   # what bench.py runs with
# the ceiling probe is built HERE, outside the profiler (bench.py never compiles: hipcc execs clang, an exec hop behind a GPU-initialising preload)
{ [ -f tools/libceiling_probe.so ] && [ ! tools/ceiling_probe.hip -nt tools/libceiling_probe.so ]; } || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/ceiling_probe.hip -o tools/libceiling_probe.so
{ [ -f tools/libfloor_probe.so ] && [ ! tools/floor_probe.hip -nt tools/libfloor_probe.so ] && [ ! fp8-mps-metal_amd/csrc/fp8mi_gemm.hip -nt tools/libfloor_probe.so ]; } || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -fvisibility=hidden -std=c++17 tools/floor_probe.hip -o tools/libfloor_probe.so
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o kt -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/kt.err" || { echo "kernel-trace run failed"; tail -5 $O/kt.err; exit 1; }
python tools/summarize_prof.py "$O/kt" > "$O/kernel_trace_summary.txt"
for w in gemm gemv gemv_c1 gemv_sq flux mid wide skinny decode quantize quantize_rne dequant; do
  n=6; [ $w = quantize ] && n=3; [ $w = quantize_rne ] && n=3; [ $w = dequant ] && n=3
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch_$w" -o p -- python tools/run_workload.py $w $n > /dev/null 2>&1 || { echo "FETCH pass failed for $w"; exit 1; }
  timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write_$w" -o p -- python tools/run_workload.py $w $n > /dev/null 2>&1 || { echo "WRITE pass failed for $w"; exit 1; }
done
python - <<PY
import csv, glob, json, os, collections
O = "$O"
out = {}; lines = []
for w in ("gemm", "gemv", "gemv_c1", "gemv_sq", "flux", "mid", "wide", "skinny", "decode", "quantize", "quantize_rne", "dequant"):
    vals = {}
    for kind in ("fetch", "write"):
        f = glob.glob(os.path.join(O, f"{kind}_{w}", "**", "*counter_collection.csv"), recursive=True)[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            # the workload's own kernel only (the synthetic-data generator also runs amax / encode kernels)
            want = {"gemm": ("gemm_kernel",), "flux": ("gemm256_kernel", "gemm_kernel"), "mid": ("gemm256_kernel", "gemm_kernel"), "wide": ("gemm_kernel",), "decode": ("gemm_kernel",), "gemv": ("gemv_mx_kernel", "gemv_kernel"), "gemv_c1": ("gemv_kernel", "gemv_mx_kernel"),
                    "gemv_sq": ("gemv_kernel",), "skinny": ("gemv_mx_kernel", "skinny_kernel"), "quantize": ("encode_kernel<0, 0, false>",),
                    "quantize_rne": ("encode_kernel<0, 1, false>",), "dequant": ("dequant_kernel",)}[w]
            if any(x in r["Kernel_Name"] for x in want):
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        k, v = max(agg.items(), key=lambda kv: len(kv[1]))
        vals[kind] = (sum(v) / len(v), k)
    fetch_kb, write_kb = vals["fetch"][0], vals["write"][0]
    # MI355X_MICROARCH.md HBM: on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads -> x2;
    # WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Units: KiB.
    traffic = int((2 * fetch_kb + write_kb) * 1024)
    out[w] = traffic
    lines.append(f"{w:9s} FETCH_SIZE {fetch_kb:12.1f} KiB (x2 gfx950 correction)  WRITE_SIZE {write_kb:12.1f} KiB  -> traffic/launch {traffic:14d} B   [{vals['fetch'][1][:70]}]")
import sys
sys.path.insert(0, "$R")
import bench   # the fingerprint of the kernel sources these numbers belong to (bench.py prints traffic only when it matches)
import socket, subprocess
try:
    gpu = subprocess.run(["rocm-smi", "--showproductname"], capture_output=True, text=True, timeout=20).stdout
    gpu = next((l.split(":")[-1].strip() for l in gpu.splitlines() if "Card Series" in l or "Card series" in l), "")
except Exception:
    gpu = ""
json.dump({"source_sha": bench.source_fingerprint(), "box": {"host": socket.gethostname(), "gpu": gpu, "dev_kernarg": os.environ.get("HIP_FORCE_DEV_KERNARG", "")},
           "traffic": out}, open(os.path.join(O, "pmc_traffic.json"), "w"), indent=1)
open(os.path.join(O, "pmc_traffic_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
head -20 "$O/kernel_trace_summary.txt"
