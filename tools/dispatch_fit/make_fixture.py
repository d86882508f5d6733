"""tests/golden/dispatch_times_cold_*.json from the raw output of tools/sweep_regret.py (round 4, COLD weights: >= 320 MiB in rotation for every shape,
tools/r4_cold.sh).  Per shape: the median over its repeats of each kernel's time (us)."""
import glob, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from data import load_raw
R = os.path.join(ROOT, "gpurun_out", "r4c2")
G = os.path.join(ROOT, "tests", "golden")
HOW = ("tools/sweep_regret.py on MI355X, round 4 (tools/r4_cold.sh): every product kernel that takes the shape, per-dispatch events, median of 16 launches, kernel order "
       "shuffled per shape, COLD weights (>= 320 MiB of weight buffers in rotation, every launch on the next one); ")


def dump(name, source, pats):
    data = {}
    for pat in pats:
        d = load_raw(os.path.join(R, pat))
        assert d, pat
        data.update(d)
    rows = [[M, K, N, out, {k: round(v, 2) for k, v in sorted(d.items())}] for (M, K, N, out), d in sorted(data.items())]
    with open(os.path.join(G, name), "w") as f:
        f.write('{"source": %s, "unit": "us", "columns": ["M", "K", "N", "out_dtype", "times by kernel"], "shapes": [\n' % json.dumps(HOW + source))
        f.write(",\n".join(json.dumps(r, separators=(",", ":")) for r in rows))
        f.write("\n]}\n")
    print(name, len(rows))


dump("dispatch_times_cold_fit.json", "seeds 101-110, 133-142 (LLM / diffusion dimensions), 113-114, 143-145 (DIMS=ext: small and ragged dimensions), 117-118, 146-148 (MS=200..1024 with "
     "the small tiles offered), 121-122, 149-150 (fp32 output): what the dispatch's cost model is FITTED on", ["std*.txt", "ext1*.txt", "midm*.txt", "f32_*.txt", "ext_f32_*.txt"])
dump("dispatch_times_cold_anchors.json", "the BASELINE configurations, bench.py's workloads and their neighbours, three repeats: fitted on with 6x weight", ["anchors_*.txt"])
dump("dispatch_times_cold_heldout.json", "seeds 127-130 (standard), 131 (DIMS=ext), 132 (MS=200..1024): NEVER fitted on", ["heldout*.txt"])
dump("dispatch_times_cold_nows.json", "NOWS=1 (no split-K workspace: a sharded linear's calls, workspace-less callers), seeds 123-126 (standard, DIMS=ext, MS=200..1024): NEVER fitted on",
     ["nows*.txt"])
