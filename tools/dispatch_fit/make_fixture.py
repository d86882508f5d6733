"""tests/golden/dispatch_times_r03.json: the measured per-kernel times (us, median over the repeats of one shape) of the raw regret sweeps of round 3
(tools/sweep_regret.py on MI355X, seeds 11-22) - the data the dispatch's cost model was fitted on and is regression-tested against on the CPU."""
import glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "dispatch_fit"))
from data import load_raw
data = load_raw(os.path.join(ROOT, "gpurun_out/cstore/regret*.txt"))
rows = [[M, K, N, out, {k: round(v, 2) for k, v in sorted(d.items())}] for (M, K, N, out), d in sorted(data.items())]
doc = {"source": "tools/sweep_regret.py (AUTO against every product kernel that accepts the shape; per-dispatch events, median of 16 launches over rotating weight buffers), "
                 "MI355X, end of round 3, seeds 11-22; median over the repeats of a shape", "unit": "us", "columns": ["M", "K", "N", "out_dtype", "times by kernel"], "shapes": rows}
with open(os.path.join(ROOT, "tests", "golden", "dispatch_times_r03.json"), "w") as f:
    f.write('{"source": %s, "unit": "us", "columns": %s, "shapes": [\n' % (json.dumps(doc["source"]), json.dumps(doc["columns"])))
    f.write(",\n".join(json.dumps(r, separators=(",", ":")) for r in rows))
    f.write("\n]}\n")
print(len(rows), "shapes")
