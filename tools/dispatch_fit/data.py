"""Measured per-kernel times the dispatch's cost model is fitted on: either the raw output of tools/sweep_regret.py (gpurun_out/, scratch) or the
committed fixtures made from it by make_fixture.py (tests/golden/dispatch_times_cold_*.json: median over the repeats of a shape)."""
import glob, json, os, re, statistics
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, "tests", "golden")
# round 4, COLD weights (>= 320 MiB in rotation for every shape; tools/r4_cold.sh).  The sweeps of round 3 and of the first half of round 4 capped the rotation at
# 16 buffers: every weight matrix below 16 MiB was timed resident in the Infinity Cache - they are retired (git history: dispatch_times_r03.json, _r04*.json)
FIXTURE_FIT = os.path.join(G, "dispatch_times_cold_fit.json")          # fitted on
FIXTURE_ANCHORS = os.path.join(G, "dispatch_times_cold_anchors.json")  # BASELINE configs, bench workloads and neighbours: fitted on, 6x weight
FIXTURE_HELDOUT = os.path.join(G, "dispatch_times_cold_heldout.json")  # never fitted on
FIXTURE_NOWS = os.path.join(G, "dispatch_times_cold_nows.json")        # no workspace; never fitted on
FIXTURE = FIXTURE_FIT
LINE = re.compile(r"^M=\s*(\d+) K=\s*(\d+) N=\s*(\d+): auto\((\w+)\)\s+([\d.]+)\s+(.*?)\s+\| auto/best")


def load_raw(pattern):
    """{(M, K, N, out): {kernel: median us}} from sweep_regret.py output files ('f32' in the file name = fp32 output)"""
    data = defaultdict(lambda: defaultdict(list))
    for p in sorted(glob.glob(pattern)):
        out = "f32" if "f32" in os.path.basename(p) else "bf16"
        for ln in open(p, errors="replace"):
            m = LINE.match(ln)
            if m:
                toks = m.group(6).split()
                for name, t in zip(toks[0::2], toks[1::2]):
                    data[(int(m.group(1)), int(m.group(2)), int(m.group(3)), out)][name].append(float(t))
    return {k: {n: statistics.median(v) for n, v in d.items()} for k, d in data.items()}


def load_fixture(path=FIXTURE):
    return {(M, K, N, out): times for M, K, N, out, times in json.load(open(path))["shapes"]}


def load_all():
    """everything the model is fitted on (a shape measured in both files: the mean)"""
    out = {}
    for path in (FIXTURE_FIT, FIXTURE_ANCHORS):
        for k, v in load_fixture(path).items():
            if k in out:
                for n, t in v.items():
                    out[k][n] = 0.5 * (out[k][n] + t) if n in out[k] else t
            else:
                out[k] = dict(v)
    return out
