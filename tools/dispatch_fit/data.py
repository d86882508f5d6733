"""Measured per-kernel times the dispatch's cost model is fitted on: either the raw output of tools/sweep_regret.py (gpurun_out/, scratch) or the
committed fixture made from it (tests/golden/dispatch_times_r03.json: median over the repeats of a shape)."""
import glob, json, os, re, statistics
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FIXTURE = os.path.join(ROOT, "tests", "golden", "dispatch_times_r03.json")
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


FIXTURE_R04 = FIXTURE.replace("r03", "r04")
FIXTURE_EXT = FIXTURE.replace("r03", "r04_ext")          # small / ragged dimensions (fitted on)
FIXTURE_HELDOUT = FIXTURE.replace("r03", "r04_heldout")  # never fitted on
FIXTURE_NOWS = FIXTURE.replace("r03", "r04_nows")        # no workspace; never fitted on
FIXTURE_EXT_NOWS = FIXTURE.replace("r03", "r04_ext_nows")
FIXTURE_MIDM = FIXTURE.replace("r03", "r04_midm")        # 200 <= M <= 1024 with the small tiles offered (fitted on)
FIXTURE_MIDM_NOWS = FIXTURE.replace("r03", "r04_midm_nows")
FIXTURE_ANCHORS = FIXTURE.replace("r03", "r04_anchors")  # the BASELINE configs, bench.py's workloads and their neighbours, three repeats (fitted on, weighted)


def load_fixture(path=FIXTURE):
    return {(M, K, N, out): times for M, K, N, out, times in json.load(open(path))["shapes"]}


def load_all():
    """everything the model is fitted on: round 3's sweeps + round 4's seeds 31-33 + the small / ragged dimensions + the mid-M sweeps (a shape measured twice: the mean)"""
    out = {}
    for path in (FIXTURE, FIXTURE_R04, FIXTURE_EXT, FIXTURE_MIDM, FIXTURE_ANCHORS):
        for k, v in load_fixture(path).items():
            if k in out:
                for n, t in v.items():
                    out[k][n] = 0.5 * (out[k][n] + t) if n in out[k] else t
            else:
                out[k] = dict(v)
    return out
