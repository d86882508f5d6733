#!/usr/bin/env python3
"""Check every counter named in tools/pmc_gemm.sh against `rocprofv3 -L` output (gpurun_out/avail.txt) and the per-block
slot limits of MI355X_MICROARCH.md.  Runs without a GPU.   python tools/check_pmc_names.py [avail.txt]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
avail = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "avail.txt")
names = set(re.findall(r"Counter_Name\s*:\s*(\S+)", open(avail).read()))
text = open(os.path.join(ROOT, "tools", "pmc_gemm.sh")).read()
passes = re.findall(r'^ "([A-Za-z0-9_ ]+)"$', text, flags=re.M)
LIMIT = {"SQ": 8, "TCC": 4, "TA": 2, "TCP": 2, "TD": 2, "GRBM": 2}
COST = {"FETCH_SIZE": ("TCC", 3), "WRITE_SIZE": ("TCC", 2)}
bad = 0
for i, p in enumerate(passes):
    use = {}
    for c in p.split():
        if c not in names:
            print(f"pass {i}: {c} is not listed for this GPU"); bad += 1
        blk, cost = COST.get(c, (c.split("_")[0], 1))
        use[blk] = use.get(blk, 0) + cost
    for blk, n in use.items():
        if n > LIMIT.get(blk, 2):
            print(f"pass {i}: {n} {blk} slots used, limit {LIMIT.get(blk, 2)}"); bad += 1
print(f"{len(passes)} passes checked, {bad} problem(s)")
sys.exit(1 if bad else 0)
