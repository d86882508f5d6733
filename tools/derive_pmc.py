#!/usr/bin/env python3
"""Sorted counter list + derived figures from a tools/pmc_gemm.sh summary.   python tools/derive_pmc.py <summary.txt> "<title>" <algorithmic MB>"""
import re, sys
vals = {}; kern = None
for l in open(sys.argv[1]):
    m = re.match(r"\s+([A-Za-z0-9_]+)\s+([0-9.]+)\s+\(n=", l)
    if m: vals[m.group(1)] = float(m.group(2))
    elif l.startswith("  ") and "kernel" in l and kern is None: kern = l.strip()
title, alg_mb = sys.argv[2], float(sys.argv[3])
print(f"# {title}\n# kernel: {kern}\n# tools/pmc_gemm.sh: 13 separate rocprofv3 --pmc passes (names checked against rocprofv3 -L; <= 8 SQ / 4 TCC / 2 TA,TCP,TD per pass); all completed.")
print("# values = mean per dispatch over 6 launches on rotating (cold) weights, summed over the chip.\n")
for k in sorted(vals): print(f"  {k:38s} {vals[k]:16.1f}")
g = vals.get
cu = 256.0
print("\n# derived")
l2 = g("TCC_REQ_sum", 0) * 128 / 1e6
print(f"  L2 -> CU bytes (TCC_REQ x 128 B)                  {l2:8.1f} MB   = {l2 / alg_mb:.1f} x the algorithmic {alg_mb:.1f} MB")
print(f"  L2 hit rate (HIT / (HIT + MISS))                   {g('TCC_HIT_sum', 0) / max(g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 0), 1):.3f}")
print(f"  fabric reads (FETCH_SIZE KiB x 2, gfx950)          {g('FETCH_SIZE', 0) * 2 * 1024 / 1e6:.1f} MB   writes (WRITE_SIZE) {g('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB")
print(f"  MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES)) {g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(4 * g('SQ_BUSY_CU_CYCLES', 1), 1):.3f}   of SIMD-cycles while a CU is busy")
wc = max(g("SQ_WAVE_CYCLES", 1), 1)
print(f"  wave-cycles: waiting (SQ_WAIT_ANY) {g('SQ_WAIT_ANY', 0) / wc:.2f}   issue-stalled (SQ_WAIT_INST_ANY) {g('SQ_WAIT_INST_ANY', 0) / wc:.2f}")
ta, nw = g("TA_TA_BUSY_sum", 0) / cu, g("TA_BUFFER_WAVEFRONTS_sum", 1) / cu
print(f"  address path: TA busy {ta:10.0f} cycles per CU for {nw:.0f} buffer wave-instructions = {ta / max(nw, 1):.1f} cycles each (16 = 64 B/clk); CU busy {g('SQ_BUSY_CU_CYCLES', 0) / cu:.0f} cycles")
lat = g("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(g("TCP_TCC_READ_REQ_sum", 1), 1)
print(f"  L1 (TCP): pending-stall {g('TCP_PENDING_STALL_CYCLES_sum', 0) / cu:10.0f} cycles per CU; mean L2 read latency {lat:.0f} cycles")
print(f"  data return (TD): busy {g('TD_TD_BUSY_sum', 0) / cu:.0f} cycles per CU, of which stalled on the L1 {g('TD_TC_STALL_sum', 0) / cu:.0f}")
print(f"  LDS: bank-conflict cycles / active cycles           {g('SQ_LDS_BANK_CONFLICT', 0) / max(g('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}   (LDS index active {g('SQ_LDS_IDX_ACTIVE', 0) / cu:.0f} cycles per CU)")
