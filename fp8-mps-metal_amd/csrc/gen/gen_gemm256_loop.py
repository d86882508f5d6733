#!/usr/bin/env python3
"""Generates csrc/fp8mi_gemm256_loop.inc: the hand-scheduled K loop of the 256x256 one-wave-per-SIMD FP8 GEMM
(fp8mi_gemm256.hip) as two inline-asm statements (the pipelined loop and the NaN-scrubbing redo loop).

Why a generator: hipcc cannot express this loop.  It treats the MFMA builtin as a pure value computation and
re-clusters all 64 MFMAs of a K-step regardless of scheduling fences, and it cannot keep 256 accumulators in AGPRs
next to 128 fragment VGPRs without shuffling them.  The loop is therefore written with fixed
physical registers:

    a[0:255]     accumulators, a[(tm * 8 + tn) * 4 + j]   (W fragment tn = MFMA operand A, X fragment tm = operand B)
    v[120:127]   LDS addresses of the accumulator dump;  v119 landing register of the L2 prefetch loads (never read);
    v118         running source offset of the stage DMA (row part: range-checked by the buffer descriptor, see dma_block)

The accumulators leave through the LDS: behind the last MFMA and a barrier (the ring is idle then) the loop writes fragment
rows tm = 0..3 of every wave into the wave's 32-KiB quarter of the ring as plain fp32 rows (16 rows x 512 B per fragment
row, 16-byte chunks XOR-swizzled by the row so that the writes and the row-wise read-back are bank-conflict free); rows
4..7 stay in a[128:255] and are declared to the compiler as four 32-float OUTPUT operands pinned there, which a second
small asm statement (FP8MI_GEMM256_DUMP_HI) takes as inputs once the HIP epilogue has consumed the first half.  The HIP
side never touches an AGPR itself: hidden liveness in AGPRs does not survive hipcc (it parks its own values there), and
reading 1024-bit AGPR operands element-wise made it spill.
    v[128:191]   W fragments (rows of B, n), 8 registers per fragment tn
    v[192:255]   X fragments (rows of A, m), 8 registers per fragment tm

Schedule of one K-step (one wave; 64 MFMAs = 2048 matrix-pipe cycles), tn-major:

    row tn = 0:  MFMA(0, tm), each behind a counted lgkmcnt wait for X fragment tm (read at the end of the previous step)
                 s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier   -> stage t+1 has landed for everyone, slot of stage t is free
    rows 1..6:   MFMA(tn, tm); after row tn the W fragment tn is dead: its registers are re-read from stage t+1
                 the 16 LDS-DMA instructions of stage t+2 are issued one every DMA_EVERY MFMAs from DMA_FIRST on
    row 7:       MFMA(7, tm); X fragment tm is dead after it: re-read from stage t+1 (XDELAY MFMAs later)

so fragment reads, the global->LDS stream and the MFMAs of one wave overlap without a second wave on the SIMD, with
one barrier per K-step.  Usage: python gen_gemm256_loop.py > ../fp8mi_gemm256_loop.inc
"""
import sys

# the product schedule; VARIANTS (diagnostic build only) override single entries for A/B timing
PRODUCT = dict(
    dma_first=8,      # index of the MFMA after which the first DMA of a step is issued (>= barrier_after + 1)
    dma_every=3,      # MFMAs between DMA instructions
    xdelay=1,         # X fragment tm is re-read this many MFMAs after MFMA(7, tm)
    barrier_after=7,  # the step's barrier sits behind this MFMA index
    schedule=1,       # 1: DMA under MFMAs dma_first.. of the step, one barrier; 2: DMA as an even stream, two barriers (see step2)
    pf=2,             # > 0: each step one L2 prefetch instruction per wave for the stage `pf` steps ahead of the DMA's (see pf_group)
    pf_at=56,         # index of the MFMA after which it is issued (behind the step's last DMA)
    pf_stagger=0,     # > 0: wave w issues it pf_stagger * (3 - w) MFMAs earlier (needs the kernel's per-wave split of the lines)
    wave_delay=0,     # > 0: behind the step's barrier wave w idles w * wave_delay cycles (see wave_delay_group)
    # timing-only ablations (wrong results): which parts of the steady-state step are left out
    no_vmwait=False, no_dma=False, no_barrier=False, no_reads=False, no_mfma=False,
)
VARIANTS = {
    1: dict(no_vmwait=True),
    2: dict(no_dma=True),
    3: dict(no_barrier=True),
    4: dict(no_reads=True),
    5: dict(no_mfma=True),
    6: dict(dma_every=2),
    7: dict(dma_every=1),
    8: dict(dma_every=4),
    9: dict(xdelay=0),
    10: dict(no_dma=True, no_reads=True, no_barrier=True),
    11: dict(pf=0),
    12: dict(pf=3),
    13: dict(pf=1),
    14: dict(barrier_after=3, dma_first=4),
    15: dict(pf=4),
    16: dict(pf_at=20),
    17: dict(schedule=2),
    18: dict(schedule=2, xdelay=0),
    19: dict(pf_at=56),              # + the kernel's per-wave split of the prefetch lines (kSplitPf)
    20: dict(pf_stagger=12),         # ... issued at MFMA 20 / 32 / 44 / 56 by wave 0 / 1 / 2 / 3
    21: dict(pf_stagger=4),
    # round 3: de-phase the four waves behind the step's barrier (wave w idles w x wave_delay cycles), so that their stage DMA
    # instructions reach the CU's one address path one after the other instead of together
    22: dict(wave_delay=16),
    23: dict(wave_delay=24),
    24: dict(wave_delay=32),
    25: dict(wave_delay=48),
    26: dict(wave_delay=32, dma_every=2),
    27: dict(wave_delay=24, dma_every=4),
}
# variants of the 256x128 form (diagnostic build; kernel ids 190 + n): its steps last half as long, so the same prefetch distance in stages is half the time
NVARIANTS = {
    1: dict(pf=3), 2: dict(pf=4), 3: dict(pf=6), 4: dict(pf=1), 5: dict(pf=0), 6: dict(dma_every_n=1), 7: dict(dma_every_n=1, pf=4), 8: dict(pf=8),
}
P = dict(PRODUCT)

TN = 8   # W fragments per wave: 8 = 256x256 workgroup tile (wave 128x128), 4 = 256x128 (wave 128x64); set by the emitters below
WF = lambda tn: 128 + 8 * tn
XF = lambda tm: 192 + 8 * tm
ACC = lambda tn, tm: (tm * TN + tn) * 4   # fragment row tm = one block of 4 TN AGPRs


def mfma(tn, tm):
    a = ACC(tn, tm)
    return (f"v_mfma_scale_f32_16x16x128_f8f6f4 a[{a}:{a+3}], v[{WF(tn)}:{WF(tn)+7}], v[{XF(tm)}:{XF(tm)+7}], "
            f"a[{a}:{a+3}], %[vscale], %[vscale] op_sel_hi:[0,0,0]")


def rd(base_reg, t, lo_addr, hi_addr):
    off = t * 2048  # 16 rows x 128 B
    return [f"ds_read_b128 v[{base_reg}:{base_reg+3}], {lo_addr} offset:{off}",
            f"ds_read_b128 v[{base_reg+4}:{base_reg+7}], {hi_addr} offset:{off}"]


def rdW(tn, which):  # which: 'c' current slot, 'n' next slot
    return rd(WF(tn), tn, f"%[blo_{which}]", f"%[bhi_{which}]")


def rdX(tm, which):
    return rd(XF(tm), tm, f"%[alo_{which}]", f"%[ahi_{which}]")


def dma_block(slot_m0, kreg, first=True, tail=False):
    """16 DMA instructions of one stage as a list of 16 instruction groups.  The row part of the source offset travels in a
    VGPR (v118 = va0 / vb0 + jo x row-group stride) and only k in the scalar offset: the hardware range-checks the VGPR offset
    against the descriptor's num_records, so the rows of a ragged last m-tile beyond M read as zeros without a mask.
    tail: the stage is the LAST one of a K that is not a multiple of 128 - the lanes whose 16-byte chunk lies beyond K start from an out-of-range
    offset (va0t / vb0t: kOOB for those lanes, va0 / vb0 otherwise), so the chunk reads as zeros (+0.0 in e4m3) instead of the next row's bytes."""
    groups = []
    for j in range(8 + TN):   # 8 row groups of the A panel, TN of the B panel per wave
        g = []
        if j == 0:
            g += [f"s_mov_b32 m0, {slot_m0}", f"s_mov_b32 %[t0], {kreg}", "v_mov_b32 v118, " + ("%[va0t]" if tail else "%[va0]")]
        elif j == 8:
            g += ["s_add_u32 m0, m0, 0x1000", "v_mov_b32 v118, " + ("%[vb0t]" if tail else "%[vb0]")]
        else:
            g += ["s_add_u32 m0, m0, 0x1000"]
        g += ["s_nop 0"]
        if j < 8:
            g += ["buffer_load_dwordx4 v118, %[ra], %[t0] offen lds", "v_add_u32 v118, %[sa], v118"]
        else:
            g += ["buffer_load_dwordx4 v118, %[rb], %[t0] offen lds", "v_add_u32 v118, %[sb], v118"]
        groups.append(g)
    return groups


def pf_group():
    """L2 prefetch: one dword per 128-byte line of this wave's share of a later stage (the tiles that run on one XCD at one
    time share A and B panels; each warms 1/8 of its A and 1/4 of its B panel - fp8mi_gemm256.hip sets up the offsets).
    The data is discarded; the load is younger than the step's DMAs, so the counted vmcnt wait lets it stay in flight."""
    return [f"s_add_u32 %[t1], %[k2], {hex(128 * P['pf'])}", "s_min_u32 %[t1], %[t1], %[klast]",
            "buffer_load_dword v119, %[pfoff], %[rpf], %[t1] offen"]


def wave_delay_group():
    """wave w (%[wave]) idles about w * P['wave_delay'] cycles: a counted loop of s_nop (the barrier releases all four waves in the
    same cycle; left alone they reach every DMA instruction of the step together and queue in the CU's address path)"""
    n = max(P["wave_delay"] - 12, 4)   # the loop's own SALU + branch cost ~12 cycles per trip
    nops = []
    while n > 0:
        k = min(n, 8)
        nops.append(f"s_nop {k - 1}")
        n -= k
    return ["s_mov_b32 %[t1], %[wave]", "4:", "s_cmp_eq_u32 %[t1], 0", "s_cbranch_scc1 5f"] + nops + ["s_sub_u32 %[t1], %[t1], 1", "s_branch 4b", "5:"]


def swap_slots():
    out = []
    for r in ("alo", "ahi", "blo", "bhi"):
        out.append(f"v_swap_b32 %[{r}_c], %[{r}_n]")
    out += ["s_mov_b32 %[t1], %[m0_c]", "s_mov_b32 %[m0_c], %[m0_n]", "s_mov_b32 %[m0_n], %[t1]"]
    return out


def canonical_prologue_reads():
    out = []
    for tn in range(TN - 1):
        out += rdW(tn, "c")
    for tm in range(8):
        out += rdX(tm, "c")
    out += rdW(TN - 1, "c")
    return out


def step(dma, reads, tail=False):
    NM, LR = 8 * TN, 8 * (TN - 1)   # MFMAs per step, index of the first MFMA of the last row
    pre = [[] for _ in range(NM)]
    post = [[] for _ in range(NM)]
    for tm in range(8):
        pre[tm].append(f"s_waitcnt lgkmcnt({min(15, 2 * (7 - tm) + 2)})")
    if reads:
        BARRIER_AFTER = P["barrier_after"]
        vm = 1 if (P["pf"] and dma) else 0   # the previous step's prefetch may stay in flight (tail steps: drain everything)
        post[BARRIER_AFTER] += ["s_waitcnt lgkmcnt(0)" if P["no_vmwait"] else f"s_waitcnt vmcnt({vm}) lgkmcnt(0)"]
        if not P["no_barrier"]:
            post[BARRIER_AFTER] += ["s_barrier"]
        if P["wave_delay"] and dma:
            post[BARRIER_AFTER] += wave_delay_group()
        for tn in range(TN - 1):  # W fragment tn is dead behind its row; its re-read also has to sit behind the barrier
            post[max(8 * tn + 7, BARRIER_AFTER)] += rdW(tn, "n")
        for tm in range(8):
            post[min(NM - 1, LR + tm + P["xdelay"])] += rdX(tm, "n")
        post[NM - 1] += rdW(TN - 1, "n")
        pre[LR].append(f"s_waitcnt lgkmcnt({2 * (TN - 1)})")
    else:
        pre[LR].append("s_waitcnt lgkmcnt(0)")
    every = P["dma_every"] if TN == 8 else P.get("dma_every_n", 2)   # 12 instructions under 24 MFMAs on the 256x128 tile
    pf_at = P["pf_at"] if TN == 8 else NM - 2
    if dma and not P["no_dma"]:
        for j, g in enumerate(dma_block("%[m0_c]", "%[k2]", tail=tail)):
            post[min(NM - 1, P["dma_first"] - 1 + j * every)] += g
        if P["pf"] and not P["pf_stagger"]:
            post[pf_at] += pf_group()
        elif P["pf"]:
            for w in range(4):   # one copy per wave, each behind a scalar test of the wave's index
                g = [f"s_cmp_lg_u32 %[wave], {w}", f"s_cbranch_scc1 3{w}f"] + pf_group() + [f"3{w}:"]
                post[pf_at - P["pf_stagger"] * (3 - w)] += g
    out = []
    for i in range(NM):
        out += pre[i]
        if not P["no_mfma"]:
            out.append(mfma(i // 8, i % 8))
        out += [l for l in post[i] if not (P["no_reads"] and l.startswith("ds_read"))]
    if reads:
        out += swap_slots()
    if dma:
        out.append("s_add_u32 %[k2], %[k2], 0x80")
    return out


def dump(rows):
    """fragment rows `rows` (4 of them) of this wave's accumulators -> its share of the ring, fp32 rows of 16 TN columns, chunks swizzled"""
    L = []
    for tn in range(TN):
        L += [f"v_xor_b32 v{120 + tn}, {tn * 64}, %[dkey]", f"v_add_u32 v{120 + tn}, %[drow], v{120 + tn}"]
    for tm in rows:
        for tn in range(TN):
            a = ACC(tn, tm)
            L.append(f"ds_write_b128 v{120 + tn}, a[{a}:{a + 3}] offset:{(tm % 4) * 1024 * TN}")
    L.append("s_waitcnt lgkmcnt(0)")
    return L


# ---- schedule 2: the stage DMA as ONE even stream over the whole step, two barriers -------------------------------------
# The address path of a CU takes ~28 cycles per 1-KiB DMA instruction (profiles/r02_pmc_gemm_flux.txt): 64 of them per K-step
# need ~1800 cycles, more than the 46 MFMAs (1472 cycles) schedule 1 issues them under - it stalls ~320 cycles per step
# in the issue (ablation: profiles/r02_gemm256_variants.txt).  Here a stage is issued B panel first, then A panel, one
# instruction every 4 MFMAs: 8, 12, .. 60 of this step and 0, 4 of the next (m0 and the offset register simply carry over).
# That works because the two panels have different deadlines: the W fragments (B) of stage t+1 are re-read from row 0 on,
# the X fragments (A) only in row 7.  Barrier 1 (behind MFMA 7): every wave is done reading stage t -> its slot may be
# refilled; B of stage t+1 has landed (vmcnt 9: the A instructions and the prefetch issued since may stay in flight).
# Barrier 2 (behind MFMA 55): A of stage t+1 has landed (vmcnt 12: the 8 B and 4 A instructions of stage t+2 issued since).
def dma_group2(j, slot_m0, kreg):
    """instruction j = 0..15 of a stage's stream: 0..7 B panel (LDS groups 32 + w + 4 jo), 8..15 A panel"""
    g = []
    if j == 0:
        g += [f"s_add_u32 m0, {slot_m0}, 0x8000", f"s_mov_b32 %[t0], {kreg}", "v_mov_b32 v118, %[vb0]"]
    elif j == 8:
        g += [f"s_sub_u32 m0, m0, {hex(0x8000 + 7 * 0x1000)}", "v_mov_b32 v118, %[va0]"]
    else:
        g += ["s_add_u32 m0, m0, 0x1000"]
    g += ["s_nop 0"]
    if j < 8:
        g += ["buffer_load_dwordx4 v118, %[rb], %[t0] offen lds", "v_add_u32 v118, %[sb], v118"]
    else:
        g += ["buffer_load_dwordx4 v118, %[ra], %[t0] offen lds", "v_add_u32 v118, %[sa], v118"]
    return g


def step2(dma, reads, pf=True):
    pre = [[] for _ in range(64)]
    post = [[] for _ in range(64)]
    for tm in range(8):
        pre[tm].append(f"s_waitcnt lgkmcnt({min(15, 2 * (7 - tm) + 2)})")
    if reads:
        # the stream of stage t+1 ends here (its A-panel instructions 14, 15); m0 / t0 carry over from the previous step
        post[0] += dma_group2(14, None, None)
        post[4] += dma_group2(15, None, None)
        post[7] += ["s_waitcnt vmcnt(9) lgkmcnt(0)", "s_barrier"]
        for tn in range(7):
            post[8 * tn + 7] += rdW(tn, "n")
        post[55] += [f"s_waitcnt vmcnt({12 if dma else 0})", "s_barrier"]
        for tm in range(8):
            post[min(63, 56 + tm + P["xdelay"])] += rdX(tm, "n")
        post[63] += rdW(7, "n")
        pre[56].append("s_waitcnt lgkmcnt(14)")
    else:
        pre[56].append("s_waitcnt lgkmcnt(0)")
    if dma:
        for j in range(14):
            post[8 + 4 * j] += dma_group2(j, "%[m0_c]", "%[k2]")
        if P["pf"] and pf:
            post[62] += pf_group()
    out = []
    for i in range(64):
        out += pre[i]
        out.append(mfma(i // 8, i % 8))
        out += post[i]
    if reads:
        out += swap_slots()
    if dma:
        out.append("s_add_u32 %[k2], %[k2], 0x80")
    return out


def pipelined2():
    L = []
    for j in range(16):   # stage 0, whole
        L += dma_group2(j, "%[m0_c]", "%[k2]")
    L.append("s_add_u32 %[k2], %[k2], 0x80")
    for j in range(14):   # stage 1 up to the two instructions the first step issues
        L += dma_group2(j, "%[m0_n]", "%[k2]")
    L.append("s_add_u32 %[k2], %[k2], 0x80")
    if P["pf"]:
        L += pf_group()
    L += zero_acc()
    L += [f"s_waitcnt vmcnt({14 + (1 if P['pf'] else 0)})", "s_barrier"]
    L += canonical_prologue_reads()
    L += ["s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 2f", "1:"]
    L += step2(True, True)
    L += ["s_sub_u32 %[nloop], %[nloop], 1", "s_cmp_lg_u32 %[nloop], 0", "s_cbranch_scc1 1b", "2:"]
    L += step2(False, True)
    L += step2(False, False)
    L += ["s_nop 7", "s_nop 7", "s_nop 7", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    L += dump(range(4))
    return L


# ---- fused tail (round 3): the epilogue of the common case runs UNDER the last K-step's MFMAs, straight from the AGPRs -----------
# The LDS-dump epilogue costs 11.4 k cycles per 256x256 tile (in-kernel stamps: dump, read back, scale, convert, store with one wave per
# SIMD and nothing to hide the latencies) behind a K loop of 64 k.  For per-tensor scales without bias / scale_result on tiles that are
# whole in N the same arithmetic - (acc * s1) * s2, one packed convert, the ring kernels' rounding sequence - is done in registers:
#   * a fragment pair (tn, tn+1; tm) is final once both its MFMAs of the last step have run; `v_accvgpr_read` copies its 8 accumulators
#     into the registers of W fragments the step has finished with;
#   * 4 `v_permlane16_swap_b32` exchange lane-group rows between the two fragments so that every lane holds 8 CONSECUTIVE columns
#     (lane group fg: columns (tn + (fg & 1)) * 16 + (fg >> 1) * 8 ..+7 of row tm * 16 + fr) - one 16-byte store per lane (two for
#     fp32), 64 contiguous bytes per row and instruction;
#   * 8 packed multiplies, 4 packed converts, one `buffer_store_dwordx4 ... nt` whose descriptor range-checks the rows of a ragged
#     last m-tile; the row block of tm travels in the scalar offset;
#   * the ops of pair p are issued from MFMA 16 p + 16 on, FUSED_Q per MFMA (an 8-pass MFMA leaves 7 issue slots), the rest behind the
#     last MFMA: the tail is bound by ~27 VALU ops per (pair, tm) instead of by LDS round trips.
# NaN bytes (reference: decode to 0): the sum of the fragments (pair p, tm in p's own row blocks) covers every row block and every
# column block of the wave tile; a NaN byte poisons a whole row or column of accumulators, so a NaN in that sum proves one took part.
FUSED_Q = 7


def fused_item(p, tm, out):
    tn = 2 * p
    S = 128 + 8 * ((p * 8 + tm) % 2)   # two alternating sets of 8 temporaries in the registers of W fragments 0 and 1 (dead by then)
    ops = []
    if tm == 0:
        ops.append("s_mov_b32 %[t0], 0")
    ops += [f"v_accvgpr_read_b32 v{S + j}, a{ACC(tn, tm) + j}" for j in range(4)]
    ops += [f"v_accvgpr_read_b32 v{S + 4 + j}, a{ACC(tn + 1, tm) + j}" for j in range(4)]
    ops += [f"v_permlane16_swap_b32 v{S + j}, v{S + 4 + j}" for j in range(4)]
    sel = (tm in (2 * p, 2 * p + 1)) if TN == 8 else (tm // 4 == p)
    if sel:
        ops += [f"v_pk_add_f32 v[118:119], v[118:119], v[{S + 2 * j}:{S + 2 * j + 1}]" for j in range(4)]
    ops += [f"v_pk_mul_f32 v[{S + 2 * j}:{S + 2 * j + 1}], v[{S + 2 * j}:{S + 2 * j + 1}], v[120:121]" for j in range(4)]
    ops += [f"v_pk_mul_f32 v[{S + 2 * j}:{S + 2 * j + 1}], v[{S + 2 * j}:{S + 2 * j + 1}], v[122:123]" for j in range(4)]
    if out == "f32":
        ops += [f"buffer_store_dwordx4 v[{S}:{S + 3}], %[voff], %[rc], %[t0] offen offset:{p * 128} nt",
                f"buffer_store_dwordx4 v[{S + 4}:{S + 7}], %[voff], %[rc], %[t0] offen offset:{p * 128 + 16} nt"]
    else:
        cvt = "v_cvt_pk_bf16_f32" if out == "bf16" else "v_cvt_pk_f16_f32"
        ops += [f"{cvt} v{S + j}, v{S + 2 * j}, v{S + 2 * j + 1}" for j in range(4)]
        ops += [f"buffer_store_dwordx4 v[{S}:{S + 3}], %[voff], %[rc], %[t0] offen offset:{p * 64} nt"]
    ops.append("s_add_u32 %[t0], %[t0], %[srow]")
    return ops


def fused_last_step(out):
    NM, LR = 8 * TN, 8 * (TN - 1)
    pre = [[] for _ in range(NM)]
    for tm in range(8):
        pre[tm].append(f"s_waitcnt lgkmcnt({min(15, 2 * (7 - tm) + 2)})")
    pre[LR].append("s_waitcnt lgkmcnt(0)")
    queue = []   # (index of the MFMA the op may follow, op)
    for p in range(TN // 2):
        for tm in range(8):
            queue += [(16 * p + 16, op) for op in fused_item(p, tm, out)]
    L = ["v_mov_b32 v118, 0", "v_mov_b32 v119, 0",
         "v_mov_b32 v120, %[s1]", "v_mov_b32 v121, %[s1]", "v_mov_b32 v122, %[s2]", "v_mov_b32 v123, %[s2]"]
    qi = 0
    for i in range(NM):
        L += pre[i]
        L.append(mfma(i // 8, i % 8))
        n = 0
        while qi < len(queue) and queue[qi][0] <= i and n < FUSED_Q:
            L.append(queue[qi][1]); qi += 1; n += 1
    L += ["s_nop 7", "s_nop 7"]
    L += [op for _, op in queue[qi:]]
    L += ["v_add_f32 %[nanv], v118, v119"]
    return L


def zero_acc():
    return [f"v_accvgpr_write_b32 a{i}, 0" for i in range(32 * TN)]


def pipelined(fused=None):
    L = []
    if fused:   # the two per-tensor scales: scalar loads that land under the first K-steps (their first use is in the tail)
        L += ["s_load_dword %[s1], %[pm1], 0x0", "s_load_dword %[s2], %[pm2], 0x0"]
    # prologue: stages 0 and 1 in flight, accumulators cleared under their latency
    for g in dma_block("%[m0_c]", "%[k2]"):
        L += g
    L.append("s_add_u32 %[k2], %[k2], 0x80")
    for g in dma_block("%[m0_n]", "%[k2]"):
        L += g
    L.append("s_add_u32 %[k2], %[k2], 0x80")
    if P["pf"]:
        L += pf_group()
    L += zero_acc()
    L += [f"s_waitcnt vmcnt({8 + TN + (1 if P['pf'] else 0)})", "s_barrier"]
    L += canonical_prologue_reads()
    # nloop = nk - 2 steps stage two steps ahead; the LAST of them (it stages the last K-step: the K tail, if any) is peeled and takes the
    # tail operands (equal to the plain ones when K is a multiple of 128).  (4 SALU instructions ahead of the loop: 16 bytes, the
    # hand-written stream keeps its 8-byte phase)
    L += ["s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 3f", "s_sub_u32 %[nloop], %[nloop], 1", "s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 2f", "s_nop 0", "1:"]
    L += step(True, True)
    L += ["s_sub_u32 %[nloop], %[nloop], 1", "s_cmp_lg_u32 %[nloop], 0", "s_cbranch_scc1 1b", "2:"]
    L += step(True, True, tail=True)
    L += ["3:"]
    L += step(False, True)
    if fused:
        # one statement, three tails (the output type is a launch-time switch; as three statements hipcc kept a private copy of
        # every read-write operand per statement and spilled them): FP8MI_F32 = 0, FP8MI_F16 = 1, FP8MI_BF16 = 2
        L += ["s_cmp_eq_u32 %[od], 2", "s_cbranch_scc1 10f", "s_cmp_eq_u32 %[od], 1", "s_cbranch_scc1 11f"]
        L += fused_last_step("f32") + ["s_branch 12f", "10:"]
        L += fused_last_step("bf16") + ["s_branch 12f", "11:"]
        L += fused_last_step("f16") + ["12:"]
        return L
    L += step(False, False)
    L += ["s_nop 7", "s_nop 7", "s_nop 7", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    L += dump(range(4))
    return L


def scrub_reg(r):
    return [f"v_or_b32 %[vt0], 0x80808080, v{r}", "v_not_b32 %[vt0], %[vt0]",
            "v_add_u32 %[vt0], 0x7f7f7f7f, %[vt0]", "v_or_b32 %[vt0], 0x7f7f7f7f, %[vt0]", "v_not_b32 %[vt0], %[vt0]",
            "v_lshrrev_b32 %[vt1], 7, %[vt0]", "v_sub_u32 %[vt1], %[vt0], %[vt1]", "v_or_b32 %[vt0], %[vt0], %[vt1]",
            "v_not_b32 %[vt0], %[vt0]", f"v_and_b32 v{r}, v{r}, %[vt0]"]


def scrub_step(dma, tail=False):
    L = ["s_waitcnt vmcnt(0)", "s_barrier"]
    L += canonical_prologue_reads()
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]  # everyone has read the slot: it may be refilled
    if dma:
        for g in dma_block("%[m0_c]", "%[k2]", tail=tail):
            L += g
        L.append("s_add_u32 %[k2], %[k2], 0x80")
    for r in list(range(128, 128 + 8 * TN)) + list(range(192, 256)):
        L += scrub_reg(r)
    for i in range(8 * TN):
        L.append(mfma(i // 8, i % 8))
    L += swap_slots()
    return L


def scrubbed():
    L = []
    for g in dma_block("%[m0_c]", "%[k2]"):
        L += g
    L.append("s_add_u32 %[k2], %[k2], 0x80")
    for g in dma_block("%[m0_n]", "%[k2]"):
        L += g
    L.append("s_add_u32 %[k2], %[k2], 0x80")
    L += zero_acc()
    L += ["s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 3f", "s_sub_u32 %[nloop], %[nloop], 1", "s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 2f", "1:"]
    L += scrub_step(True)
    L += ["s_sub_u32 %[nloop], %[nloop], 1", "s_cmp_lg_u32 %[nloop], 0", "s_cbranch_scc1 1b", "2:"]
    L += scrub_step(True, tail=True)
    L += ["3:"]
    L += scrub_step(False)
    L += scrub_step(False)
    L += ["s_nop 7", "s_nop 7", "s_nop 7", "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
    L += dump(range(4))
    return L


def emit(name, lines, scrub, fused=False):
    print(f"#define {name}() \\")
    print("    asm volatile( \\")
    for l in lines:
        print(f'        "{l}\\n\\t" \\')
    nblk = TN // 2   # fragment rows 4..7 (16 TN AGPRs) as 32-float operands: acc4.. pinned to a[16 TN + 32 i : + 31]
    outs = [] if fused else [f'"={{a[{16 * TN + 32 * i}:{16 * TN + 32 * i + 31}]}}"(acc{4 + i})' for i in range(nblk)]
    if fused:
        outs += ['[nanv] "=&v"(nanv)', '[s1] "=&s"(fs1)', '[s2] "=&s"(fs2)']
    outs += ['[alo_c] "+v"(alo_c)', '[ahi_c] "+v"(ahi_c)', '[blo_c] "+v"(blo_c)', '[bhi_c] "+v"(bhi_c)',
            '[alo_n] "+v"(alo_n)', '[ahi_n] "+v"(ahi_n)', '[blo_n] "+v"(blo_n)', '[bhi_n] "+v"(bhi_n)',
            '[k2] "+s"(k2)', '[m0_c] "+s"(m0_c)', '[m0_n] "+s"(m0_n)', '[nloop] "+s"(nloop)', '[t0] "=&s"(t0)', '[t1] "=&s"(t1)']
    if scrub:
        outs += ['[vt0] "=&v"(vt0)', '[vt1] "=&v"(vt1)']
    ins = ['[va0] "v"(va0)', '[vb0] "v"(vb0)', '[va0t] "v"(va0t)', '[vb0t] "v"(vb0t)', '[vscale] "v"(vscale)', '[ra] "s"(ra)', '[rb] "s"(rb)', '[sa] "s"(sa)', '[sb] "s"(sb)',
           '[drow] "v"(drow)', '[dkey] "v"(dkey)', '[pfoff] "v"(pfoff)', '[rpf] "s"(rpf)', '[klast] "s"(klast)', '[wave] "s"(wave_s)']
    # m0 is written by the DMA groups (s_mov_b32 / s_add_u32 m0): declared, so that LLVM never keeps an M0 initialisation of its own live across the statement
    if fused:
        ins += ['[rc] "s"(frc)', '[voff] "v"(fvoff)', '[srow] "s"(fsrow)', '[pm1] "s"(fpm1)', '[pm2] "s"(fpm2)', '[od] "s"(fod)']
    clob = [f'"v{i}"' for i in range(118, 256)] + [f'"a{i}"' for i in range(32 * TN if fused else 16 * TN)] + ['"m0"', '"scc"', '"memory"']
    print("        : " + ", ".join(outs) + " \\")
    print("        : " + ", ".join(ins) + " \\")
    print("        : " + ", ".join(clob) + ")")
    print()


def emit_dump_hi(name):
    print(f"#define {name}() \\")
    print("    asm volatile( \\")
    for l in dump(range(4, 8)):
        print(f'        "{l}\\n\\t" \\')
    ins = [f'"{{a[{16 * TN + 32 * i}:{16 * TN + 32 * i + 31}]}}"(acc{4 + i})' for i in range(TN // 2)] + ['[drow] "v"(drow)', '[dkey] "v"(dkey)']
    print("        : \\")
    print("        : " + ", ".join(ins) + " \\")
    print("        : " + ", ".join(f'"v{i}"' for i in range(120, 120 + TN)) + ', "memory")')
    print()


if __name__ == "__main__":
    print("// GENERATED by csrc/gen/gen_gemm256_loop.py - do not edit; see that file for the schedule.")
    print(f"// product schedule: {PRODUCT}")
    emit("FP8MI_GEMM256_LOOP", pipelined2() if P["schedule"] == 2 else pipelined(), False)
    emit("FP8MI_GEMM256_LOOP_SCRUB", scrubbed(), True)
    emit_dump_hi("FP8MI_GEMM256_DUMP_HI")
    TN = 4   # the 256x128 workgroup tile (wave 128x64): same schedule, 32 MFMAs per step
    P.clear(); P.update(PRODUCT)
    emit("FP8MI_GEMM256_LOOP_N128", pipelined(), False)
    emit("FP8MI_GEMM256_LOOP_SCRUB_N128", scrubbed(), True)
    emit_dump_hi("FP8MI_GEMM256_DUMP_HI_N128")
    TN = 8
    print("#ifdef FP8MI_DIAG  // schedule variants and timing-only ablations (libfp8mi_diag.so, kernel ids 80 + variant)")
    print("// fused tail (epilogue from the AGPRs under the last K-step; measured SLOWER than the LDS dump: its 64-byte row segments store at")
    print("// ~290 cycles per instruction and wave against ~200 for whole 256-byte rows, profiles/r03_tail_probe.txt) - FP8MI_DEBUG bit 2 selects it")
    emit("FP8MI_GEMM256_LOOP_FUSED", pipelined(True), False, fused=True)
    TN = 4
    P.clear(); P.update(PRODUCT)
    emit("FP8MI_GEMM256_LOOP_FUSED_N128", pipelined(True), False, fused=True)
    TN = 8
    for v, over in sorted(VARIANTS.items()):
        P.clear(); P.update(PRODUCT); P.update(over)
        print(f"// variant {v}: {over}")
        emit(f"FP8MI_GEMM256_LOOP_V{v}", pipelined2() if P["schedule"] == 2 else pipelined(), False)
    print("#define FP8MI_GEMM256_VARIANTS " + " ".join(f"X({v})" for v in sorted(VARIANTS)))
    TN = 4
    for v, over in sorted(NVARIANTS.items()):
        P.clear(); P.update(PRODUCT); P.update(over)
        print(f"// 256x128 variant {v}: {over}")
        emit(f"FP8MI_GEMM256_LOOP_N128_V{v}", pipelined(), False)
    print("#define FP8MI_GEMM256_NVARIANTS " + " ".join(f"X({v})" for v in sorted(NVARIANTS)))
    TN = 8
    print("#endif")
