#!/usr/bin/env python3
"""Generates csrc/fp8mi_gemm256_loop.inc: the hand-scheduled K loop of the 256x256 one-wave-per-SIMD FP8 GEMM
(fp8mi_gemm256.hip) as two inline-asm statements (the pipelined loop and the NaN-scrubbing redo loop).

Why a generator: hipcc cannot express this loop.  It treats the MFMA builtin as a pure value computation and
re-clusters all 64 MFMAs of a K-step regardless of scheduling fences, and it cannot keep 256 accumulators in AGPRs
next to 128 fragment VGPRs without shuffling them.  The loop is therefore written with fixed
physical registers:

    a[0:255]     accumulators, a[(tn * 8 + tm) * 4 + j]   (W fragment tn = MFMA operand A, X fragment tm = operand B)
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

DMA_FIRST = 8      # index of the MFMA after which the first DMA of a step is issued (>= 8: behind the barrier)
DMA_EVERY = 3      # MFMAs between DMA instructions
XDELAY = 1         # X fragment tm is re-read this many MFMAs after MFMA(7, tm)
BARRIER_AFTER = 7  # the step's barrier sits behind this MFMA index

WF = lambda tn: 128 + 8 * tn
XF = lambda tm: 192 + 8 * tm
ACC = lambda tn, tm: (tn * 8 + tm) * 4


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


def dma_block(slot_m0, kreg, first=True):
    """16 DMA instructions of one stage as a list of 16 instruction groups."""
    groups = []
    for j in range(16):
        g = []
        if j == 0:
            g += [f"s_mov_b32 m0, {slot_m0}", f"s_mov_b32 %[t0], {kreg}"]
        elif j == 8:
            g += ["s_add_u32 m0, m0, 0x1000", f"s_mov_b32 %[t0], {kreg}"]
        else:
            g += ["s_add_u32 m0, m0, 0x1000", f"s_add_u32 %[t0], %[t0], {'%[sa]' if j < 8 else '%[sb]'}"]
        g += ["s_nop 0"]
        if j < 8:
            g += ["buffer_load_dwordx4 %[va0], %[ra], %[t0] offen lds"]
        else:
            g += ["buffer_load_dwordx4 %[vb0], %[rb], %[t0] offen lds"]
        groups.append(g)
    return groups


def swap_slots():
    out = []
    for r in ("alo", "ahi", "blo", "bhi"):
        out.append(f"v_swap_b32 %[{r}_c], %[{r}_n]")
    out += ["s_mov_b32 %[t0], %[m0_c]", "s_mov_b32 %[m0_c], %[m0_n]", "s_mov_b32 %[m0_n], %[t0]"]
    return out


def canonical_prologue_reads():
    out = []
    for tn in range(7):
        out += rdW(tn, "c")
    for tm in range(8):
        out += rdX(tm, "c")
    out += rdW(7, "c")
    return out


def step(dma, reads):
    pre = [[] for _ in range(64)]
    post = [[] for _ in range(64)]
    for tm in range(8):
        pre[tm].append(f"s_waitcnt lgkmcnt({min(15, 2 * (7 - tm) + 2)})")
    if reads:
        post[BARRIER_AFTER] += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
        first_row_after_barrier = BARRIER_AFTER // 8
        for tn in range(first_row_after_barrier + 1):  # W fragments of the rows already done
            post[BARRIER_AFTER] += rdW(tn, "n")
        for tn in range(first_row_after_barrier + 1, 7):
            post[8 * tn + 7] += rdW(tn, "n")
        for tm in range(8):
            post[min(63, 56 + tm + XDELAY)] += rdX(tm, "n")
        post[63] += rdW(7, "n")
        pre[56].append("s_waitcnt lgkmcnt(14)")
    else:
        pre[56].append("s_waitcnt lgkmcnt(0)")
    if dma:
        for j, g in enumerate(dma_block("%[m0_c]", "%[k2]")):
            post[DMA_FIRST - 1 + j * DMA_EVERY] += g
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


def zero_acc():
    return [f"v_accvgpr_write_b32 a{i}, 0" for i in range(256)]


def pipelined():
    L = []
    # prologue: stages 0 and 1 in flight, accumulators cleared under their latency
    for g in dma_block("%[m0_c]", "%[k2]"):
        L += g
    L.append("s_add_u32 %[k2], %[k2], 0x80")
    for g in dma_block("%[m0_n]", "%[k2]"):
        L += g
    L.append("s_add_u32 %[k2], %[k2], 0x80")
    L += zero_acc()
    L += ["s_waitcnt vmcnt(16)", "s_barrier"]
    L += canonical_prologue_reads()
    L += ["s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 2f", "1:"]
    L += step(True, True)
    L += ["s_sub_u32 %[nloop], %[nloop], 1", "s_cmp_lg_u32 %[nloop], 0", "s_cbranch_scc1 1b", "2:"]
    L += step(False, True)
    L += step(False, False)
    L += ["s_nop 7", "s_nop 7", "s_nop 7"]
    return L


def scrub_reg(r):
    return [f"v_or_b32 %[vt0], 0x80808080, v{r}", "v_not_b32 %[vt0], %[vt0]",
            "v_add_u32 %[vt0], 0x7f7f7f7f, %[vt0]", "v_or_b32 %[vt0], 0x7f7f7f7f, %[vt0]", "v_not_b32 %[vt0], %[vt0]",
            "v_lshrrev_b32 %[vt1], 7, %[vt0]", "v_sub_u32 %[vt1], %[vt0], %[vt1]", "v_or_b32 %[vt0], %[vt0], %[vt1]",
            "v_not_b32 %[vt0], %[vt0]", f"v_and_b32 v{r}, v{r}, %[vt0]"]


def scrub_step(dma):
    L = ["s_waitcnt vmcnt(0)", "s_barrier"]
    L += canonical_prologue_reads()
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]  # everyone has read the slot: it may be refilled
    if dma:
        for g in dma_block("%[m0_c]", "%[k2]"):
            L += g
        L.append("s_add_u32 %[k2], %[k2], 0x80")
    for r in range(128, 256):
        L += scrub_reg(r)
    for i in range(64):
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
    L += ["s_cmp_eq_u32 %[nloop], 0", "s_cbranch_scc1 2f", "1:"]
    L += scrub_step(True)
    L += ["s_sub_u32 %[nloop], %[nloop], 1", "s_cmp_lg_u32 %[nloop], 0", "s_cbranch_scc1 1b", "2:"]
    L += scrub_step(False)
    L += scrub_step(False)
    L += ["s_nop 7", "s_nop 7", "s_nop 7"]
    return L


def emit(name, lines, scrub):
    print(f"#define {name}() \\")
    print("    asm volatile( \\")
    for l in lines:
        print(f'        "{l}\\n\\t" \\')
    outs = ['[alo_c] "+v"(alo_c)', '[ahi_c] "+v"(ahi_c)', '[blo_c] "+v"(blo_c)', '[bhi_c] "+v"(bhi_c)',
            '[alo_n] "+v"(alo_n)', '[ahi_n] "+v"(ahi_n)', '[blo_n] "+v"(blo_n)', '[bhi_n] "+v"(bhi_n)',
            '[k2] "+s"(k2)', '[m0_c] "+s"(m0_c)', '[m0_n] "+s"(m0_n)', '[nloop] "+s"(nloop)', '[t0] "=&s"(t0)']
    if scrub:
        outs += ['[vt0] "=&v"(vt0)', '[vt1] "=&v"(vt1)']
    ins = ['[va0] "v"(va0)', '[vb0] "v"(vb0)', '[vscale] "v"(vscale)', '[ra] "s"(ra)', '[rb] "s"(rb)', '[sa] "s"(sa)', '[sb] "s"(sb)']
    clob = [f'"v{i}"' for i in range(128, 256)] + [f'"a{i}"' for i in range(256)] + ['"scc"', '"memory"']
    print("        : " + ", ".join(outs) + " \\")
    print("        : " + ", ".join(ins) + " \\")
    print("        : " + ", ".join(clob) + ")")
    print()


def emit_readers():
    """read_acc_row<TM>: the 8 accumulator quads (tn = 0..7) of fragment row TM out of the AGPRs (register names must be literal)."""
    print("template <int TM> FP8MI_DEVICE void read_acc_row(f32x4 (&r)[8]);")
    for tm in range(8):
        print(f"template <> FP8MI_DEVICE void read_acc_row<{tm}>(f32x4 (&r)[8])")
        print("{")
        print("    float f[32];")
        for half in range(2):
            lines, outs = [], []
            for q in range(16):
                idx = half * 16 + q
                tn, j = idx // 4, idx % 4
                lines.append(f"v_accvgpr_read_b32 %{q}, a{ACC(tn, tm) + j}")
                outs.append(f'"=v"(f[{idx}])')
            print('    asm volatile("' + "\\n\\t".join(lines) + '" : ' + ", ".join(outs) + ");")
        print("    for (int tn = 0; tn < 8; ++tn) r[tn] = f32x4{f[4 * tn], f[4 * tn + 1], f[4 * tn + 2], f[4 * tn + 3]};")
        print("}")
    print()


if __name__ == "__main__":
    print("// GENERATED by csrc/gen/gen_gemm256_loop.py - do not edit; see that file for the schedule.")
    print(f"// DMA_FIRST={DMA_FIRST} DMA_EVERY={DMA_EVERY} XDELAY={XDELAY} BARRIER_AFTER={BARRIER_AFTER}")
    emit("FP8MI_GEMM256_LOOP", pipelined(), False)
    emit("FP8MI_GEMM256_LOOP_SCRUB", scrubbed(), True)
    emit_readers()
