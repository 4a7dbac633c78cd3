#!/usr/bin/env python3
"""Writes pygim_amd/csrc/lds_kernel_gen.hpp: the LDS-staged product kernels (k_lds_spmm_*) for gfx950.

The hot loop is hand-written gfx950 assembly inside a HIP kernel (the C++ prologue only turns kernel
arguments into wave-uniform pointers).  Why assembly: a wave keeps KA rows' running sums in KA vector
registers (one 64-feature slice per register, one feature per lane) and picks the register of each stored entry
with the wave-uniform VGPR index (s_set_gpr_idx_*), which HIP C++ cannot express for 200 registers.

Per stored entry (token) the wave issues
    v_bfi_b32   addr = (token & 0x3ff00) | (lane * 4)              -- row of the X chunk in LDS (the token names the buffer too)
    ds_read_b32 x    = LDS[addr]                                   -- 64 lanes x 4 B, conflict-free
    s_set_gpr_idx_idx token                                        -- M0[7:0] = accumulator index
    v_add       acc[idx] += x
which is the reference's scratchpad loop (spmm_default/dpu_kernels/spmm_mul_csr_dpu.c:108-126) with the
accumulators of ~1 500 rows resident per compute unit.  Schedule format: lds_plan.hpp.

Structure of a workgroup (NW waves, one per (tile of rows, 64-feature slice)):
    for every chunk (slot) of the tile:       -- 80 KiB (320 columns) of one slice of X, double-buffered in LDS
        touch the NEXT slot's token lines (vector load, result unused -> the scalar token loads hit L2)
        LDS-DMA this wave's share of the NEXT chunk (global_load_lds_dwordx4)
        batches of BATCH tokens, software-pipelined: the LDS reads of batch i + 1 are issued before the
            adds of batch i, the scalar load of batch i + 2 before both (three token sets in SGPRs, two x sets in VGPRs)
        s_waitcnt vmcnt(0); s_barrier
    store the accumulators to their rows of C

What was measured on the way (profiles/r03_lds_kernel.md): the first form loaded tokens one batch ahead with s_load and
waited lgkmcnt(0) per batch -- 10.1 ms, 62 % of the wave cycles in s_waitcnt: the token stream is read once, so every scalar
load paid the HBM latency, and SMEM results cannot be waited for one by one.  Touching the lines ahead through the vector
memory path (L2) brought 7.8 ms; moving the touches in front of the chunk DMA (so that the slot's closing vmcnt(0) never
waits for a young touch) and pipelining the batch loop brought the rest.
"""
import os
import sys


class Geo:
    """register geometry of one kernel variant: NW waves per workgroup, KA accumulators per wave, BATCH tokens per scalar load.
    VGPRs: temporaries v[T0..T0+7], two x / address sets of BATCH registers from X0, accumulators v[ACC0..ACC0+KA] (the last
    one is the dummy that padding tokens add into).  SGPRs: three token sets from TOK0, control registers from CTL0."""

    def __init__(self, NW, KA, BATCH, T0, weighted=False, regmap=None, X0=None, ACC0=None, dummy=True):
        self.NW, self.KA, self.BATCH, self.T0, self.weighted = NW, KA, BATCH, T0, weighted
        self.X0 = T0 + 8 if X0 is None else X0
        self.ACC0 = self.X0 + 2 * BATCH if ACC0 is None else ACC0
        self.regmap = regmap                  # explicit names of (VB, VM, VL16, VL4, VT0, VT1, VZ, VL128) instead of v[T0..T0+7]
        self.acc_regs = None                  # accumulator REGISTERS to clear (default KA [+ the dummy]); 8-byte elements: 2 per row
        self.lane_shift = 2                   # LDS base registers = lane << lane_shift (+ 64 KiB, + 128 KiB): 4-byte lanes, or 8-byte
        self.dummy = dummy                    # accumulator KA exists (the token kernels' padding tokens add into it)
        self.vmax = 512 // (NW // 4)          # VGPRs per lane at NW / 4 waves per SIMD
        assert self.ACC0 + KA + (1 if dummy else 0) <= min(self.vmax, 256), (self.ACC0 + KA + 1, self.vmax)
        self.pieces = 80 // NW                # 1 KiB DMA pieces of an 80 KiB chunk (320 columns x 256 bytes) per wave
        self.threads = NW * 64
        self.CTL0 = 88                        # control registers s88..s100; token sets below them (s32 is reserved: start at s36)
        self.TOK0 = self.CTL0 - (6 if weighted else 3) * BATCH   # weighted: three more sets for the entries' values
        assert self.TOK0 % 4 == 0 and self.TOK0 >= 36


GEOS = {8: Geo(8, 192, 16, 16), 16: Geo(16, 96, 8, 4)}
GEO_W16 = Geo(16, 96, 8, 4, weighted=True)   # same plan geometry as GEOS[16]
GEO_L16 = Geo(16, 80, 16, 4)                 # long slots (community-structured graphs): 16-token batches halve the per-batch bookkeeping
GEO_CODE = Geo(16, 96, 8, 4)                 # the code-stream kernels: chunks of 192 columns (48 KiB, three buffers): 3 DMA pieces per wave
GEO_CODE.pieces = 48 // 16
# the 8-wave code-stream kernels (round 4): 228 accumulators per wave, 2 waves per SIMD with 256 VGPRs each -- 1 824-row tiles, so that
# the Reddit-shaped product of four slices is two rounds of workgroups instead of three.  Register map = lds_plan.hpp lds_code_regs(8):
# v0 lane id (the compiler's), v1..v3 LDS bases, v4 lane * 16 (DMA and touch), v5 touch destination, x v6..v27, accumulators v28..v255;
# the store stage borrows x registers (v8, v9 temporaries; v6, v7 epilogue factors)
GEO_CODE8 = Geo(8, 228, 8, 1, regmap=("v1", "v2", "v4", "v1", "v8", "v9", "v3", "v4"), X0=6, ACC0=28, dummy=False)
GEO_CODE8.pieces = 0   # (taken from the plan: LdsArgs.piece_bytes)
# ... and its 8-byte form (INT64 / DBL64): 114 rows per wave, a register pair per running sum, rows of 512 bytes in LDS (lane * 8)
GEO_CODE8_64 = Geo(8, 114, 8, 1, regmap=("v1", "v2", "v4", "v1", "v8", "v9", "v3", "v4"), X0=6, ACC0=28, dummy=False)
GEO_CODE8_64.pieces = 0
GEO_CODE8_64.acc_regs = 228
GEO_CODE8_64.lane_shift = 3


def body(op_add, g, ablate=0, op_mul=None, deq=None, code=False, out_kind=None, stamps=False):
    """ablate (timing experiments only, results wrong): 1 = one accumulator index per batch, 2 = no LDS reads,
    3 = no address computation and no LDS reads, 4 = no accumulation, 5 = no scalar token loads inside the batch loop,
    6 = no workgroup barrier, 7 = no chunk DMA, 8 = no token-line touches; 9 = (correct results) LDS reads interleaved with the adds"""
    # codes 10..12: the bare batch loop (no scalar token loads, barrier, DMA or touches), whole / LDS side only / accumulate side only
    TNT = " nt" if ablate in (17, 19) else ""     # experiment (results stay right): the token-line touches bypass the L1
    DNT = " nt" if ablate in (18, 19) else ""     # experiment: the chunk DMA bypasses the L1
    AB = {0: set(), 17: set(), 18: set(), 19: set(), 10: {5, 6, 7, 8}, 11: {5, 6, 7, 8, 4}, 12: {5, 6, 7, 8, 3}}.get(ablate, {ablate})
    KA, BATCH, NW, ACC0 = g.KA, g.BATCH, g.NW, g.ACC0
    VB, VM, VL16, VL4, VT0, VT1, VZ, VL128 = g.regmap or tuple(f"v{g.T0 + i}" for i in range(8))  # lane*4+buffer, mask, lane*16, lane*4, 2 tmp, zero, lane*128
    XS = [g.X0, g.X0 + BATCH]
    TOK = [g.TOK0 + i * BATCH for i in range(3)]
    WGT = [g.TOK0 + (3 + i) * BATCH for i in range(3)] if g.weighted else None
    assert bool(op_mul) == bool(g.weighted)
    c = g.CTL0
    # control SGPRs
    NBL, NLEFT, CIDN, TMP = (f"s{c + i}" for i in range(4))              # batches left, slots left, next chunk id, temporary
    PA_LO, PA_HI, PA = f"s{c + 4}", f"s{c + 5}", f"s[{c + 4}:{c + 5}]"   # 64-bit temporary (touch / DMA source address); even
    DLDS, BUF = f"s{c + 6}", f"s{c + 7}"                                   # DMA LDS base (also a temporary), buffer select (0 / 0x14000: the DMA destination; the reads take the buffer from the token)
    TMP2 = DLDS
    TP_LO, TP_HI, TP = f"s{c + 8}", f"s{c + 9}", f"s[{c + 8}:{c + 9}]"      # token stream of this wave (fixed)
    NBN, TOFF, ROT = (f"s{c + 10 + i}" for i in range(3))                  # batches of the next slot, token offset (newest loaded batch), token-set rotation
    assert c + 12 <= 101 and c % 2 == 0
    # result stage (aliases of loop registers)
    KREG, ROWID, EX_LO = NBL, NLEFT, c + 2
    EX = f"s[{EX_LO}:{EX_LO + 1}]"
    NP_LO, NP_HI, NP = TP_LO, TP_HI, TP                                    # (the row map is walked with the token pointer's registers)
    tokload = f"s_load_dwordx{BATCH}"
    L = []
    a = L.append

    def header(r):
        # slot header (lds_plan.hpp): upper halves of tokens 0..2 of the slot's first batch, here in token set r
        a(f"s_lshr_b32 {NBL}, s{TOK[r]}, 18")         # batches of the slot
        a(f"s_lshr_b32 {NBN}, s{TOK[r] + 1}, 18")     # batches of the slot after it (token prefetch)
        a(f"s_lshr_b32 {CIDN}, s{TOK[r] + 2}, 18")    # chunk id of the slot after it (DMA)

    def stamp(k):
        # measurement build (k_*_ts, tunable lds_stamp): the constant 100 MHz clock at four points of a wave's life -> stamps[(block * NW + wave) * 4 + k].
        # v8 / v9 / v10 are x registers: free before the stream starts and after it returns; s[72:73] are free in the shell
        if not stamps:
            return
        a("s_memrealtime s[72:73]")
        a("s_waitcnt lgkmcnt(0)")
        a("v_mov_b32 v8, s72")
        a("v_mov_b32 v9, s73")
        a("v_mov_b32 v10, 0")
        a(f"global_store_dwordx2 v10, v[8:9], %[stamps] offset:{8 * k}")

    stamp(0)
    # ---- set-up
    a(f"v_lshlrev_b32 {VL4}, {g.lane_shift}, %[lane]")
    if VL128 != VL16:
        a(f"v_lshlrev_b32 {VL128}, 7, %[lane]")
    a(f"v_lshlrev_b32 {VL16}, 4, %[lane]")
    a(f"v_mov_b32 {VM}, 0x3ff00")
    a(f"v_lshlrev_b32 {VB}, {g.lane_shift}, %[lane]")       # (the token carries the LDS row of both buffers: the lane offset is all that is added)
    a(f"v_mov_b32 {VZ}, 0")
    for i in range(g.acc_regs or (KA + (1 if g.dummy else 0))):
        a(f"v_mov_b32 v{ACC0 + i}, 0")
    if code:
        # CODE-STREAM form (lds_plan.hpp lds_code_from_plan): the whole slot loop -- DMA of the chunks, the entries' LDS reads and
        # adds, waits and barriers -- is ONE straight-line instruction stream per (tile, wave), compiled from the schedule when the
        # group is created; the kernel sets up its registers (LdsCodeRegs), jumps into it and comes back for the store stage
        a(f"v_add_u32 {VM}, 0x10000, {VB}")        # LDS rows 256..511
        a(f"v_add_u32 {VZ}, 0x20000, {VB}")        # LDS rows 512..639
        a("s_icache_inv")                          # (the code of a freed group may have lived at these addresses)
        a("s_cmp_eq_u32 %[nch], 0")
        a("s_cbranch_scc1 L_out_%=")
        a("s_mov_b32 s80, %[xs_lo]")
        a("s_mov_b32 s81, %[xs_hi]")
        a("s_mov_b32 s82, %[ldsw]")
        a("s_mov_b64 s[84:85], %[code]")
        a("s_add_u32 s84, s84, 0x800")             # the stream touches its own lines 2 KiB ahead, 8 lines (1 KB) per touch:
        if VL128 != VL16:
            a(f"v_and_b32 {VL128}, 7, %[lane]")    # lane offsets (lane % 8) * 128
            a(f"v_lshlrev_b32 {VL128}, 7, {VL128}")  # (the 8-wave shell touches with lane * 16: the same 8 lines)
        a("s_addc_u32 s85, s85, 0")
        # half-split plans (round 6, lds_plan.hpp LdsGeometry::half_split): the stream adds under these two masks -- lanes 0..31 / lanes 32..63
        a("s_mov_b32 s76, -1")
        a("s_mov_b32 s77, 0")
        a("s_mov_b32 s78, 0")
        a("s_mov_b32 s79, -1")
        stamp(1)
        a("s_swappc_b64 s[86:87], %[code]")
        a("s_branch L_out_%=")                     # (the token loop below is not part of this form)
    cut = len(L)
    a(f"s_mov_b64 {TP}, %[tok]")
    a(f"s_mov_b32 {NLEFT}, %[nch]")
    a(f"s_mov_b32 {BUF}, 0")
    a(f"s_mov_b32 {ROT}, 0")
    a(f"s_mov_b32 {TOFF}, 0")
    a(f"s_cmp_eq_u32 {NLEFT}, 0")
    a("s_cbranch_scc1 L_out_%=")
    a(f"{tokload} s[{TOK[0]}:{TOK[0] + BATCH - 1}], {TP}, 0x0")
    if g.weighted:
        a(f"s_mov_b32 {TMP}, %[wdelta]")
        a(f"{tokload} s[{WGT[0]}:{WGT[0] + BATCH - 1}], {TP}, {TMP}")

    next_id = [0]

    def dma(cid, bufsel_reg):
        # this wave's 1 KiB pieces of the 80 KiB chunk `cid` -> LDS buffer `bufsel`
        a(f"s_mul_i32 {TMP}, {cid}, 0x14000")           # 80 KiB per chunk (chunk ids are 14-bit: no overflow)
        a(f"s_add_u32 {PA_LO}, %[xs_lo], {TMP}")
        a(f"s_addc_u32 {PA_HI}, %[xs_hi], 0")
        a(f"s_add_u32 {DLDS}, {bufsel_reg}, %[ldsw]")
        # one M0 per four pieces: the instruction offset moves the global AND the LDS address of an LDS-DMA load (measured: the
        # product stays bit-exact; 26 -> 10 instructions per wave and chunk, 4.20 -> 4.00 ms)
        for i in range(g.pieces):
            if i % 4 == 0:
                if i:
                    a(f"s_add_u32 {PA_LO}, {PA_LO}, 0x1000")
                    a(f"s_addc_u32 {PA_HI}, {PA_HI}, 0")
                    a(f"s_add_u32 {DLDS}, {DLDS}, 0x1000")
                a(f"s_mov_b32 m0, {DLDS}")
                a("s_nop 0")
            if 7 not in AB:
                a(f"global_load_lds_dwordx4 {VL16}, {PA} offset:{(i % 4) * 1024}{DNT}")

    if 15 in AB:   # experiment: static priority for the later-dispatched half of the workgroup (waves NW/2 ..)
        a("s_cmp_ge_u32 %[wave], " + str(NW // 2))
        a("s_cbranch_scc0 L_noprio_%=")
        a("s_setprio 1")
        a("L_noprio_%=:")
    a(f"s_mov_b32 {CIDN}, %[cid0]")
    dma(CIDN, BUF)
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    header(0)
    a("s_barrier")
    # ---- slot loop: NBL / NBN / CIDN hold this slot's batches, the next slot's batches and the next slot's chunk id
    a("L_slot_%=:")
    # The NEXT chunk first (LDS-DMA), then ONE touch: a vector load (result unused) of the first lines of the NEXT slot's tokens, so
    # that the scalar loads of the batch loop hit the XCD's L2 -- lane l touches line l, at most 17 lines (2 KiB); longer slots
    # touch 2 KiB ahead batch by batch (below).  vmcnt retires in issue order, so the slot's closing s_waitcnt vmcnt(NT) (all but
    # the NT youngest) covers the DMA without waiting for the touch.  TP + TOFF = this slot's first batch.
    a(f"s_cmp_gt_u32 {NLEFT}, 1")
    a("s_cbranch_scc0 L_nodma_%=")
    a(f"s_xor_b32 {TMP2}, {BUF}, 0x14000")
    dma(CIDN, TMP2)
    a("L_nodma_%=:")
    a(f"s_mul_i32 {TMP}, {NBL}, {BATCH * 4}")
    a(f"s_add_u32 {TMP}, {TMP}, {TOFF}")
    a(f"s_add_u32 {PA_LO}, {TP_LO}, {TMP}")
    a(f"s_addc_u32 {PA_HI}, {TP_HI}, 0")
    a(f"s_add_u32 {TMP}, {NBN}, {2 * (128 // (BATCH * 4)) - 1}")
    a(f"s_lshr_b32 {TMP}, {TMP}, {(128 // (BATCH * 4)).bit_length() - 1}")   # 128-byte lines of the next slot's tokens, + 1
    a(f"s_min_u32 {TMP}, {TMP}, 17")
    a(f"s_bfm_b64 exec, {TMP}, 0")
    if 8 not in AB:
        a(f"global_load_dword {VT1}, {VL128}, {PA}{TNT}")
    if g.weighted:   # the same lines of the value stream (it sits wdelta bytes behind the tokens)
        a(f"s_add_u32 {PA_LO}, {PA_LO}, %[wdelta]")
        a(f"s_addc_u32 {PA_HI}, {PA_HI}, 0")
        a(f"global_load_dword {VT1}, {VL128}, {PA}")
    a("s_mov_b64 exec, -1")
    a(f"s_sub_u32 {NBL}, {NBL}, 1")               # batches left after the one in hand (every list has at least its header batch)
    a(f"s_cmp_eq_u32 {ROT}, 0")
    a("s_cbranch_scc1 L_E0_%=")
    a(f"s_cmp_eq_u32 {ROT}, 1")
    a("s_cbranch_scc1 L_E1_%=")
    a("s_branch L_E2_%=")

    def load_next(r, touch=True):
        a(f"s_add_u32 {TOFF}, {TOFF}, {hex(BATCH * 4)}")
        if 5 not in AB:
            a(f"{tokload} s[{TOK[r]}:{TOK[r] + BATCH - 1}], {TP}, {TOFF}")
        if g.weighted:
            a(f"s_add_u32 {TMP}, {TOFF}, %[wdelta]")
            a(f"{tokload} s[{WGT[r]}:{WGT[r] + BATCH - 1}], {TP}, {TMP}")
        if not touch:
            return
        # long slots: the line 2 KiB ahead (all lanes one address = one request), except in the slot's last batches
        # (a touch issued there would still be in flight at the slot's closing vmcnt(0)).  With 8-token batches (32 bytes)
        # only every third body carries the touch: consecutive touches are then 96 bytes apart, no 128-byte line is skipped
        a(f"s_cmp_gt_u32 {NBL}, {max(4, 128 // BATCH)}")
        a(f"s_cbranch_scc0 L_nt{next_id[0]}_%=")
        a(f"v_mov_b32 {VT0}, {TOFF}")
        if 8 not in AB:
            a(f"global_load_dword {VT1}, {VT0}, {TP} offset:2048{TNT}")
        if g.weighted:
            a(f"v_add_u32 {VT0}, %[wdelta], {VT0}")
            a(f"global_load_dword {VT1}, {VT0}, {TP} offset:2048")
        a(f"L_nt{next_id[0]}_%=:")
        next_id[0] += 1

    def reads(r, x):
        if 3 not in AB:
            for i in range(BATCH):
                a(f"v_bfi_b32 v{XS[x] + i}, {VM}, s{TOK[r] + i}, {VB}")
        if not (AB & {2, 3}):
            for i in range(BATCH):
                a(f"ds_read_b32 v{XS[x] + i}, v{XS[x] + i}")

    def reads_adds_interleaved(rn, xn, r, x):
        # the next batch's LDS reads between an accumulator-index write and the add that depends on it
        for i in range(BATCH):
            a(f"v_bfi_b32 v{XS[xn] + i}, {VM}, s{TOK[rn] + i}, {VB}")
        if g.weighted:
            for i in range(BATCH):
                a(f"{op_mul} v{XS[x] + i}, s{WGT[r] + i}, v{XS[x] + i}")
        a(f"s_set_gpr_idx_on s{TOK[r]}, gpr_idx(SRC1,DST)")
        for i in range(BATCH):
            if i:
                a(f"s_set_gpr_idx_idx s{TOK[r] + i}")
            a(f"ds_read_b32 v{XS[xn] + i}, v{XS[xn] + i}")
            a(f"{op_add} v{ACC0}, v{XS[x] + i}, v{ACC0}")
        a("s_set_gpr_idx_off")

    def adds(r, x):
        if 4 in AB:
            return
        if g.weighted:   # product and sum round separately, as in the CPU loop (no FMA)
            for i in range(BATCH):
                a(f"{op_mul} v{XS[x] + i}, s{WGT[r] + i}, v{XS[x] + i}")
        if 16 in AB:   # experiment: the accumulate phase at raised priority
            a("s_setprio 2")
        a(f"s_set_gpr_idx_on s{TOK[r]}, gpr_idx(SRC1,DST)")
        a(f"{op_add} v{ACC0}, v{XS[x]}, v{ACC0}")
        for i in range(1, BATCH):
            if 1 not in AB:
                a(f"s_set_gpr_idx_idx s{TOK[r] + i}")
            a(f"{op_add} v{ACC0}, v{XS[x] + i}, v{ACC0}")
        a("s_set_gpr_idx_off")
        if 16 in AB:
            a("s_setprio 0")

    def count_and_exit(r_next, x_next):
        a(f"s_sub_u32 {NBL}, {NBL}, 1")               # SCC = borrow: no batch was left
        a(f"s_cbranch_scc1 L_D{r_next}{x_next}_%=")

    # entries: tokens of the slot's first batch sit in set r
    for r in (1, 2, 0):
        a(f"L_E{r}_%=:")
        load_next((r + 1) % 3)
        reads(r, 0)
        count_and_exit(r, 0)
        if r != 0:
            a(f"s_branch L_B{r}0_%=")
    # (E0 falls through into B00) steady-state bodies in cycle order: (0,A) (1,B) (2,A) (0,B) (1,A) (2,B)
    rr, xx = 0, 0
    for _ in range(6):
        a(f"L_B{rr}{xx}_%=:")
        a("s_waitcnt lgkmcnt(0)")            # batch i's reads (issued a phase ago) and batch i + 1's tokens are in
        load_next((rr + 2) % 3, touch=(rr == 0 or BATCH * 4 * 3 > 128))
        if ablate == 9:
            reads_adds_interleaved((rr + 1) % 3, 1 - xx, rr, xx)
        else:
            reads((rr + 1) % 3, 1 - xx)
            adds(rr, xx)
        rr, xx = (rr + 1) % 3, 1 - xx
        count_and_exit(rr, xx)
    a("s_branch L_B00_%=")
    # drains: the last batch of the slot (tokens in set r, reads in flight in x-set x)
    for r in range(3):
        for x in range(2):
            a(f"L_D{r}{x}_%=:")
            a("s_waitcnt lgkmcnt(0)")
            adds(r, x)
            header((r + 1) % 3)                  # the next slot's first batch is in (loaded a body ago, waited for above)
            a(f"s_mov_b32 {ROT}, {(r + 1) % 3}")
            a("s_branch L_slotend_%=")
    a("L_slotend_%=:")
    nt = (2 if g.weighted else 1) - (1 if 8 in AB else 0)
    a(f"s_waitcnt vmcnt({nt}) lgkmcnt(0)")        # my pieces of the next chunk have landed (the touch behind them may not have)
    if 6 not in AB:
        a("s_barrier")                           # ... and everybody is done reading the current one
    a(f"s_xor_b32 {BUF}, {BUF}, 0x14000")
    a(f"s_sub_u32 {NLEFT}, {NLEFT}, 1")
    a(f"s_cmp_gt_u32 {NLEFT}, 0")
    a("s_cbranch_scc1 L_slot_%=")
    # ---- results: acc[k] -> C[rowmap[k]] (lanes beyond the slice's width masked off)
    if code:
        del L[cut - 1:]                            # (nothing between the stream's return and the store stage)
    a("L_out_%=:")
    stamp(2)
    # Row ids of the wave's KA accumulators.  Token kernels: eight per scalar load of the row map, each load waited for (29 round trips to
    # memory per wave).  Code-stream kernels (round 6; the stamps of profiles/r06_stamps.txt: 22 us of every workgroup's life were this
    # loop, the loads' latency): the whole row map of the wave comes in ONE round trip -- lane l of v[RM + j] = rowmap[64 j + l], up to four
    # vector loads in flight together -- and the loop takes its eight ids per step out of the registers with v_readlane_b32.
    ka_stride = (KA + 7) & ~7
    RM = 20                                           # v20..v23: x registers, free once the stream has returned
    nblk = (ka_stride + 63) // 64 if code else 1
    if code:
        assert g.X0 <= RM and RM + 4 + 1 <= ACC0 and nblk <= 4
        a("s_mov_b64 exec, -1")
        a(f"v_lshlrev_b32 v{RM + 4}, 2, %[lane]")
        for j in range(nblk):
            cnt = min(64, ka_stride - 64 * j)
            if cnt < 64:
                a(f"s_bfm_b64 exec, {cnt}, 0")     # (nothing is read past the wave's part of the row map)
            a(f"global_load_dword v{RM + j}, v{RM + 4}, %[rowmap] offset:{256 * j}")
        a("s_mov_b64 exec, -1")
        a("s_waitcnt vmcnt(0)")

    def row_loop(row_body, tail):
        """the loop over the wave's accumulators, eight per step: row_body(i, rid, tag) stores accumulator KREG + i to row `rid` of C"""
        for j in range(nblk):
            a(f"s_mov_b32 {KREG}, {64 * j}")
            a(f"L_orow{j}_%=:")
            if code:
                for i in range(8):
                    a(f"s_sub_u32 {DLDS}, {KREG}, {64 * j - i}" if 64 * j - i >= 0 else f"s_add_u32 {DLDS}, {KREG}, {i - 64 * j}")
                    a(f"v_readlane_b32 s{TOK[0] + i}, v{RM + j}, {DLDS}")
            else:
                a(f"s_load_dwordx8 s[{TOK[0]}:{TOK[0] + 7}], {NP}, 0x0")
                a("s_waitcnt lgkmcnt(0)")
            for i in range(8):
                row_body(i, f"s{TOK[0] + i}", f"{j}{i}")
            if not code:
                a(f"s_add_u32 {NP_LO}, {NP_LO}, 32")
                a(f"s_addc_u32 {NP_HI}, {NP_HI}, 0")
            a(f"s_add_u32 {KREG}, {KREG}, 8")
            a(f"s_cmp_lt_u32 {KREG}, {min(64 * j + 64, ka_stride) if code else KA}")
            a(f"s_cbranch_scc1 L_orow{j}_%=")
        tail()

    if out_kind in ("f64", "i64"):
        # 8-byte elements (round 4): row k's running sum is the register pair v[ACC0 + 2k : ACC0 + 2k + 1], one feature per lane (64 to a
        # slice of 512 bytes); the store writes 8 bytes per lane, or adds into C first (v_add_f64 / a 64-bit integer add)
        assert code and g.regmap and VT0 == "v8" and VT1 == "v9"
        a(f"s_mov_b64 {EX}, exec")
        a("v_cmp_gt_u32 vcc, %[wvalid], %[lane]")
        a("s_and_b64 exec, exec, vcc")
        a(f"s_mov_b64 {NP}, %[rowmap]")

        def row64(i, rid, tag):
            a(f"s_cmp_eq_u32 {rid}, -1")
            a(f"s_cbranch_scc1 L_oskip{tag}_%=")
            a(f"s_mul_i32 {PA_LO}, {rid}, %[ldc]")
            a(f"s_mul_hi_u32 {PA_HI}, {rid}, %[ldc]")
            a(f"s_add_u32 {PA_LO}, {PA_LO}, %[c_lo]")
            a(f"s_addc_u32 {PA_HI}, {PA_HI}, %[c_hi]")
            a(f"s_add_u32 {DLDS}, {KREG}, {i}")
            a(f"s_lshl_b32 {DLDS}, {DLDS}, 1")
            a(f"s_set_gpr_idx_on {DLDS}, gpr_idx(SRC0)")
            a(f"v_mov_b32 v8, v{ACC0}")
            a(f"v_mov_b32 v9, v{ACC0 + 1}")
            a("s_set_gpr_idx_off")
            a("s_cmp_eq_u32 %[accum], 0")
            a(f"s_cbranch_scc1 L_ost{tag}_%=")
            a(f"global_load_dwordx2 v[16:17], {VB}, {PA}")
            a("s_waitcnt vmcnt(0)")
            if out_kind == "f64":
                a("v_add_f64 v[8:9], v[16:17], v[8:9]")
            else:
                a("v_add_co_u32 v8, vcc, v16, v8")
                a("v_addc_co_u32 v9, vcc, v17, v9, vcc")
            a(f"L_ost{tag}_%=:")
            a(f"global_store_dwordx2 {VB}, v[8:9], {PA}")
            a(f"L_oskip{tag}_%=:")

        def tail64():
            a(f"s_mov_b64 exec, {EX}")
            a("s_waitcnt vmcnt(0)")

        row_loop(row64, tail64)
        stamp(3)
        return L
    if out_kind in ("i8", "i8_deq", "i16_deq"):
        # "i16_deq": the INT16 stream's own dequantising store (sign-extended 16-bit halves), same shape as "i8_deq"
        # INT8 through the INT16 stream (round 4): the slice-major copy holds the int8 features WIDENED to 16 bits (a slice = 128 features
        # = the same 256 bytes), the stream's v_pk_add_u16 sums wrap modulo 2^16, and the low byte of each half IS the modular int8 sum
        # (models/quantize.py:22-23 quantises to int8; torch's int8 sums wrap the same way).  A lane holds features 2l and 2l + 1:
        # "i8":     two byte stores per lane (features beyond the width masked off one by one: widths need not be even)
        # "i8_deq": sign-extend, float(sum) * scale, the optional per-column epilogue, two dword stores per lane
        assert code and g.regmap
        VL2, VL8, T2, T3, VP, VQ = "v10", "v11", "v16", "v17", 12, 14
        EXA, EXB = "s[80:81]", "s[82:83]"                   # (the stream's registers are free again)
        a(f"s_mov_b64 {EX}, exec")
        a(f"v_lshlrev_b32 {VL2}, 1, %[lane]")
        a(f"v_lshlrev_b32 {VL8}, 3, %[lane]")
        ST = "s84"                                          # (TMP is half of the saved exec mask in this stage)
        a(f"s_add_u32 {ST}, %[wvalid], 1")                  # [wvalid] = features of this slice (<= 128)
        a(f"s_lshr_b32 {ST}, {ST}, 1")
        a(f"v_cmp_gt_u32 vcc, {ST}, %[lane]")               # lane l holds a valid feature 2l
        a(f"s_mov_b64 {EXA}, vcc")
        a(f"s_lshr_b32 {ST}, %[wvalid], 1")
        a(f"v_cmp_gt_u32 vcc, {ST}, %[lane]")               # ... and a valid feature 2l + 1
        a(f"s_mov_b64 {EXB}, vcc")
        a(f"s_mov_b64 {NP}, %[rowmap]")
        a(f"s_mov_b32 {KREG}, 0")
        if out_kind in ("i8_deq", "i16_deq"):
            a("s_cmp_eq_u64 %[pmul], 0")
            a("s_cbranch_scc1 L_nopost_%=")
            a(f"s_mov_b64 exec, {EXA}")                        # (nothing is read past the last column's factor)
            a(f"global_load_dword v{VP}, {VL8}, %[pmul]")
            a(f"global_load_dword v{VQ}, {VL8}, %[padd]")
            a(f"s_mov_b64 exec, {EXB}")
            a(f"global_load_dword v{VP + 1}, {VL8}, %[pmul] offset:4")
            a(f"global_load_dword v{VQ + 1}, {VL8}, %[padd] offset:4")
            a(f"s_mov_b64 exec, {EX}")
            a("s_waitcnt vmcnt(0)")
            a("L_nopost_%=:")
        def row8(i, rid, tag):
            a(f"s_cmp_eq_u32 {rid}, -1")
            a(f"s_cbranch_scc1 L_oskip{tag}_%=")
            a(f"s_mul_i32 {PA_LO}, {rid}, %[ldc]")
            a(f"s_mul_hi_u32 {PA_HI}, {rid}, %[ldc]")
            a(f"s_add_u32 {PA_LO}, {PA_LO}, %[c_lo]")
            a(f"s_addc_u32 {PA_HI}, {PA_HI}, %[c_hi]")
            a(f"s_add_u32 {DLDS}, {KREG}, {i}")
            a(f"s_set_gpr_idx_on {DLDS}, gpr_idx(SRC0)")
            a(f"v_mov_b32 {VT0}, v{ACC0}")
            a("s_set_gpr_idx_off")
            if out_kind == "i8":
                a(f"s_mov_b64 exec, {EXA}")
                a(f"global_store_byte {VL2}, {VT0}, {PA}")
                a(f"s_mov_b64 exec, {EXB}")
                a(f"global_store_byte_d16_hi {VL2}, {VT0}, {PA} offset:1")
                a(f"s_mov_b64 exec, {EX}")
            else:
                bits = 8 if out_kind == "i8_deq" else 16
                a(f"v_bfe_i32 {T2}, {VT0}, 0, {bits}")
                a(f"v_bfe_i32 {T3}, {VT0}, 16, {bits}")
                a(f"v_cvt_f32_i32 {T2}, {T2}")
                a(f"v_cvt_f32_i32 {T3}, {T3}")
                a(f"v_mul_f32 {T2}, %[scale], {T2}")
                a(f"v_mul_f32 {T3}, %[scale], {T3}")
                a("s_cmp_eq_u64 %[pmul], 0")
                a(f"s_cbranch_scc1 L_np{tag}_%=")
                a(f"v_mul_f32 {T2}, v{VP}, {T2}")
                a(f"v_mul_f32 {T3}, v{VP + 1}, {T3}")
                a(f"v_add_f32 {T2}, v{VQ}, {T2}")
                a(f"v_add_f32 {T3}, v{VQ + 1}, {T3}")
                a("s_cmp_eq_u32 %[relu], 0")
                a(f"s_cbranch_scc1 L_np{tag}_%=")
                a(f"v_max_f32 {T2}, 0, {T2}")
                a(f"v_max_f32 {T3}, 0, {T3}")
                a(f"L_np{tag}_%=:")
                a(f"s_mov_b64 exec, {EXA}")
                a(f"global_store_dword {VL8}, {T2}, {PA}")
                a(f"s_mov_b64 exec, {EXB}")
                a(f"global_store_dword {VL8}, {T3}, {PA} offset:4")
                a(f"s_mov_b64 exec, {EX}")
            a(f"L_oskip{tag}_%=:")

        def tail8():
            a(f"s_mov_b64 exec, {EX}")
            a("s_waitcnt vmcnt(0)")

        row_loop(row8, tail8)
        stamp(3)
        return L
    a(f"s_mov_b64 {EX}, exec")
    a("v_cmp_gt_u32 vcc, %[wvalid], %[lane]")
    a("s_and_b64 exec, exec, vcc")
    a(f"s_mov_b64 {NP}, %[rowmap]")
    a(f"s_mov_b32 {KREG}, 0")
    if deq:
        # per-column epilogue factors of this slice's 64 features, one per lane (the x registers are free by now)
        a("s_cmp_eq_u64 %[pmul], 0")
        a("s_cbranch_scc1 L_nopost_%=")
        a(f"global_load_dword v{XS[0]}, {VL4}, %[pmul]")
        a(f"global_load_dword v{XS[0] + 1}, {VL4}, %[padd]")
        a("s_waitcnt vmcnt(0)")
        a("L_nopost_%=:")
    def row4(i, rid, tag):
        a(f"s_cmp_eq_u32 {rid}, -1")
        a(f"s_cbranch_scc1 L_oskip{tag}_%=")
        a(f"s_mul_i32 {PA_LO}, {rid}, %[ldc]")
        a(f"s_mul_hi_u32 {PA_HI}, {rid}, %[ldc]")
        a(f"s_add_u32 {PA_LO}, {PA_LO}, %[c_lo]")
        a(f"s_addc_u32 {PA_HI}, {PA_HI}, %[c_hi]")
        a(f"s_add_u32 {DLDS}, {KREG}, {i}")            # (TMP holds half of the saved exec mask here)
        if code and not deq:
            a("s_mov_b64 exec, -1")                    # (a half-split plan needs the accumulator's upper lanes too: the copy below is made by every lane)
        a(f"s_set_gpr_idx_on {DLDS}, gpr_idx(SRC0)")
        a(f"v_mov_b32 {VT0}, v{ACC0}")
        a("s_set_gpr_idx_off")
        if code and not deq:
            # half-split plans: lanes 0..31 hold the row's sum over the lower column range, lanes 32..63 over the upper one -- the row's sum is their sum
            # (v_permlane32_swap_b32 vdst, src: lanes 0..31 of vdst <-> lanes 32..63 of src; with two copies of the accumulator one ends up [hi | hi], the
            # other [lo | lo]); every lane takes part, the store below is masked again (vcc still holds the lanes of this slice)
            a("s_cmp_eq_u32 %[hsplit], 0")
            a(f"s_cbranch_scc1 L_nh{tag}_%=")
            a(f"v_mov_b32 {VT1}, {VT0}")
            a("s_nop 1")
            a(f"v_permlane32_swap_b32 {VT0}, {VT1}")
            a("s_nop 1")
            a(f"{op_add} {VT0}, {VT1}, {VT0}")
            a(f"L_nh{tag}_%=:")
            a(f"s_and_b64 exec, {EX}, vcc")
        if deq:
            # the conv layers' dequantisation in the store (models/quantize.py:35-38): float(sum) * scale, as the sweep's fused store
            if deq == "i32":
                a(f"v_cvt_f32_i32 {VT0}, {VT0}")
            a(f"v_mul_f32 {VT0}, %[scale], {VT0}")
            a("s_cmp_eq_u64 %[pmul], 0")
            a(f"s_cbranch_scc1 L_np{tag}_%=")
            a(f"v_mul_f32 {VT0}, v{XS[0]}, {VT0}")         # product and sum rounded separately, as k_post_affine / torch's a * y + b
            a(f"v_add_f32 {VT0}, v{XS[0] + 1}, {VT0}")
            a("s_cmp_eq_u32 %[relu], 0")
            a(f"s_cbranch_scc1 L_np{tag}_%=")
            a(f"v_max_f32 {VT0}, 0, {VT0}")
            a(f"L_np{tag}_%=:")
        else:
            a("s_cmp_eq_u32 %[accum], 0")
            a(f"s_cbranch_scc1 L_ost{tag}_%=")
            a(f"global_load_dword {VT1}, {VL4}, {PA}")
            a("s_waitcnt vmcnt(0)")
            a(f"{op_add} {VT0}, {VT1}, {VT0}")
            a(f"L_ost{tag}_%=:")
        a(f"global_store_dword {VL4}, {VT0}, {PA}")
        a(f"L_oskip{tag}_%=:")

    def tail4():
        a(f"s_mov_b64 exec, {EX}")
        a("s_waitcnt vmcnt(0)")

    row_loop(row4, tail4)
    stamp(3)
    if stamps:
        a("s_waitcnt vmcnt(0)")
    return L


HEADER = """// GENERATED by scripts/gen_lds_kernel.py -- do not edit; edit the generator and re-run it.
// LDS-staged product kernels for gfx950: schedule format in lds_plan.hpp, design notes in the generator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lds_plan.hpp"

namespace pygim {

constexpr uint32_t LDS_KC = 320, LDS_BYTES = 163840;   // two 80 KiB chunk buffers = the whole LDS of a CU
// the code-stream kernels take the ring geometry from the plan (LdsArgs.piece_bytes): two buffers of 320 columns, or -- products of
// one or two slices, whose workgroups do little but land chunks -- three of 192 (two chunks in flight while one is read)
constexpr uint32_t LDS_CODE_KC3 = 192;
// kernel variants: waves per workgroup -> accumulators per wave, tokens per batch
constexpr uint32_t lds_ka(uint32_t nw) { return nw == 16 ? %(KA16)du : %(KA8)du; }
constexpr uint32_t lds_batch(uint32_t nw) { return nw == 16 ? %(B16)du : %(B8)du; }
// ... and the 16-wave geometry for long slots (16-token batches, fewer accumulators)
constexpr uint32_t LDS_L16_KA = %(KAL)du, LDS_L16_BATCH = %(BL)du;

// the code-stream shells below and lds_plan.hpp's emitter agree on the register map
static_assert(lds_code_regs(16).vbase[0] == 4 && lds_code_regs(16).vbase[1] == 5 && lds_code_regs(16).vbase[2] == 10 && lds_code_regs(16).vl16 == 6 &&
              lds_code_regs(16).vtouch == 11 && lds_code_regs(16).vjunk == 9 && lds_code_regs(16).x0 == 12 && lds_code_regs(16).acc0 == 28, "16-wave code map");
static_assert(lds_code_regs(8).vbase[0] == 1 && lds_code_regs(8).vbase[1] == 2 && lds_code_regs(8).vbase[2] == 3 && lds_code_regs(8).vl16 == 4 &&
              lds_code_regs(8).vtouch == 4 && lds_code_regs(8).vjunk == 5 && lds_code_regs(8).x0 == 6 && lds_code_regs(8).acc0 == 28 &&
              LDS_CODE8_KA == %(KAC8)d, "8-wave code map");

struct LdsArgs {
    const uint32_t *tok;      // token streams (with the slot headers, lds_plan.hpp)
    const LdsTile *tiles;
    const uint32_t *rowmap;   // [ntiles][NW][KA] row of C (0xffffffff = none)
    const char *xs;           // slice-major X: [nslices][ncols_pad][64] elements of 4 bytes
    char *c;                  // row-major result
    uint64_t slice_stride;    // bytes between two slices of xs
    uint32_t ldc_bytes, w, nslices, ntiles, accumulate, xcd_group;   // w: width of the product in LANES (4-byte units)
    uint32_t wdelta;          // weighted kernels: bytes from a token to its value (the value stream follows the token stream)
    const uint32_t *deq_amax; // dequantising kernels: bits of max|x| (device), and the quantiser's log2 range
    int deq_log2;
    const float *post_mul, *post_add;  // dequantising kernels: per-column epilogue of the store, y = post_mul[f] * y + post_add[f]
    int post_relu;                     // (nullptr = none; then max(y, 0) when set) -- the sweep's fused store does the same
    const char *code;                  // code-stream kernels: the compiled schedule (executable memory) and the byte offset of
    const uint64_t *code_start;        // every (tile, wave) stream in it
    uint32_t piece_bytes;              // code-stream kernels: bytes of a chunk that one wave DMAs (chunk bytes / 16: the plan's ring geometry)
    uint32_t xcd_sx;                   // slices per XCD (round 5): 1 = an XCD streams ONE slice of X (X shared in its L2, a tile's code fetched by every
                                       // slice's XCDs); 2 / 4 = consecutive workgroups of an XCD are slices of the SAME tile and share its code in L2
    uint32_t half_split;               // (round 6) the plan folds two column ranges into the halves of a wave: the store adds the halves (plain 4-byte code-stream kernels)
    uint64_t *stamps;                  // measurement build (k_lds_code8_f32_ts): four 100 MHz clock values per wave (start, stream entered, stream left, stored)
    uint32_t xcd_contig;               // 0 = an XCD takes every xcd_group-th tile; T > 0 = a contiguous run of T tiles (plans whose neighbouring tiles stage the
                                       // same chunks -- the dense half of a density split: they then meet in that XCD's L2)
};
"""

KERNEL = """
// %(doc)s
__global__ __launch_bounds__(%(threads)d) void %(name)s(LdsArgs a) {
    extern __shared__ char lds_dyn[];
    constexpr uint32_t NW = %(NW)d, KA = %(KA)d, BATCH = %(BATCH)d, PIECE = %(piece)du;
    const uint32_t b = blockIdx.x;
    uint32_t slice, ti;
    if (a.xcd_group) {  // blocks b and b + 8 share an XCD: a group of xcd_group XCDs streams xcd_sx slices of X through its L2s; with
                        // xcd_sx > 1 workgroups i, i + 1 (.. i + 3) of an XCD are the slices of ONE tile: the same code stream, in step
        const uint32_t xcd = b & 7, i = b >> 3, sx = a.xcd_sx ? a.xcd_sx : 1u;
        slice = (xcd / a.xcd_group) * sx + i %% sx;
        ti = a.xcd_contig ? (xcd %% a.xcd_group) * a.xcd_contig + i / sx : (xcd %% a.xcd_group) + a.xcd_group * (i / sx);
    } else {
        slice = b %% a.nslices;
        ti = b / a.nslices;
    }
    if (ti >= a.ntiles) return;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const LdsTile *t = a.tiles + ti;
    const uint32_t nch = __builtin_amdgcn_readfirstlane(t->nch);
    const uint64_t tok = (uint64_t)(a.tok + (uint64_t)t->tokstart[wave] * BATCH);
    const uint32_t cid0 = __builtin_amdgcn_readfirstlane(t->chunk0);
    const uint64_t xs = (uint64_t)(a.xs + (uint64_t)slice * a.slice_stride + wave * %(piece_expr)s);
    const uint64_t rowmap = (uint64_t)(a.rowmap + ((uint64_t)ti * NW + wave) * ((KA + 7u) & ~7u));   // (LdsGeometry::ka_stride)
    const uint64_t cb = (uint64_t)(a.c + (uint64_t)slice * %(cslice)du);
    const uint32_t wvalid = __builtin_amdgcn_readfirstlane(min(%(fps)du, a.w - slice * %(fps)du));
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dyn + wave * %(piece_expr)s);
    const uint32_t scale = %(scale_expr)s;
%(post_decl)s#define PYGIM_SU(x) __builtin_amdgcn_readfirstlane((uint32_t)(x))
#define PYGIM_SU2(x) ((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x)))   /* (the builtin returns int: no sign extension into the high word) */
    const uint32_t tok_lo = PYGIM_SU(tok), tok_hi = PYGIM_SU(tok >> 32), xs_lo = PYGIM_SU(xs), xs_hi = PYGIM_SU(xs >> 32);
    const uint32_t rm_lo = PYGIM_SU(rowmap), rm_hi = PYGIM_SU(rowmap >> 32), c_lo = PYGIM_SU(cb), c_hi = PYGIM_SU(cb >> 32);
#undef PYGIM_SU
    const uint64_t tok_s = ((uint64_t)tok_hi << 32) | tok_lo, rm_s = ((uint64_t)rm_hi << 32) | rm_lo;
%(post_decl2)s
%(code_decl)s
    asm volatile(
%(asm)s
        :
        : [lane] "v"(lane), [tok] "s"(tok_s), [cid0] "s"(cid0), [nch] "s"(nch), [xs_lo] "s"(xs_lo),
          [xs_hi] "s"(xs_hi), [rowmap] "s"(rm_s), [c_lo] "s"(c_lo), [c_hi] "s"(c_hi), [ldc] "s"(a.ldc_bytes),
          [wvalid] "s"(wvalid), [accum] "s"(a.accumulate), [ldsw] "s"(ldsw), [wdelta] "s"(a.wdelta), [scale] "s"(scale), [wave] "s"(wave)%(post_ops)s%(code_ops)s
        : %(clobbers)s, "vcc", "scc", "memory");
#undef PYGIM_SU2
}
"""


ABLATE = "--ablate" in sys.argv   # also emit the timing-experiment variants of round 3 (wrong results; make -C pygim_amd/csrc ablate)


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pygim_amd", "csrc", "lds_kernel_gen.hpp")
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
    text = HEADER % dict(KA8=GEOS[8].KA, KA16=GEOS[16].KA, B8=GEOS[8].BATCH, B16=GEOS[16].BATCH, KAL=GEO_L16.KA, BL=GEO_L16.BATCH, KAC8=GEO_CODE8.KA)
    variants = []
    for nw in (8, 16):
        variants.append((f"k_lds_spmm_f32_w{nw}", "v_add_f32", nw, 0, "FLT32, unit weights: sums in stored order, bit-identical to the CPU loop"))
        variants.append((f"k_lds_spmm_i32_w{nw}", "v_add_u32", nw, 0, "INT32, unit weights: two's-complement modular sums"))
    for ab in ((6, 7, 10, 11, 12, 15, 16, 17, 18, 19) if ABLATE else ()):
        variants.append((f"k_lds_spmm_f32_w16_ab{ab}", "v_add_f32", 16, ab, f"TIMING EXPERIMENT ONLY (wrong results): ablation {ab}, see the generator"))
    variants.append(("k_lds_spmm_f32_w16_val", "v_add_f32", 16, 0, "FLT32 with values: acc += val * x, product and sum rounded separately, stored order", "v_mul_f32"))
    variants.append(("k_lds_spmm_i32_w16_val", "v_add_u32", 16, 0, "INT32 with values: modular", "v_mul_lo_u32"))
    # INT16: a lane holds TWO features (a slice is 128 features = the same 256 bytes), packed 16-bit adds wrap each half on its own
    variants.append(("k_lds_spmm_i16_w16", "v_pk_add_u16", 16, 0, "INT16, unit weights: 128 features per slice, packed modular sums"))
    variants.append(("k_lds_spmm_i16_w16_val", "v_pk_add_u16", 16, 0, "INT16 with values (the value in both halves of its dword)", "v_pk_mul_lo_u16"))
    variants.append(("k_lds_spmm_f32_w16_deq", "v_add_f32", 16, 0, "FLT32 quantised features, the store dequantises: out = sum * scale", None, "f32"))
    variants.append(("k_lds_spmm_i32_w16_deq", "v_add_u32", 16, 0, "INT32 quantised features, the store dequantises: out = float(sum) * scale", None, "i32"))
    for base, op, deq in (("f32", "v_add_f32", None), ("i32", "v_add_u32", None), ("f32", "v_add_f32", "f32"), ("i32", "v_add_u32", "i32"),
                          ("i16", "v_pk_add_u16", None)):
        variants.append((f"k_lds_spmm_{base}_w16b" + ("_deq" if deq else ""), op, "L16", 0,
                         "the same for plans with long slots: 16-token batches, 80 accumulators per wave", None, deq))
    # the code-stream form (the schedule compiled into machine code, lds_plan.hpp lds_code_from_plan): 16 waves, unit weights
    for base, op, deq in (("f32", "v_add_f32", None), ("i32", "v_add_u32", None), ("f32", "v_add_f32", "f32"), ("i32", "v_add_u32", "i32"),
                          ("i16", "v_pk_add_u16", None)):
        variants.append((f"k_lds_code_{base}" + ("_deq" if deq else ""), op, 16, 0,
                         "CODE-STREAM form: the slot loop is a straight-line instruction stream compiled from the schedule (1.5 instructions per stored entry)",
                         None, deq))
    for base, op, deq in (("f32", "v_add_f32", None), ("i32", "v_add_u32", None), ("f32", "v_add_f32", "f32"), ("i32", "v_add_u32", "i32"),
                          ("i16", "v_pk_add_u16", None)):
        variants.append((f"k_lds_code8_{base}" + ("_deq" if deq else ""), op, "C8", 0,
                         "CODE-STREAM form, 8 waves x 228 accumulators (2 waves per SIMD, 1 824-row tiles: fewer rounds of workgroups, less of X staged)",
                         None, deq))
    variants.append(("k_lds_code8_f32_ts", "v_add_f32", "C8", 0,
                     "MEASUREMENT build of k_lds_code8_f32 (tunable lds_stamp): the same product, plus four clock stamps per wave", None, None))
    variants.append(("k_lds_code8_f64", "v_add_f64", "C8W", 0,
                     "CODE-STREAM form, DBL64: 512-byte rows in LDS, one ds_read_b64 per staged column, a register pair per running sum (8 waves x 114 rows)", None, None, "f64"))
    variants.append(("k_lds_code8_i64", "v_add_co_u32", "C8W", 0,
                     "CODE-STREAM form, INT64: as DBL64, the add is v_add_co_u32 + v_addc_co_u32 (modular)", None, None, "i64"))
    variants.append(("k_lds_code8_i16_deq", "v_pk_add_u16", "C8", 0,
                     "CODE-STREAM form, INT16 quantised features, the store dequantises: out = float(int16 sum) * scale", None, "i16", "i16_deq"))
    variants.append(("k_lds_code8_i8", "v_pk_add_u16", "C8", 0,
                     "CODE-STREAM form, INT8: features widened to 16 bits in the staged copy, packed 16-bit sums, the store keeps each sum's low byte", None, None, "i8"))
    variants.append(("k_lds_code8_i8_deq", "v_pk_add_u16", "C8", 0,
                     "CODE-STREAM form, INT8 quantised features (the conv layers' own type), the store dequantises: out = float(int8 sum) * scale", None, "i8", "i8_deq"))
    for v in variants:
        name, op, nw, ab, doc = v[:5]
        op_mul = v[5] if len(v) > 5 else None
        deq = v[6] if len(v) > 6 else None
        if "_code" in name:
            g = GEO_CODE8_64 if nw == "C8W" else (GEO_CODE8 if nw == "C8" else GEO_CODE)
        else:
            g = GEO_L16 if nw == "L16" else (GEO_W16 if op_mul else GEOS[nw])
        clob = ", ".join([f'"v{i}"' for i in range(g.T0, min(g.vmax, 256))] + [f'"s{i}"' for i in range(g.TOK0, 100)])
        is_code = "_code" in name
        out_kind = v[7] if len(v) > 7 else None
        cslice, fps = {None: (256, 64), "i8": (128, 128), "i8_deq": (512, 128), "i16_deq": (512, 128), "f64": (512, 64), "i64": (512, 64)}[out_kind]
        ts = name.endswith("_ts")
        asm = "\n".join(f'        "{ln}\\n"' for ln in body(op, g, ab, op_mul, deq, code=is_code, out_kind=out_kind, stamps=ts))
        guard = "_ab" in name   # ablation builds (timing experiments, wrong results) only with -DPYGIM_LDS_ABLATE (make ablate)
        if guard:
            text += "\n#ifdef PYGIM_LDS_ABLATE"
        text += KERNEL % dict(name=name, doc=doc, asm=asm, clobbers=clob, threads=g.threads, NW=g.NW, KA=g.KA, BATCH=g.BATCH,
                              piece=g.pieces * 1024,
                              scale_expr=("__builtin_amdgcn_readfirstlane(__float_as_uint(__uint_as_float(*a.deq_amax) * 2.0f / (float)(1u << a.deq_log2)))"
                                          if deq else "0u"),
                              post_decl=((f"    const uint64_t pmul = a.post_mul ? (uint64_t)(a.post_mul + slice * {fps}u) : 0ull, "
                                          f"padd = a.post_mul ? (uint64_t)(a.post_add + slice * {fps}u) : 0ull;\n") if deq else ""),
                              cslice=cslice, fps=fps,
                              post_decl2=("    const uint64_t pmul_s = ((uint64_t)PYGIM_SU2(pmul >> 32) << 32) | PYGIM_SU2(pmul), "
                                          "padd_s = ((uint64_t)PYGIM_SU2(padd >> 32) << 32) | PYGIM_SU2(padd);" if deq else ""),
                              post_ops=(',\n          [pmul] "s"(pmul_s), [padd] "s"(padd_s), [relu] "s"(a.post_relu)' if deq else ""),
                              code_decl=("    const uint64_t code_a = (uint64_t)(a.code + a.code_start[(uint64_t)ti * NW + wave]);\n"
                                         "    const uint64_t code_s = ((uint64_t)((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(code_a >> 32))) << 32) | "
                                         "(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)code_a);" if is_code else "") +
                                        ("\n    const uint64_t stamps_a = (uint64_t)(a.stamps + ((uint64_t)b * NW + wave) * 4);\n"
                                         "    const uint64_t stamps_s = ((uint64_t)((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(stamps_a >> 32))) << 32) | "
                                         "(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)stamps_a);" if ts else ""),
                              code_ops=((',\n          [code] "s"(code_s), [hsplit] "s"(a.half_split)' if is_code else "") + (', [stamps] "s"(stamps_s)' if ts else "")),
                              piece_expr=("a.piece_bytes" if is_code else "PIECE"))
        if guard:
            text += "#endif  // PYGIM_LDS_ABLATE\n"
    text += "\n}  // namespace pygim\n"
    with open(out, "w") as f:
        f.write(text)
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    sys.exit(main())
