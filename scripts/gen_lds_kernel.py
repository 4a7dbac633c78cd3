#!/usr/bin/env python3
"""Writes pygim_amd/csrc/lds_kernel_gen.hpp: the LDS-staged product kernels (k_lds_spmm_*) for gfx950.

The hot loop is hand-written gfx950 assembly inside a HIP kernel (the C++ prologue only turns kernel
arguments into wave-uniform pointers).  Why assembly: a wave keeps KA rows' running sums in KA vector
registers (one 64-feature slice per register, one feature per lane) and picks the register of each stored entry
with the wave-uniform VGPR index (s_set_gpr_idx_*), which HIP C++ cannot express for 200 registers.

Per stored entry the wave issues
    v_bfi_b32   addr = (token & 0xff00) | (lane * 4 + buffer)      -- row of the X chunk in LDS
    ds_read_b32 x    = LDS[addr]                                   -- 64 lanes x 4 B, conflict-free
    s_set_gpr_idx_idx token                                        -- M0[7:0] = accumulator index
    v_add       acc[idx] += x
which is the reference's scratchpad loop (spmm_default/dpu_kernels/spmm_mul_csr_dpu.c:108-126) with the
accumulators of ~1 700 rows resident per compute unit.  Schedule format: lds_plan.hpp.

Register map (everything from v16 / s40 up is named in the clobber list, the compiler keeps below):
    v16..v31   x / address registers of a batch          s40..s55  token set A      s56..s71  token set B
    v32        lane*4 + current LDS buffer                 s72 batches left in the slot, s73 slot, s74 next chunk id
    v33        0xff00                                      s75 tmp, s76:77 DMA source, s78 DMA LDS base, s79 buffer select
    v34        lane*16 (DMA lane offset)                   s80:81 token pointer, s82:83 batch-count pointer,
    v36        lane*4 (store offset), v37/v38 tmp          s84:85 chunk-id pointer, s86 nch, s87 next slot's batches,
    v40..      accumulators, v[40+KA] = dummy              s88 chunk id after next, s89 k, s90..s97 row ids, s98:99 store address
"""
import os
import sys

KA = 208
BATCH = 16
NW = 8
ACC0 = 40
X0 = 16
TOKA, TOKB = 40, 56


def body(op_add, weighted=False):
    L = []
    a = L.append
    # ---- set-up
    a("v_lshlrev_b32 v36, 2, %[lane]")
    a("v_lshlrev_b32 v34, 4, %[lane]")
    a("v_mov_b32 v33, 0xff00")
    for i in range(KA + 1):
        a(f"v_mov_b32 v{ACC0 + i}, 0")
    a("s_mov_b64 s[80:81], %[tok]")
    a("s_mov_b64 s[82:83], %[nb]")
    a("s_mov_b64 s[84:85], %[chunks]")
    a("s_mov_b32 s86, %[nch]")
    a("s_mov_b32 s73, 0")
    a("s_mov_b32 s79, 0")
    a("s_cmp_eq_u32 s86, 0")
    a("s_cbranch_scc1 L_out_%=")
    a("s_load_dword s74, s[84:85], 0x0")        # chunk id of slot 0
    a("s_load_dword s88, s[84:85], 0x4")        # chunk id of slot 1
    a("s_load_dword s72, s[82:83], 0x0")        # batches of slot 0
    a(f"s_load_dwordx16 s[{TOKA}:{TOKA + 15}], s[80:81], 0x0")
    a("s_waitcnt lgkmcnt(0)")

    def dma(cid, bufsel_expr_reg):
        # 8 pieces of 1 KiB: this wave's eighth of the 64 KiB chunk `cid` -> LDS buffer `bufsel`
        a(f"s_lshl_b32 s75, {cid}, 16")
        a("s_add_u32 s76, %[xs_lo], s75")
        a("s_addc_u32 s77, %[xs_hi], 0")
        a(f"s_add_u32 s78, {bufsel_expr_reg}, %[ldsw]")
        for i in range(8):
            a("s_mov_b32 m0, s78")
            a("s_nop 0")
            a("global_load_lds_dwordx4 v34, s[76:77]")
            if i < 7:
                a("s_add_u32 s76, s76, 0x400")
                a("s_addc_u32 s77, s77, 0")
                a("s_add_u32 s78, s78, 0x400")

    dma("s74", "s79")
    a("s_waitcnt vmcnt(0)")
    a("s_barrier")
    a("s_mov_b32 s74, s88")                      # s74 = chunk id of the NEXT slot from here on
    a("s_mov_b32 s91, 0")                        # s91 = parity: which token set holds the current batch (0 = A)
    # ---- slot loop
    a("L_slot_%=:")
    a("s_add_u32 s75, s73, 1")
    a("s_cmp_lt_u32 s75, s86")
    a("s_cbranch_scc0 L_nodma_%=")
    a("s_xor_b32 s90, s79, 0x10000")
    dma("s74", "s90")
    a("L_nodma_%=:")
    a("s_load_dword s88, s[84:85], 0x8")        # chunk id of slot j + 2 (the list is padded by two)
    a("s_load_dword s87, s[82:83], 0x20")       # batches of slot j + 1 (closing row of zeros)
    a("v_add_u32 v32, s79, v36")
    a("s_cmp_eq_u32 s72, 0")
    a("s_cbranch_scc1 L_slotend_%=")
    a("s_cmp_eq_u32 s91, 0")
    a("s_cbranch_scc0 L_batchB_%=")

    def batch(P, Q, me, other):
        a(f"L_batch{me}_%=:")
        a("s_add_u32 s80, s80, 0x40")
        a("s_addc_u32 s81, s81, 0")
        a(f"s_load_dwordx16 s[{Q}:{Q + 15}], s[80:81], 0x0")   # the next batch (this slot's or the next one's)
        for i in range(BATCH):
            a(f"v_bfi_b32 v{X0 + i}, v33, s{P + i}, v32")
        for i in range(BATCH):
            a(f"ds_read_b32 v{X0 + i}, v{X0 + i}")
        a("s_waitcnt lgkmcnt(0)")
        a(f"s_set_gpr_idx_on s{P}, gpr_idx(SRC1,DST)")
        a(f"{op_add} v{ACC0}, v{X0}, v{ACC0}")
        for i in range(1, BATCH):
            a(f"s_set_gpr_idx_idx s{P + i}")
            a(f"{op_add} v{ACC0}, v{X0 + i}, v{ACC0}")
        a("s_set_gpr_idx_off")
        a("s_sub_u32 s72, s72, 1")
        a("s_cmp_eq_u32 s72, 0")
        a(f"s_cbranch_scc1 L_done{me}_%=")

    batch(TOKA, TOKB, "A", "B")
    batch(TOKB, TOKA, "B", "A")
    a("s_branch L_batchA_%=")
    a("L_doneA_%=:")
    a("s_mov_b32 s91, 1")                        # the prefetched batch sits in set B
    a("s_branch L_slotend_%=")
    a("L_doneB_%=:")
    a("s_mov_b32 s91, 0")
    a("L_slotend_%=:")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")           # my pieces of the next chunk have landed; s87 / s88 are in
    a("s_barrier")                               # ... and everybody is done reading the current one
    a("s_xor_b32 s79, s79, 0x10000")
    a("s_mov_b32 s72, s87")
    a("s_mov_b32 s74, s88")
    a("s_add_u32 s84, s84, 4")
    a("s_addc_u32 s85, s85, 0")
    a("s_add_u32 s82, s82, 0x20")
    a("s_addc_u32 s83, s83, 0")
    a("s_add_u32 s73, s73, 1")
    a("s_cmp_lt_u32 s73, s86")
    a("s_cbranch_scc1 L_slot_%=")
    # ---- results: acc[k] -> C[rowmap[k]] (lanes beyond the slice's width masked off)
    a("L_out_%=:")
    a("s_mov_b64 s[92:93], exec")
    a("v_cmp_gt_u32 vcc, %[wvalid], %[lane]")
    a("s_and_b64 exec, exec, vcc")
    a("s_mov_b64 s[82:83], %[rowmap]")
    a("s_mov_b32 s89, 0")
    a("L_orow_%=:")
    a("s_load_dword s90, s[82:83], 0x0")
    a("s_waitcnt lgkmcnt(0)")
    a("s_cmp_eq_u32 s90, -1")
    a("s_cbranch_scc1 L_oskip_%=")
    a("s_mul_i32 s98, s90, %[ldc]")
    a("s_mul_hi_u32 s99, s90, %[ldc]")
    a("s_add_u32 s98, s98, %[c_lo]")
    a("s_addc_u32 s99, s99, %[c_hi]")
    a("s_set_gpr_idx_on s89, gpr_idx(SRC0)")
    a(f"v_mov_b32 v37, v{ACC0}")
    a("s_set_gpr_idx_off")
    a("s_cmp_eq_u32 %[accum], 0")
    a("s_cbranch_scc1 L_ost_%=")
    a("global_load_dword v38, v36, s[98:99]")
    a("s_waitcnt vmcnt(0)")
    a(f"{op_add} v37, v38, v37")
    a("L_ost_%=:")
    a("global_store_dword v36, v37, s[98:99]")
    a("L_oskip_%=:")
    a("s_add_u32 s82, s82, 4")
    a("s_addc_u32 s83, s83, 0")
    a("s_add_u32 s89, s89, 1")
    a(f"s_cmp_lt_u32 s89, {KA}")
    a("s_cbranch_scc1 L_orow_%=")
    a("s_mov_b64 exec, s[92:93]")
    a("s_waitcnt vmcnt(0)")
    return L


HEADER = '''// GENERATED by scripts/gen_lds_kernel.py -- do not edit; edit the generator and re-run it.
// LDS-staged product kernels for gfx950: schedule format in lds_plan.hpp, design notes in the generator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lds_plan.hpp"

namespace pygim {

constexpr uint32_t LDS_KA = %(KA)d, LDS_NW = %(NW)d, LDS_BATCH = %(BATCH)d, LDS_KC = 256, LDS_BYTES = 131072;

struct LdsArgs {
    const uint32_t *tok;      // token streams
    const uint32_t *nb;       // batch counts
    const uint32_t *chunks;   // chunk-id lists
    const LdsTile *tiles;
    const uint32_t *rowmap;   // [ntiles][NW][KA] row of C (0xffffffff = none)
    const char *xs;           // slice-major X: [nslices][ncols_pad][64] elements of 4 bytes
    char *c;                  // row-major result
    uint64_t slice_stride;    // bytes between two slices of xs
    uint32_t ldc_bytes, w, nslices, ntiles, accumulate, xcd_group;
};
'''

KERNEL = '''
// %(doc)s
__global__ __launch_bounds__(512) void %(name)s(LdsArgs a) {
    extern __shared__ char lds_dyn[];
    const uint32_t b = blockIdx.x;
    uint32_t slice, ti;
    if (a.xcd_group) {  // blocks b and b + 8 share an XCD: an XCD (or a group of them) streams ONE slice of X through its L2
        const uint32_t xcd = b & 7, i = b >> 3;
        slice = xcd / a.xcd_group;
        ti = (xcd %% a.xcd_group) + a.xcd_group * i;
    } else {
        slice = b %% a.nslices;
        ti = b / a.nslices;
    }
    if (ti >= a.ntiles) return;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const LdsTile *t = a.tiles + ti;
    const uint32_t nch = __builtin_amdgcn_readfirstlane(t->nch);
    const uint64_t tok = (uint64_t)(a.tok + (uint64_t)t->tokstart[wave] * LDS_BATCH);
    const uint64_t nb = (uint64_t)(a.nb + t->nb_off + wave);
    const uint64_t chunks = (uint64_t)(a.chunks + t->chunk_off);
    const uint64_t xs = (uint64_t)(a.xs + (uint64_t)slice * a.slice_stride + wave * 8192u);
    const uint64_t rowmap = (uint64_t)(a.rowmap + ((uint64_t)ti * LDS_NW + wave) * LDS_KA);
    const uint64_t cb = (uint64_t)(a.c + (uint64_t)slice * 256u);
    const uint32_t wvalid = __builtin_amdgcn_readfirstlane(min(64u, a.w - slice * 64u));
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dyn + wave * 8192u);
#define PYGIM_SU(x) __builtin_amdgcn_readfirstlane((uint32_t)(x))
    const uint32_t tok_lo = PYGIM_SU(tok), tok_hi = PYGIM_SU(tok >> 32), nb_lo = PYGIM_SU(nb), nb_hi = PYGIM_SU(nb >> 32);
    const uint32_t ch_lo = PYGIM_SU(chunks), ch_hi = PYGIM_SU(chunks >> 32), xs_lo = PYGIM_SU(xs), xs_hi = PYGIM_SU(xs >> 32);
    const uint32_t rm_lo = PYGIM_SU(rowmap), rm_hi = PYGIM_SU(rowmap >> 32), c_lo = PYGIM_SU(cb), c_hi = PYGIM_SU(cb >> 32);
#undef PYGIM_SU
    const uint64_t tok_s = ((uint64_t)tok_hi << 32) | tok_lo, nb_s = ((uint64_t)nb_hi << 32) | nb_lo;
    const uint64_t ch_s = ((uint64_t)ch_hi << 32) | ch_lo, rm_s = ((uint64_t)rm_hi << 32) | rm_lo;
    asm volatile(
%(asm)s
        :
        : [lane] "v"(lane), [tok] "s"(tok_s), [nb] "s"(nb_s), [chunks] "s"(ch_s), [nch] "s"(nch), [xs_lo] "s"(xs_lo),
          [xs_hi] "s"(xs_hi), [rowmap] "s"(rm_s), [c_lo] "s"(c_lo), [c_hi] "s"(c_hi), [ldc] "s"(a.ldc_bytes),
          [wvalid] "s"(wvalid), [accum] "s"(a.accumulate), [ldsw] "s"(ldsw)
        : %(clobbers)s, "vcc", "scc", "memory");
}
'''


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pygim_amd", "csrc", "lds_kernel_gen.hpp")
    clob = ", ".join([f'"v{i}"' for i in range(16, 256)] + [f'"s{i}"' for i in range(40, 100)])
    text = HEADER % dict(KA=KA, NW=NW, BATCH=BATCH)
    for name, op, doc in (("k_lds_spmm_f32", "v_add_f32", "FLT32, unit weights: sums in stored order, bit-identical to the CPU loop"),
                          ("k_lds_spmm_i32", "v_add_u32", "INT32, unit weights: two's-complement modular sums")):
        asm = "\n".join(f'        "{ln}\\n"' for ln in body(op))
        text += KERNEL % dict(name=name, doc=doc, asm=asm, clobbers=clob)
    text += "\n}  // namespace pygim\n"
    with open(out, "w") as f:
        f.write(text)
    print("wrote", os.path.normpath(out))


if __name__ == "__main__":
    sys.exit(main())
