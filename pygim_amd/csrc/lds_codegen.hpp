// lds_codegen.hpp -- the code-stream encoder as a DATA-PARALLEL pipeline (round 5).
//
// lds_plan.hpp's lds_plan_build + lds_code_from_plan walk the graph on the HOST: column ids device -> host (458 MB for the Reddit-shaped
// graph), a token schedule (0.5 GB), two passes of instruction emission (1 GB) and the upload -- 1.2 s per group, 49-75 s when eight ranks
// share a box.  The reference's one-time step is a partition walk and a copy (spmm_default/spmm_mul_csr.c:118-330).  This file produces THE
// SAME BYTES -- the same (tile, wave) instruction streams at the same offsets -- from steps that are each one index-parallel body, a prefix
// sum or a stable sort, so that they run on the device against the resident CSR and write straight into executable memory:
//
//   host, from the row pointers alone (small): rows dealt to tiles / waves / accumulators (the row map), tiles heaviest first
//   D0  mark        per entry: chunk (col / KC) is present in its tile                                 -> per-tile chunk lists (host, tiny)
//   D1  keys        per entry: (stream = tile * NW + wave) : col : accumulator k                        (+ the value's bits, valued FLT32)
//   D2  sort        stable LSD radix sort of the keys: a stream's entries by staged column, then accumulator, then stored order --
//                   exactly the order the host encoder's per-slot stable sort produces
//   D3  col flags   per entry: first entry of its staged column (entries of one wave that hit the same column share one LDS read)
//   D4  scan        -> index of every staged column; D5 per column: first entry, LDS row
//   D6  slots       per (stream, slot): first staged column (binary search in the sorted keys); D7 groups per slot, scan -> group ids
//   D8  groups      per slot: its groups of G staged columns: LDS instructions (neighbours in one 256-row block pair up), entries
//   D9  stream pass per stream, sequential over its GROUPS (not entries): where every group's reads and adds go, and everything that is
//                   not a read or an add (chunk DMA with the chunk ids as literals, s_waitcnt with counts known here, barriers, the touches
//                   of the stream's own lines) -- first for the sizes, then (offsets known) writing
//   D10 emit        per group: its LDS reads; per entry: its v_add (and v_mul)
//
// Every body is PYGIM_HD (host + device): tests/native/lds_codegen_main.cpp runs the pipeline on the CPU (std::stable_sort, plain prefix
// sums) and compares the blob with lds_code_from_plan's byte for byte; csrc/lds_codegen_dev.hpp runs the same bodies as HIP kernels.
// Supported: the boundary hand-off (the default), one column range per row tile (col_splits = 1), no timing experiments -- anything else
// takes the host encoder.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <stdexcept>
#include <vector>

#include "lds_plan.hpp"

#if defined(__HIPCC__)
#define PYGIM_HD __host__ __device__
#else
#define PYGIM_HD
#endif

namespace pygim {

struct CgParams {
    // geometry
    uint32_t NW = 8, KA = 228, KC = 128, NBUF = 5, RB = 256, RPB = 256, G = 10, NS = 2, XW = 1;
    uint32_t wide = 0, valued = 0;
    uint32_t mulw = 0;                              // valued: dwords of an entry's multiply -- 2 (v_mul_f32 literal; v_mul_lo_u32 inline value), 4 (s_mov_b32 + v_mul_lo_u32)
    uint32_t int_inline = 0;                        // valued INT32 / INT64: every value lies in [-16, 64] (lds_int_values_inline)
    uint32_t i64_full = 0;                          // valued INT64: some value needs more than 32 bits (both halves through s[94:95])
    uint32_t opcode_add = 0x02000000u, addw = 1;   // dwords per accumulate
    uint32_t pieces = 4, chunk_bytes = 32768;
    uint32_t nrows = 0, ncols = 0, nchunks = 0, ntiles = 0, nstreams = 0;
    uint32_t S = 1, row_tiles = 0;                  // column ranges per row tile (col_splits); ntiles = row_tiles * S
    uint32_t col_bits = 1;                          // key = stream << (col_bits + 8) | col << 8 | k
    uint32_t half_H = 0;                            // half-split plans (LdsGeometry::half_split): column c of the matrix is the VIRTUAL column c mod H, staged beside its twin in one
                                                    // 256-byte row; ncols (above) = H; an entry's half (c >= H) rides in the sorted payload and picks the EXEC mask of its add
    // register map (LdsCodeRegs)
    uint32_t x0 = 6, acc0 = 28, vbase0 = 1, vbase1 = 2, vbase2 = 3, vl16 = 4, vtouch = 4, vjunk = 5;
    uint32_t s_xs = 80, s_ldsw = 82, s_cb = 84, s_ret = 86, s_pa = 92;
    PYGIM_HD uint32_t vbase(uint32_t blk) const { return blk == 0 ? vbase0 : blk == 1 ? vbase1 : vbase2; }
};

constexpr uint32_t CG_TOUCH_EVERY_DW = 256;   // (lds_code_from_plan: TOUCH_EVERY_DW)

inline CgParams cg_params(const LdsGeometry &geo, uint32_t opcode_add, bool valued, uint32_t nrows, uint32_t ncols, uint32_t gsize = 0, uint32_t nsets = 0,
                          bool int_inline = false, bool i64_full = false, uint32_t half_H = 0) {
    if (geo.NBUF >= 3 && !geo.boundary) throw std::runtime_error("lds codegen: the mid-slot hand-off takes the host encoder");
    CgParams P;
    const bool wide = geo.row_bytes == 512;
    if (geo.row_bytes != 256 && geo.row_bytes != 512) throw std::runtime_error("lds codegen: rows of 256 or 512 bytes");
    if (wide != (opcode_add == LDS_CODE_ADD_F64 || opcode_add == LDS_CODE_ADD_U64)) throw std::runtime_error("lds codegen: 512-byte rows are the 8-byte element types'");
    const LdsCodeRegs R = lds_code_regs(geo.NW, gsize, nsets, wide);
    P.NW = geo.NW; P.KA = geo.KA; P.KC = geo.KC; P.NBUF = geo.NBUF; P.RB = geo.row_bytes; P.RPB = 65536 / geo.row_bytes;
    P.G = R.gsize; P.NS = R.nsets; P.XW = wide ? 2 : 1; P.wide = wide; P.valued = valued;
    if (R.x0 + R.nx() > R.acc0 || R.acc0 + P.KA * P.XW > 256 || P.G > 12 || (P.NS - 1) * P.G > 15 || P.NS > 8)
        throw std::runtime_error("lds codegen: the geometry does not fit the register map");
    if (P.KA > 255) throw std::runtime_error("lds codegen: more than 255 accumulators per wave");
    P.opcode_add = opcode_add;
    P.i64_full = (valued && opcode_add == LDS_CODE_ADD_U64 && i64_full) ? 1 : 0;
    P.int_inline = (valued && (opcode_add == 0x68000000u || opcode_add == LDS_CODE_ADD_U64 || opcode_add == LDS_CODE_PK_ADD_U16) && int_inline && !P.i64_full) ? 1 : 0;
    P.mulw = !valued ? 0 : opcode_add == LDS_CODE_ADD_F64 ? 6 : opcode_add == LDS_CODE_ADD_U64 ? (P.i64_full ? 12 : P.int_inline ? 6 : 8) : ((opcode_add == 0x68000000u || opcode_add == LDS_CODE_PK_ADD_U16) && !int_inline) ? 4 : 2;
    P.addw = (opcode_add == LDS_CODE_PK_ADD_U16 || opcode_add == LDS_CODE_ADD_F64 || opcode_add == LDS_CODE_ADD_U64) ? 2 : 1;
    if (valued && opcode_add == LDS_CODE_ADD_U64) P.addw = 0;   // (the 64-bit sum is the v_mad_u64_u32 of the multiply)
    P.pieces = (geo.KC * geo.row_bytes / 1024) / geo.NW;
    if (P.pieces * geo.NW * 1024 != geo.KC * geo.row_bytes) throw std::runtime_error("lds codegen: a chunk is not a whole number of pieces per wave");
    P.chunk_bytes = geo.KC * geo.row_bytes;
    if ((geo.KC * geo.NBUF + P.RPB - 1) / P.RPB > 3) throw std::runtime_error("lds codegen: the ring is larger than three 64 KiB blocks");
    if (geo.NBUF > 12) throw std::runtime_error("lds codegen: more than 12 ring buffers");
    if ((geo.half_split != 0) != (half_H != 0)) throw std::runtime_error("lds codegen: a half-split plan names its H (and only such a plan does)");
    if (half_H && (valued || wide || half_H % geo.KC || ncols != half_H)) throw std::runtime_error("lds codegen: half-split plans are unit-weight plans of 4-byte types over H = whole chunks of virtual columns");
    P.half_H = half_H;
    P.nrows = nrows; P.ncols = ncols;
    P.nchunks = (ncols + geo.KC - 1) / geo.KC;
    const uint32_t R_rows = geo.rows_per_tile ? std::min(geo.rows_per_tile, geo.NW * geo.KA) : geo.NW * geo.KA;
    P.S = std::max(1u, geo.col_splits);
    P.row_tiles = (nrows + R_rows - 1) / R_rows;
    P.ntiles = P.row_tiles * P.S;
    P.nstreams = P.ntiles * geo.NW;
    if ((uint64_t)nrows * P.S >= (1ull << 32)) throw std::runtime_error("lds codegen: rows x column ranges beyond 32 bits");
    P.col_bits = 1;
    while (P.col_bits < 32 && (1ull << P.col_bits) < (uint64_t)ncols) P.col_bits++;
    uint32_t sb = 1;
    while ((1ull << sb) < (uint64_t)P.nstreams) sb++;
    if (sb + P.col_bits + 8 > 64) throw std::runtime_error("lds codegen: key wider than 64 bits");
    P.x0 = R.x0; P.acc0 = R.acc0; P.vbase0 = R.vbase[0]; P.vbase1 = R.vbase[1]; P.vbase2 = R.vbase[2];
    P.vl16 = R.vl16; P.vtouch = R.vtouch; P.vjunk = R.vjunk;
    P.s_xs = R.s_xs; P.s_ldsw = R.s_ldsw; P.s_cb = R.s_cb; P.s_ret = R.s_ret; P.s_pa = R.s_pa;
    return P;
}
inline uint32_t cg_key_bits(const CgParams &P) {
    uint32_t sb = 1;
    while ((1ull << sb) < (uint64_t)P.nstreams) sb++;
    return sb + P.col_bits + 8;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// host, from the row pointers alone: rows dealt to tiles / waves / accumulators exactly as lds_plan_build does (longest rows first,
// serpentine over the waves; tiles heaviest first), for one column range per row tile
// ------------------------------------------------------------------------------------------------------------------------------------
struct CgRows {
    std::vector<uint32_t> rowinfo;   // per row: (row tile * NW + wave) << 8 | k
    std::vector<uint32_t> rowmap_rt; // [row_tiles][NW][ka_stride]: the row at (row tile, wave, accumulator) (0xFFFFFFFF = none)
    std::vector<uint32_t> tnnz;      // [row_tiles * S]: stored entries of (row tile, column range), unsorted tile order
    std::vector<uint32_t> tile_pos;  // [row_tiles * S]: position of (row tile, column range) in the launch order (heaviest first)
    std::vector<uint32_t> rowmap;    // [ntiles][NW][ka_stride], launch order -- what the kernel's store stage reads (row + range * nrows)
    std::vector<uint32_t> tile_row0, tile_nnz;   // launch order
};
// phase a (row pointers alone): every row's tile, wave and accumulator.  One column range per tile: the tiles' entry counts as well
inline void cg_deal_rows_a(const uint32_t *rowptr, const LdsGeometry &geo, const CgParams &P, CgRows &out, const uint32_t *rorder = nullptr) {
    auto rid = [&](uint32_t pos) { return rorder ? rorder[pos] : pos; };   // (lds_plan_build: the row at a position of the tile order)
    auto rlen = [&](uint32_t pos) { const uint32_t r = rid(pos); return rowptr[r + 1] - rowptr[r]; };
    const uint32_t NW = geo.NW, KA = geo.KA, KAS = geo.ka_stride(), RS = NW * KAS;
    const uint32_t R = geo.rows_per_tile ? std::min(geo.rows_per_tile, NW * KA) : NW * KA;
    const uint32_t nrows = P.nrows, nrt = P.row_tiles;
    out.rowinfo.assign(nrows, 0);
    out.rowmap_rt.assign((size_t)nrt * RS, 0xFFFFFFFFu);
    out.tnnz.assign((size_t)nrt * P.S, 0);
    // (the row tiles are independent: host threads -- 2.4 M rows of a products-shaped part took 177 ms on one)
    lds_parallel_for(nrt, 0, [&](uint32_t rt) {
        std::vector<uint32_t> order;
        const uint32_t r0 = rt * R, r1 = std::min(nrows, r0 + R), nr = r1 - r0;
        if (P.S == 1) {
            if (!rorder) out.tnnz[rt] = rowptr[r1] - rowptr[r0];
            else
                for (uint32_t q = r0; q < r1; q++) out.tnnz[rt] += rlen(q);
        }
        order.resize(nr);
        for (uint32_t i = 0; i < nr; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return rlen(r0 + a) > rlen(r0 + b); });
        for (uint32_t i = 0; i < nr; i++) {
            const uint32_t round = i / NW, p = i % NW;
            const uint32_t w = (round & 1) ? NW - 1 - p : p;
            out.rowinfo[rid(r0 + order[i])] = ((rt * NW + w) << 8) | round;
            out.rowmap_rt[((size_t)rt * NW + w) * KAS + round] = rid(r0 + order[i]);
        }
    });
}
// phase b (the tiles' entry counts known -- S > 1: counted on the device): launch order (heaviest first, stable), row map, tile table
inline void cg_deal_rows_b(const LdsGeometry &geo, const CgParams &P, CgRows &out) {
    const uint32_t NW = geo.NW, KAS = geo.ka_stride(), RS = NW * KAS;
    const uint32_t R = geo.rows_per_tile ? std::min(geo.rows_per_tile, NW * geo.KA) : NW * geo.KA;
    const uint32_t ntiles = P.ntiles;
    std::vector<uint32_t> ord(ntiles);
    for (uint32_t t = 0; t < ntiles; t++) ord[t] = t;
    if (!geo.keep_tile_order) std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return out.tnnz[a] > out.tnnz[b]; });
    out.tile_pos.assign(ntiles, 0);
    for (uint32_t i = 0; i < ntiles; i++) out.tile_pos[ord[i]] = i;
    out.rowmap.assign((size_t)ntiles * RS, 0xFFFFFFFFu);
    out.tile_row0.assign(ntiles, 0);
    out.tile_nnz.assign(ntiles, 0);
    for (uint32_t t = 0; t < ntiles; t++) {
        const uint32_t rt = t / P.S, cs = t % P.S, ti = out.tile_pos[t];
        out.tile_row0[ti] = rt * R;
        out.tile_nnz[ti] = out.tnnz[t];
        for (uint32_t q = 0; q < RS; q++) {
            const uint32_t r = out.rowmap_rt[(size_t)rt * RS + q];
            if (r != 0xFFFFFFFFu) out.rowmap[(size_t)ti * RS + q] = r + cs * P.nrows;
        }
    }
}

// per-tile chunk lists from the presence flags (D0): chunks[choff[ti] .. + nch[ti]) ascending
struct CgChunks {
    std::vector<uint32_t> nch, choff, chunks;   // choff has ntiles + 1 entries
    uint64_t slots = 0;
};
inline void cg_chunk_lists(const uint8_t *flags, const CgParams &P, CgChunks &out) {
    out.nch.assign(P.ntiles, 0);
    out.choff.assign(P.ntiles + 1, 0);
    out.chunks.clear();
    for (uint32_t ti = 0; ti < P.ntiles; ti++) {
        out.choff[ti] = (uint32_t)out.chunks.size();
        for (uint32_t c = 0; c < P.nchunks; c++)
            if (flags[(size_t)ti * P.nchunks + c]) out.chunks.push_back(c);
        out.nch[ti] = (uint32_t)out.chunks.size() - out.choff[ti];
    }
    out.choff[P.ntiles] = (uint32_t)out.chunks.size();
    out.slots = out.chunks.size();
    out.chunks.push_back(0);   // (never empty)
}

// ------------------------------------------------------------------------------------------------------------------------------------
// device-side tables (plain pointers; the CPU emulation points them at vectors)
// ------------------------------------------------------------------------------------------------------------------------------------
struct CgTables {
    // inputs
    const uint32_t *rowptr = nullptr, *colind = nullptr, *vals_in = nullptr;   // CSR (vals: raw bits of 4-byte values, valued only)
    const uint64_t *vals_in64 = nullptr;                                        // valued DBL64: the values; the sorted 4-byte payload is then the ENTRY INDEX
    const uint32_t *rowinfo = nullptr;                                          // per row
    const uint32_t *tile_pos = nullptr;                                         // [row_tiles * S] -> launch position of the tile
    uint32_t *tnnz = nullptr;                                                   // [row_tiles * S] stored entries (counted when S > 1)
    const uint32_t *nch = nullptr, *choff = nullptr, *chunks = nullptr;         // per tile
    uint8_t *flags = nullptr;                                                   // [ntiles][nchunks]
    // per entry
    uint64_t *keys = nullptr;
    uint32_t *vals = nullptr;          // sorted with the keys
    uint32_t *colflag = nullptr;       // 1 = first entry of its staged column
    uint32_t *colx = nullptr;          // exclusive scan of colflag
    // per staged column
    uint32_t *col_first = nullptr;     // first entry (+ sentinel)
    uint16_t *col_lrow = nullptr;      // LDS row (column inside the chunk + buffer * KC)
    uint32_t *col_g = nullptr;         // group << 4 | position inside the group
    // per (stream, slot)  [nstreams slots in stream-major order, + sentinel]
    uint32_t *slot_firstcol = nullptr, *slot_ng = nullptr, *slot_firstgroup = nullptr;
    // per group
    uint32_t *g_nent = nullptr, *g_first = nullptr, *g_firstcol = nullptr;   // entries, first entry, first column
    uint32_t *g_n0 = nullptr;                                                // half-split plans: the group's entries of the LOWER column range (their adds come first, under the lower half of EXEC)
    uint8_t *g_nlds = nullptr, *g_ncols = nullptr, *g_xset = nullptr;
    uint32_t *g_rpos = nullptr, *g_apos = nullptr;                           // dword offsets inside the code blob (absolute)
    // per stream
    uint32_t *stream_dw = nullptr;     // size of the stream in dwords (multiple of 64)
    const uint64_t *start = nullptr;   // byte offset of the stream in the blob
    uint32_t *code = nullptr;
    unsigned long long *stats = nullptr;   // [0] entries read in pairs
    uint64_t nnz = 0;
    uint32_t ncols_total = 0, ngroups = 0, nsj = 0;
};

PYGIM_HD inline uint32_t cg_sbase(const CgParams &P, const CgTables &T, uint32_t s) {   // first (stream, slot) index of stream s
    const uint32_t ti = s / P.NW, w = s % P.NW;
    return P.NW * T.choff[ti] + w * T.nch[ti];
}
PYGIM_HD inline uint32_t cg_slot_of(const CgTables &T, uint32_t ti, uint32_t ch) {   // position of chunk ch in tile ti's list (it is there)
    const uint32_t *a = T.chunks + T.choff[ti];
    uint32_t lo = 0, hi = T.nch[ti];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < ch) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
PYGIM_HD inline uint32_t cg_key_stream(const CgParams &P, uint64_t key) { return (uint32_t)(key >> (P.col_bits + 8)); }
PYGIM_HD inline uint32_t cg_key_col(const CgParams &P, uint64_t key) { return (uint32_t)((key >> 8) & ((1ull << P.col_bits) - 1)); }

// column range of a chunk: range c holds chunks [nchunks * c / S, nchunks * (c + 1) / S)   (lds_plan_build: ch_lo, ch_hi)
PYGIM_HD inline uint32_t cg_range_of(const CgParams &P, uint32_t ch) {
    if (P.S == 1) return 0;
    uint32_t c = (uint32_t)((uint64_t)ch * P.S / P.nchunks);
    while (c > 0 && ch < (uint32_t)((uint64_t)P.nchunks * c / P.S)) c--;
    while (c + 1 < P.S && ch >= (uint32_t)((uint64_t)P.nchunks * (c + 1) / P.S)) c++;
    return c;
}
// the stream of an entry: its row's (row tile, wave), its column's range -> the tile's launch position
PYGIM_HD inline uint32_t cg_stream_of(const CgParams &P, const CgTables &T, uint32_t ri, uint32_t col) {
    const uint32_t rw = ri >> 8, rt = rw / P.NW, w = rw % P.NW;
    return T.tile_pos[rt * P.S + cg_range_of(P, col / P.KC)] * P.NW + w;
}
// (stream, slot) of a slot index (the device runs a thread per index)
PYGIM_HD inline void cg_sj_decode(const CgParams &P, const CgTables &T, uint32_t sj, uint32_t *s, uint32_t *j) {
    uint32_t lo = 0, hi = P.ntiles;   // last tile whose first slot index is <= sj (tiles without chunks own no index)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (P.NW * T.choff[mid] <= sj) lo = mid;
        else hi = mid;
    }
    // (tiles with nch == 0 share their choff with the next: step to the one that owns sj)
    while (lo + 1 < P.ntiles && T.nch[lo] == 0) lo++;
    const uint32_t rel = sj - P.NW * T.choff[lo], n = T.nch[lo];
    *s = lo * P.NW + rel / n;
    *j = rel % n;
}
// half-split plans: the virtual column of a stored column, and which half of the staged row holds it
PYGIM_HD inline uint32_t cg_vcol(const CgParams &P, uint32_t c) { return (P.half_H && c >= P.half_H) ? c - P.half_H : c; }
PYGIM_HD inline uint32_t cg_half(const CgParams &P, uint32_t c) { return (P.half_H && c >= P.half_H) ? 1u : 0u; }
// D0 / D1: per row (the device runs a wave per row, lanes over its entries)
PYGIM_HD inline uint32_t cg_tile_of_entry(const CgParams &P, const CgTables &T, uint32_t row, uint32_t e) {   // (row tile, column range), unsorted index
    return ((T.rowinfo[row] >> 8) / P.NW) * P.S + cg_range_of(P, cg_vcol(P, T.colind[e]) / P.KC);
}
PYGIM_HD inline void cg_mark_entry(const CgParams &P, const CgTables &T, uint32_t row, uint32_t e) {
    const uint32_t c = cg_vcol(P, T.colind[e]);
    const uint32_t ti = cg_stream_of(P, T, T.rowinfo[row], c) / P.NW;
    T.flags[(size_t)ti * P.nchunks + c / P.KC] = 1;
}
PYGIM_HD inline void cg_key_entry(const CgParams &P, const CgTables &T, uint32_t row, uint32_t e) {
    const uint32_t ri = T.rowinfo[row], c = cg_vcol(P, T.colind[e]);
    T.keys[e] = ((uint64_t)cg_stream_of(P, T, ri, c) << (P.col_bits + 8)) | ((uint64_t)c << 8) | (ri & 255u);
    if (P.valued) T.vals[e] = P.wide ? e : T.vals_in[e];   // (8-byte values: the entry's index rides with the key)
    else if (P.half_H) T.vals[e] = cg_half(P, T.colind[e]);   // (the stable sort keeps a row's lower-range entry of a virtual column ahead of its upper-range one: CSR order)
}
// D3: per sorted entry
PYGIM_HD inline void cg_colflag(const CgParams &P, const CgTables &T, uint64_t i) {
    uint32_t f = 1;
    if (i > 0 && !P.valued) {
        const uint64_t a = T.keys[i - 1], b = T.keys[i];
        f = (a >> 8) != (b >> 8);   // another stream, or another column (another chunk is another column)
    }
    T.colflag[i] = f;
}
// D5: per sorted entry that starts a staged column
PYGIM_HD inline void cg_col_fill(const CgParams &P, const CgTables &T, uint64_t i) {
    if (!T.colflag[i]) return;
    const uint32_t c = T.colx[i];
    const uint64_t key = T.keys[i];
    const uint32_t s = cg_key_stream(P, key), col = cg_key_col(P, key), ch = col / P.KC;
    const uint32_t j = cg_slot_of(T, s / P.NW, ch);
    T.col_first[c] = (uint32_t)i;
    T.col_lrow[c] = (uint16_t)(col % P.KC + (j % P.NBUF) * P.KC);
}
// D6: per (stream, slot) index sj in [0, nsj]: first staged column of the slot (== that of the next slot when the wave has no entry in it)
PYGIM_HD inline void cg_slot_bounds(const CgParams &P, const CgTables &T, uint32_t sj, uint32_t s, uint32_t j) {
    if (sj >= T.nsj) {
        T.slot_firstcol[sj] = T.ncols_total;
        return;
    }
    const uint32_t ch = T.chunks[T.choff[s / P.NW] + j];
    const uint64_t target = ((uint64_t)s << (P.col_bits + 8)) | ((uint64_t)ch * P.KC << 8);
    uint64_t lo = 0, hi = T.nnz;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (T.keys[mid] < target) lo = mid + 1;
        else hi = mid;
    }
    T.slot_firstcol[sj] = lo < T.nnz ? T.colx[lo] : T.ncols_total;
}
// D7: groups of a slot
PYGIM_HD inline void cg_slot_ngroups(const CgParams &P, const CgTables &T, uint32_t sj) {
    const uint32_t m = T.slot_firstcol[sj + 1] - T.slot_firstcol[sj];
    T.slot_ng[sj] = (m + P.G - 1) / P.G;
}
// D8: per slot: its groups (G staged columns each), the LDS instructions they take, the entries they serve
PYGIM_HD inline unsigned long long cg_slot_groups(const CgParams &P, const CgTables &T, uint32_t sj) {
    const uint32_t c0 = T.slot_firstcol[sj], c1 = T.slot_firstcol[sj + 1];
    uint32_t g = T.slot_firstgroup[sj];
    unsigned long long pairs = 0;
    for (uint32_t c = c0; c < c1; c += P.G, g++) {
        const uint32_t ce = c + P.G < c1 ? c + P.G : c1;
        uint32_t nlds = 0;
        for (uint32_t u = 0; c + u < ce; u++) {
            T.col_g[c + u] = (g << 4) | u;
            if (P.wide) { nlds++; continue; }
            if ((u & 1) == 0) {
                // (the host encoder's rule: at an even position of the group, the next column of the SLOT pairs up when it sits in the
                // same 64 KiB block of LDS rows; the group size is even, so the partner is in the same group)
                const bool paired = c + u + 1 < c1 && (T.col_lrow[c + u + 1] / P.RPB) == (T.col_lrow[c + u] / P.RPB);
                if (paired) {
                    nlds++;
                    pairs += (unsigned long long)(T.col_first[c + u + 2] - T.col_first[c + u]);
                } else {
                    nlds += (c + u + 1 < ce) ? 2 : 1;
                }
            }
        }
        T.g_first[g] = T.col_first[c];
        T.g_nent[g] = T.col_first[ce] - T.col_first[c];
        if (P.half_H) {
            uint32_t n0 = 0;
            for (uint32_t i = T.col_first[c]; i < T.col_first[ce]; i++) n0 += T.vals[i] == 0u;
            T.g_n0[g] = n0;
        }
        T.g_firstcol[g] = c;
        T.g_nlds[g] = (uint8_t)nlds;
        T.g_ncols[g] = (uint8_t)(ce - c);
    }
    return pairs;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// D9: the stream pass -- lds_code_from_plan's emit_stream at GROUP granularity (boundary hand-off).  write == false: sizes only.
// ------------------------------------------------------------------------------------------------------------------------------------
struct CgEmit {
    uint32_t *out;       // nullptr: count only
    uint64_t n;          // dwords so far
    uint32_t since_touch;
    uint64_t vm_touch;
    PYGIM_HD void put(uint32_t a) {
        if (out) out[n] = a;
        n++;
    }
    PYGIM_HD void op(uint32_t a) { put(a); since_touch++; }
    PYGIM_HD void op(uint32_t a, uint32_t b) { put(a); put(b); since_touch += 2; }
    PYGIM_HD void skip(uint32_t dw) { n += dw; since_touch += dw; }   // words another step writes (reads, adds)
};

PYGIM_HD inline void cg_stream_pass(const CgParams &P, const CgTables &T, uint32_t s, bool write) {
    const uint32_t ti = s / P.NW;
    const uint32_t nch = T.nch[ti];
    const uint32_t *chunk_ids = T.chunks + T.choff[ti];
    const uint32_t sj0 = cg_sbase(P, T, s);
    CgEmit e;
    const uint64_t base_dw = write ? T.start[s] / 4 : 0;
    e.out = write ? T.code + base_dw : nullptr;
    e.n = 0;
    e.since_touch = 0;
    e.vm_touch = 0;
    auto s_add_lit = [&](uint32_t sdst, uint32_t ssrc, uint32_t lit) { e.op(0x80000000u | (sdst << 16) | (0xFFu << 8) | ssrc, lit); };
    auto s_addc0 = [&](uint32_t sdst, uint32_t ssrc) { e.op(0x82000000u | (sdst << 16) | (0x80u << 8) | ssrc); };
    auto touch = [&]() {
        s_add_lit(P.s_cb, P.s_cb, (e.since_touch + 5) * 4);
        s_addc0(P.s_cb + 1, P.s_cb + 1);
        e.op(0xDC508000u, (P.vjunk << 24) | (P.s_cb << 16) | P.vtouch);
        e.since_touch = 0;
        e.vm_touch++;
    };
    auto dma = [&](uint32_t cid, uint32_t buf) {
        s_add_lit(P.s_pa, P.s_xs, cid * P.chunk_bytes);
        s_addc0(P.s_pa + 1, P.s_xs + 1);
        for (uint32_t i = 0; i < P.pieces; i++) {
            if (i % 4 == 0) {
                if (i) {
                    s_add_lit(P.s_pa, P.s_pa, 0x1000);
                    s_addc0(P.s_pa + 1, P.s_pa + 1);
                }
                s_add_lit(124 /* m0 */, P.s_ldsw, buf * P.chunk_bytes + (i / 4) * 0x1000);
                e.op(0xBF800000u);
            }
            e.op(0xDDF48000u | ((i % 4) * 1024), (P.s_pa << 16) | P.vl16);
        }
    };
    uint64_t dma_issued = 0;
    uint64_t landed_mark[16];
    for (int q = 0; q < 16; q++) landed_mark[q] = 0;
    auto vm_now = [&]() { return dma_issued + e.vm_touch; };
    auto dma_chunk = [&](uint32_t j) {
        dma(chunk_ids[j], j % P.NBUF);
        dma_issued += P.pieces;
        landed_mark[j & 15] = vm_now();
    };
    auto wait_landed = [&](uint32_t j) {
        const uint64_t younger = vm_now() - landed_mark[j & 15];
        const uint32_t nn = (uint32_t)(younger < 63 ? younger : 63);
        e.op(0xBF8C0F70u | (nn & 15) | (((nn >> 4) & 3) << 14));
    };
    auto wait_lgkm = [&](uint32_t n) { e.op(0xBF8CC07Fu | ((n < 15u ? n : 15u) << 8)); };
    // groups whose reads are issued and whose adds are not, oldest first
    uint32_t pend_g[10], pend_nlds[10], npend = 0;
    uint32_t gcount = 0;
    auto consume_oldest = [&]() {
        uint32_t younger = 0;
        for (uint32_t q = 1; q < npend; q++) younger += pend_nlds[q];
        wait_lgkm(younger);
        const uint32_t g = pend_g[0];
        if (write) T.g_apos[g] = (uint32_t)(base_dw + e.n);
        e.skip(T.g_nent[g] * (P.mulw + P.addw));
        if (P.half_H) e.skip((T.g_n0[g] > 0 ? 1u : 0u) + (T.g_n0[g] < T.g_nent[g] ? 1u : 0u) + 1u);   // s_mov_b64 exec, <half> per range present + s_mov_b64 exec, -1
        for (uint32_t q = 1; q < npend; q++) {
            pend_g[q - 1] = pend_g[q];
            pend_nlds[q - 1] = pend_nlds[q];
        }
        npend--;
    };
    if (nch) {
        for (uint32_t j = 0; j + 1 < P.NBUF && j < nch; j++) dma_chunk(j);
        wait_landed(0);
        e.op(0xBF8A0000u);
    }
    for (uint32_t j = 0; j < nch; j++) {
        bool dma_due = j + P.NBUF - 1 < nch;
        const uint32_t g0 = T.slot_firstgroup[sj0 + j], g1 = T.slot_firstgroup[sj0 + j + 1];
        for (uint32_t g = g0; g < g1; g++) {
            const uint32_t nlds = T.g_nlds[g];
            if (write) {
                T.g_rpos[g] = (uint32_t)(base_dw + e.n);
                T.g_xset[g] = (uint8_t)(gcount % P.NS);
            }
            e.skip(2 * nlds);
            pend_g[npend] = g;
            pend_nlds[npend] = nlds;
            npend++;
            gcount++;
            if (dma_due) {
                dma_chunk(j + P.NBUF - 1);
                dma_due = false;
            }
            if (npend >= P.NS) consume_oldest();
            if (e.since_touch >= CG_TOUCH_EVERY_DW) touch();
        }
        if (dma_due) dma_chunk(j + P.NBUF - 1);
        while (npend > 1) consume_oldest();
        wait_lgkm(0);
        if (j + 1 < nch) wait_landed(j + 1);
        e.op(0xBF8A0000u);
    }
    while (npend) consume_oldest();
    e.op(0xBF8C0F70u);
    e.op(0xBE801D00u | P.s_ret);
    while (e.n % 64) e.put(0xBF800000u);
    if (!write) T.stream_dw[s] = (uint32_t)e.n;
}

// D10: per group: its LDS reads
PYGIM_HD inline void cg_emit_reads(const CgParams &P, const CgTables &T, uint32_t g) {
    if (P.half_H) {   // the EXEC masks around the group's adds (lds_code_from_plan: s[76:77] = lanes 0..31, s[78:79] = lanes 32..63, set by the kernel)
        uint32_t *a = T.code + T.g_apos[g];
        const uint32_t n0 = T.g_n0[g], n1 = T.g_nent[g] - n0;
        uint32_t at = 0;
        if (n0) { a[at] = 0xBEFE0100u | 76u; at += 1 + n0 * P.addw; }
        if (n1) { a[at] = 0xBEFE0100u | 78u; at += 1 + n1 * P.addw; }
        a[at] = 0xBEFE01C1u;
    }
    const uint32_t c0 = T.g_firstcol[g], nc = T.g_ncols[g];
    const uint32_t xb = P.x0 + P.G * P.XW * T.g_xset[g];
    uint32_t *w = T.code + T.g_rpos[g];
    // (the partner of a column at an even position may lie beyond this group's last column only when it is the slot's last: then
    // there is none -- g_ncols says so)
    for (uint32_t u = 0; u < nc;) {
        const uint32_t r0 = T.col_lrow[c0 + u], blk = r0 / P.RPB;
        if (P.wide) {
            *w++ = 0xD8EC0000u | ((r0 % P.RPB) * P.RB);
            *w++ = ((xb + u * 2) << 24) | P.vbase(blk);
            u++;
        } else if ((u & 1) == 0 && u + 1 < nc && (T.col_lrow[c0 + u + 1] / P.RPB) == blk) {
            const uint32_t r1 = T.col_lrow[c0 + u + 1];
            *w++ = 0xD8700000u | ((r1 & 255) << 8) | (r0 & 255);
            *w++ = ((xb + u) << 24) | P.vbase(blk);
            u += 2;
        } else {
            *w++ = 0xD86C0000u | ((r0 & 255) << 8);
            *w++ = ((xb + u) << 24) | P.vbase(blk);
            u++;
        }
    }
}
// D10: per sorted entry: its accumulate (and, valued, its multiply: all of a group's products come before its sums)
PYGIM_HD inline void cg_emit_entry(const CgParams &P, const CgTables &T, uint64_t i) {
    const uint32_t c = T.colx[i] + T.colflag[i] - 1;
    const uint32_t cg = T.col_g[c], g = cg >> 4, u = cg & 15;
    const uint32_t q = (uint32_t)i - T.g_first[g];
    const uint32_t k = (uint32_t)T.keys[i] & 255u;
    const uint32_t vx = P.x0 + P.G * P.XW * T.g_xset[g] + u * P.XW, vk = P.acc0 + k * P.XW;
    uint32_t *w = T.code + T.g_apos[g];
    if (P.valued) {
        if (P.opcode_add == LDS_CODE_ADD_U64) {   // INT64 (values that fit int32): lds_code_from_plan's sequence of 32-bit pieces
            const uint64_t v64 = T.vals_in64[T.vals[i]];
            const int32_t v = (int32_t)(uint32_t)v64;
            const uint32_t xl = vx, xh = vx + 1, ah = vk + 1, src = P.int_inline ? lds_inline_int(v) : LDS_CODE_S_VAL;
            uint32_t *m = w + P.mulw * q;
            if (!P.int_inline) { *m++ = 0xBE8000FFu | (LDS_CODE_S_VAL << 16); *m++ = (uint32_t)v; }
            if (P.i64_full) { *m++ = 0xBE8000FFu | ((LDS_CODE_S_VAL + 1) << 16); *m++ = (uint32_t)(v64 >> 32); }
            *m++ = 0xD2850000u | xh; *m++ = src | ((256 + xh) << 9);
            if (!P.i64_full) *m++ = v < 0 ? (0x6A000000u | (xh << 17) | (xl << 9) | (256 + xh)) : 0xBF800000u;
            *m++ = 0x68000000u | (ah << 17) | (ah << 9) | (256 + xh);
            if (P.i64_full) {
                *m++ = 0xD2850000u | xh; *m++ = (LDS_CODE_S_VAL + 1) | ((256 + xl) << 9);
                *m++ = 0x68000000u | (ah << 17) | (ah << 9) | (256 + xh);
            }
            *m++ = 0xD1E86A00u | vk; *m++ = src | ((256 + xl) << 9) | ((256 + vk) << 18);   // v_mad_u64_u32 acc[0:1], vcc, V, x_lo, acc[0:1]
            return;
        } else if (P.opcode_add == LDS_CODE_ADD_F64) {   // DBL64: the value through an SGPR pair
            const uint64_t v = T.vals_in64[T.vals[i]];
            w[6 * q] = 0xBE8000FFu | (LDS_CODE_S_VAL << 16);
            w[6 * q + 1] = (uint32_t)v;
            w[6 * q + 2] = 0xBE8000FFu | ((LDS_CODE_S_VAL + 1) << 16);
            w[6 * q + 3] = (uint32_t)(v >> 32);
            w[6 * q + 4] = 0xD2810000u | vx;
            w[6 * q + 5] = LDS_CODE_S_VAL | ((256 + vx) << 9);
        } else if (P.opcode_add == LDS_CODE_PK_ADD_U16) {   // INT16 / INT8: v_pk_mul_lo_u16 x, V, x op_sel_hi:[0,1], V inline or through s94
            uint32_t *m = w + P.mulw * q;
            if (!P.int_inline) { *m++ = 0xBE8000FFu | (LDS_CODE_S_VAL << 16); *m++ = T.vals[i]; }
            *m++ = 0xD3814000u | vx;
            *m++ = 0x10000000u | ((256 + vx) << 9) | (P.int_inline ? lds_inline_int((int32_t)T.vals[i]) : LDS_CODE_S_VAL);
        } else if (P.opcode_add == 0x68000000u) {   // INT32: the value inline, or through an SGPR
            if (P.int_inline) {
                w[2 * q] = 0xD2850000u | vx;
                w[2 * q + 1] = lds_inline_int((int32_t)T.vals[i]) | ((256 + vx) << 9);
            } else {
                w[4 * q] = 0xBE8000FFu | (LDS_CODE_S_VAL << 16);
                w[4 * q + 1] = T.vals[i];
                w[4 * q + 2] = 0xD2850000u | vx;
                w[4 * q + 3] = LDS_CODE_S_VAL | ((256 + vx) << 9);
            }
        } else {
            w[2 * q] = 0x0A0000FFu | (vx << 17) | (vx << 9);
            w[2 * q + 1] = T.vals[i];
        }
        w += P.mulw * T.g_nent[g];
    }
    if (P.half_H) {   // the lower range's adds first (one word of EXEC ahead of each range present), a range's entries in the group's order
        uint32_t q0 = 0;
        for (uint32_t j = T.g_first[g]; j < (uint32_t)i; j++) q0 += T.vals[j] == 0u;
        const uint32_t n0 = T.g_n0[g];
        w += T.vals[i] == 0u ? 1 + q0 * P.addw : (n0 ? 1 + n0 * P.addw : 0) + 1 + (q - q0) * P.addw;
    } else
        w += q * P.addw;
    if (P.opcode_add == LDS_CODE_PK_ADD_U16) {
        w[0] = 0xD38A4000u | vk;
        w[1] = 0x18000000u | ((256 + vk) << 9) | (256 + vx);
    } else if (P.opcode_add == LDS_CODE_ADD_F64) {
        w[0] = 0xD2800000u | vk;
        w[1] = (256 + vx) | ((256 + vk) << 9);
    } else if (P.opcode_add == LDS_CODE_ADD_U64) {
        w[0] = 0x32000000u | (vk << 17) | (vk << 9) | (256 + vx);
        w[1] = 0x38000000u | ((vk + 1) << 17) | ((vk + 1) << 9) | (256 + vx + 1);
    } else {
        w[0] = P.opcode_add | (vk << 17) | (vk << 9) | (256 + vx);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The pipeline on the CPU (the checker of the bodies above; the device runs lds_codegen_dev.hpp).  Returns the blob + stream offsets in
// the form of LdsCodeHost, the row map and the tile table (sorted order) for comparison with lds_plan_build / lds_code_from_plan.
// ------------------------------------------------------------------------------------------------------------------------------------
struct CgHostResult {
    std::vector<uint32_t> code;      // incl. the 8192 dwords of s_nop behind the last stream
    std::vector<uint64_t> start;
    CgRows rows;
    CgChunks chunks;
    uint64_t entries = 0, pairs = 0, shared = 0;
};
// vals: 4-byte values (FLT32 / INT32 raw bits); vals64: DBL64 values (then vals is ignored: the payload is the entry index)
inline void cg_run_on_host(const uint32_t *rowptr, const uint32_t *col, const uint32_t *vals, uint32_t nrows, uint32_t ncols, const LdsGeometry &geo,
                           uint32_t opcode_add, CgHostResult &out, uint32_t gsize = 0, uint32_t nsets = 0, const uint32_t *rorder = nullptr,
                           const uint64_t *vals64 = nullptr, uint32_t half_H = 0) {   // half_H: col holds the STORED columns, ncols = H (the virtual columns)
    std::vector<uint32_t> idx_payload;
    if (vals64) {   // (any non-null 4-byte array switches the valued form on)
        idx_payload.assign(1, 0);
        vals = idx_payload.data();
    }
    const uint64_t nnz = rowptr[nrows];
    const CgParams P = cg_params(geo, opcode_add, vals != nullptr, nrows, ncols, gsize, nsets,
                                 (vals != nullptr && !vals64 && (opcode_add == 0x68000000u || opcode_add == LDS_CODE_PK_ADD_U16) && lds_int_values_inline(vals, nnz)) ||
                                     (vals64 && opcode_add == LDS_CODE_ADD_U64 && [&] {
                                         for (uint64_t i = 0; i < nnz; i++) { const int64_t v = (int64_t)vals64[i]; if (v < -16 || v > 64) return false; }
                                         return true; }()),
                                 vals64 && opcode_add == LDS_CODE_ADD_U64 && [&] {
                                     for (uint64_t i = 0; i < nnz; i++) { const int64_t v = (int64_t)vals64[i]; if (v != (int64_t)(int32_t)v) return true; }
                                     return false; }(), half_H);
    const bool payload = vals != nullptr || half_H != 0;   // something rides with the keys through the sort
    cg_deal_rows_a(rowptr, geo, P, out.rows, rorder);
    CgTables T;
    T.rowptr = rowptr; T.colind = col; T.vals_in = vals; T.vals_in64 = vals64; T.rowinfo = out.rows.rowinfo.data(); T.nnz = nnz;
    if (P.S > 1)   // (the device counts these with one atomic per (row, range))
        for (uint32_t r = 0; r < nrows; r++)
            for (uint32_t e = rowptr[r]; e < rowptr[r + 1]; e++) out.rows.tnnz[cg_tile_of_entry(P, T, r, e)]++;
    cg_deal_rows_b(geo, P, out.rows);
    T.tile_pos = out.rows.tile_pos.data();
    std::vector<uint8_t> flags((size_t)P.ntiles * P.nchunks + 1, 0);
    T.flags = flags.data();
    for (uint32_t r = 0; r < nrows; r++)
        for (uint32_t e = rowptr[r]; e < rowptr[r + 1]; e++) cg_mark_entry(P, T, r, e);
    cg_chunk_lists(flags.data(), P, out.chunks);
    T.nch = out.chunks.nch.data(); T.choff = out.chunks.choff.data(); T.chunks = out.chunks.chunks.data();
    std::vector<uint64_t> keys(nnz + 1);
    std::vector<uint32_t> v(payload ? nnz + 1 : 1);
    T.keys = keys.data(); T.vals = v.data();
    for (uint32_t r = 0; r < nrows; r++)
        for (uint32_t e = rowptr[r]; e < rowptr[r + 1]; e++) cg_key_entry(P, T, r, e);
    {   // D2: stable sort by key (the value rides along)
        std::vector<uint32_t> perm(nnz);
        for (uint64_t i = 0; i < nnz; i++) perm[i] = (uint32_t)i;
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
        std::vector<uint64_t> k2(nnz + 1);
        std::vector<uint32_t> v2(payload ? nnz + 1 : 1);
        for (uint64_t i = 0; i < nnz; i++) {
            k2[i] = keys[perm[i]];
            if (payload) v2[i] = v[perm[i]];
        }
        keys.swap(k2);
        v.swap(v2);
        T.keys = keys.data(); T.vals = v.data();
    }
    std::vector<uint32_t> colflag(nnz + 1, 0), colx(nnz + 1, 0);
    T.colflag = colflag.data(); T.colx = colx.data();
    for (uint64_t i = 0; i < nnz; i++) cg_colflag(P, T, i);
    uint32_t run = 0;
    for (uint64_t i = 0; i < nnz; i++) { colx[i] = run; run += colflag[i]; }
    T.ncols_total = run;
    std::vector<uint32_t> col_first((size_t)run + 3, (uint32_t)nnz), col_g((size_t)run + 1, 0);
    std::vector<uint16_t> col_lrow((size_t)run + 2, 0);
    T.col_first = col_first.data(); T.col_lrow = col_lrow.data(); T.col_g = col_g.data();
    for (uint64_t i = 0; i < nnz; i++) cg_col_fill(P, T, i);
    T.nsj = (uint32_t)(P.NW * out.chunks.slots);
    std::vector<uint32_t> sfc((size_t)T.nsj + 2, 0), sng((size_t)T.nsj + 2, 0), sfg((size_t)T.nsj + 2, 0);
    T.slot_firstcol = sfc.data(); T.slot_ng = sng.data(); T.slot_firstgroup = sfg.data();
    for (uint32_t sj = 0; sj < T.nsj; sj++) {
        uint32_t s = 0, j = 0;
        cg_sj_decode(P, T, sj, &s, &j);
        if (cg_sbase(P, T, s) + j != sj || j >= T.nch[s / P.NW]) throw std::runtime_error("lds codegen: slot index decode");
        cg_slot_bounds(P, T, sj, s, j);
    }
    cg_slot_bounds(P, T, T.nsj, 0, 0);
    for (uint32_t sj = 0; sj < T.nsj; sj++) cg_slot_ngroups(P, T, sj);
    run = 0;
    for (uint32_t sj = 0; sj <= T.nsj; sj++) { sfg[sj] = run; run += sj < T.nsj ? sng[sj] : 0; }
    T.ngroups = run;
    std::vector<uint32_t> g_nent((size_t)run + 1), g_first((size_t)run + 1), g_firstcol((size_t)run + 1), g_rpos((size_t)run + 1), g_apos((size_t)run + 1);
    std::vector<uint8_t> g_nlds((size_t)run + 1), g_ncols((size_t)run + 1), g_xset((size_t)run + 1);
    std::vector<uint32_t> g_n0(half_H ? (size_t)run + 1 : 1);
    T.g_n0 = g_n0.data();
    T.g_nent = g_nent.data(); T.g_first = g_first.data(); T.g_firstcol = g_firstcol.data(); T.g_rpos = g_rpos.data(); T.g_apos = g_apos.data();
    T.g_nlds = g_nlds.data(); T.g_ncols = g_ncols.data(); T.g_xset = g_xset.data();
    unsigned long long pairs = 0;
    for (uint32_t sj = 0; sj < T.nsj; sj++) pairs += cg_slot_groups(P, T, sj);
    std::vector<uint32_t> stream_dw(P.nstreams + 1, 0);
    T.stream_dw = stream_dw.data();
    for (uint32_t s = 0; s < P.nstreams; s++) cg_stream_pass(P, T, s, false);
    out.start.assign(P.nstreams, 0);
    uint64_t total = 0;
    for (uint32_t s = 0; s < P.nstreams; s++) {
        out.start[s] = total * 4;
        total += stream_dw[s];
    }
    out.code.assign((size_t)total + 8192, 0xBF800000u);
    T.start = out.start.data();
    T.code = out.code.data();
    for (uint32_t s = 0; s < P.nstreams; s++) cg_stream_pass(P, T, s, true);
    for (uint32_t g = 0; g < T.ngroups; g++) cg_emit_reads(P, T, g);
    for (uint64_t i = 0; i < nnz; i++) cg_emit_entry(P, T, i);
    out.entries = nnz;
    out.pairs = pairs;
    out.shared = nnz - T.ncols_total;
}

}  // namespace pygim
