// lds_plan.hpp -- host-side schedule of the LDS-staged product (k_lds_spmm, lds_kernel_gen.hpp).
//
// What the schedule describes (the reference keeps a DPU's slice of the dense operand in its
// scratchpad and walks the stored entries of its rows against it,
// spmm_default/dpu_kernels/spmm_mul_csr_dpu.c:108-126; the host cuts rows into per-DPU ranges,
// spmm_mul_csr.c:118-259 -- here the "DPU" is a compute unit, its scratchpad the 160 KiB LDS and
// its row range a TILE whose running sums live in the CU's vector registers):
//
//   * rows are cut into tiles of R = NW * KA consecutive rows (NW waves of a 512-thread workgroup,
//     KA accumulator registers per wave: one register holds a row's 64-feature slice, one feature
//     per lane).  Inside a tile the rows are dealt to the waves longest first, serpentine, so that
//     every wave carries the same number of stored entries.
//   * columns are cut into chunks of KC = 320 columns; a chunk of one 64-feature slice of X is
//     80 KiB of LDS (two of them are the whole 160 KiB of a CU: fewer, longer slots than 2 x 64 KiB).  A tile lists the chunks in which it has entries (uniform graphs: all of them;
//     community-structured graphs: a few) -- the workgroup streams exactly those through a
//     double-buffered LDS ring.
//   * per (tile, wave) a token stream: chunk after chunk, inside a chunk row after row (the order
//     of the wave's accumulator index), inside a row in stored order.  A token is
//     (LDS row: column inside the chunk, + KC in odd slots = the second buffer) << 8 | accumulator index, so that one stored entry costs the wave
//     one ds_read_b32 and one indexed v_add.  Every row is summed by one wave in stored order:
//     float results are bit-identical to the sequential CPU loop.
//   * every (slot, wave) list is padded to whole batches of BATCH tokens -- at least one batch -- with tokens that add
//     a row of the chunk into a dummy accumulator.  The first batch of a list is also the slot's HEADER: bits 18..31
//     of its tokens 0, 1, 2 carry the list's batch count, the NEXT slot's batch count and the NEXT slot's chunk id (what
//     the wave needs to size its loop, its token prefetch and its share of the next chunk's DMA) -- the token stream is
//     the only array the kernel's scalar unit follows.
//
// Pure C++ (no HIP): built here for the device, and by the CPU tests against an emulator.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define PYGIM_LDS_HD_INLINE __host__ __device__ inline
#else
#define PYGIM_LDS_HD_INLINE inline
#endif

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>

namespace pygim {

struct LdsTile {            // 96 bytes, read by the kernel with scalar loads
    uint32_t nch;           // chunks (slots) this tile streams
    uint32_t chunk_off;     // first entry of its chunk-id list (nch + 2 entries, the last id repeated)
    uint32_t nb_off;        // first entry of its batch counts, [nch + 2][NW] (two closing rows of zeros)
    uint32_t row0;          // first row of the tile
    uint32_t nnz;           // stored entries of the tile
    uint32_t chunk0;        // chunk id of slot 0 (the later ones travel in the token stream's slot headers)
    uint32_t pad[2];
    uint32_t tokstart[16];  // per wave: first batch of its token stream (units of BATCH tokens)
};

constexpr uint32_t LDS_TOK_SLACK = 4096 + 64;  // tokens readable past the last one (next batch; touched lines)

// token = (LDS row of the entry's column: 10 bits, both buffers) << 8 | accumulator index; the 14 bits above carry the slot headers
constexpr uint32_t LDS_HDR_SHIFT = 18, LDS_HDR_MAX = 0x3FFFu;

struct LdsGeometry {
    uint32_t NW = 8;      // consumer waves per workgroup
    uint32_t KA = 208;    // accumulators (rows) per wave; accumulator KA is the dummy
    uint32_t KC = 320;    // columns per chunk: two chunks of 64-feature rows (2 x 80 KiB) fill the 160 KiB of a CU's LDS
    uint32_t BATCH = 16;  // tokens per batch (one scalar load)
    uint32_t rows_per_tile = 0;  // 0 = NW * KA; fewer rows per tile = more, lighter tiles (to fill whole rounds of workgroups)
    uint32_t keep_tile_order = 0;   // 1 = tiles stay in the order of the rows (neighbouring tiles stage the same chunks: launch_lds gives an XCD a run of them); 0 = heaviest first
    uint32_t col_splits = 1;     // S > 1: every row tile becomes S workgroup tiles, each with 1/S of the chunk range; tile (t, c) writes its
                                 // partial sums to row r + c * nrows (the row map says so): the caller sums the S row blocks afterwards.
                                 // For row shares too short to fill the chip with whole-X workgroups (a rank's share on N GPUs)
    uint32_t ka_stride() const { return (KA + 7) & ~7u; }   // row-map entries per wave: the store stage reads the map eight rows at a time
    uint32_t row_bytes = 256;    // bytes of a staged row of X: 256 (64 features of 4 bytes, 128 of 2) or 512 (64 features of 8 bytes: INT64 / DBL64
                                 // code streams, round 4 -- ds_read_b64 per entry, a register PAIR per accumulator and per staged value)
    uint32_t boundary = 0;       // code streams, rings of >= 3 buffers: 1 = the workgroup meets at the slot BOUNDARY (as a ring of two must): chunk
                                 // j + NBUF - 1 is issued at the start of slot j, NBUF - 1 chunks are in flight while one is read -- one more than with
                                 // the barrier in the middle of a slot -- at the price of draining the read pipeline once per slot: for plans whose
                                 // workgroups mostly land chunks (few stored entries per staged column)
    uint32_t half_split = 0;     // (round 6) 1 = products of at most 32 lanes with TWO column ranges folded into the two halves of a wave: a staged row is
                                 // [X[c] | X[c + H]] (128 bytes each), the plan is made of the virtual columns c mod H, its value slot carries each entry's half (0 / 1),
                                 // the code stream adds under the lower / upper half of EXEC, and the store adds the halves (host encoder only)
    uint32_t NBUF = 2;           // chunk buffers of the LDS ring: slot j streams into buffer j % NBUF = LDS rows [KC * (j % NBUF), + KC).
                                 // The token kernels: 2 x 320 columns.  The code-stream kernels: 3 x 192 (two chunks in flight: landing a chunk
                                 // takes ~1.1 us whatever else the CU does, and with one chunk in flight that is the length of every slot)
};

struct LdsPlanHost {
    LdsGeometry geo;
    uint32_t ntiles = 0, nchunks = 0;
    uint64_t ntokens = 0;            // incl. padding
    uint64_t slots = 0;              // sum over tiles of nch: 64 KiB chunk fills per 64-feature slice
    std::vector<uint32_t> tok;       // ntokens + LDS_TOK_SLACK (the kernel loads one batch ahead and touches lines 4 KiB ahead)
    std::vector<uint32_t> wts;       // valued matrices: the entries' 4-byte values (raw bits) in token order, 0 for padding; else empty
    std::vector<uint32_t> nb, chunks, rowmap;
    std::vector<LdsTile> tiles;      // heaviest tile first (workgroups are dispatched in index order)
    std::atomic<bool> header_overflow{false};   // (set by any worker) a batch count or chunk id does not fit a 16-bit header field: the plan cannot be used
};

// runs fn(i) for i in [0, n) on `threads` threads; an exception in a worker is carried to the caller (a std::thread body that
// throws would end the process)
// host threads worth starting: the hardware's, or the container's CPU quota when that is lower (cgroup v2 cpu.max / v1 cfs quota: the GPU
// boxes of this pool show 256 CPUs and grant 16 -- 256 workers then take longer than 16)
inline unsigned lds_default_threads() {
    static const unsigned cached = [] {
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        long long quota = -1, period = 100000;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(h, "%lld", &period) != 1) period = 100000;
                fclose(h);
            }
        }
        if (quota > 0 && period > 0) hw = (unsigned)std::min<long long>(hw, std::max<long long>(1, (quota + period - 1) / period));
        return hw;
    }();
    return cached;
}

template <typename F> inline void lds_parallel_for(uint32_t n, unsigned threads, F &&fn) {
    if (threads == 0) threads = lds_default_threads();
    threads = std::min<unsigned>(threads, std::max(1u, n));
    std::atomic<uint32_t> next(0);
    std::atomic<bool> failed(false);
    auto body = [&]() {
        try {
            for (;;) {
                const uint32_t i = next.fetch_add(1);
                if (i >= n || failed.load(std::memory_order_relaxed)) return;
                fn(i);
            }
        } catch (...) {
            failed.store(true);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (unsigned i = 1; i < threads; i++) pool.emplace_back(body);
    } catch (...) {
        // (no more threads: the ones that started, and this one, do the work)
    }
    body();
    for (auto &th : pool) th.join();
    if (failed.load()) throw std::runtime_error("lds plan: a worker failed (out of memory?)");
}

// rowptr / col: CSR with sorted column ids inside every row (checked by the caller).
// threads = 0: std::thread::hardware_concurrency().
// rorder (round 5): a permutation of the rows -- tile t holds rows rorder[t * R .. (t + 1) * R) instead of R consecutive ones (rows with
// similar neighbourhoods side by side share LDS reads); nullptr = consecutive rows.  Every row is still summed by one wave in stored order.
inline void lds_plan_build(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const LdsGeometry &geo,
                           LdsPlanHost &out, unsigned threads = 0, const uint32_t *vals = nullptr, const uint32_t *rorder = nullptr) {
    auto rid = [&](uint32_t pos) { return rorder ? rorder[pos] : pos; };   // row at position pos of the tile order
    auto rlen = [&](uint32_t pos) { const uint32_t r = rid(pos); return rowptr[r + 1] - rowptr[r]; };
    const uint32_t NW = geo.NW, KA = geo.KA, KC = geo.KC, B = geo.BATCH;
    const uint32_t KAS = geo.ka_stride();
    const uint32_t RS = NW * KAS;                                                      // row-map stride of a tile
    const uint32_t R = geo.rows_per_tile ? std::min(geo.rows_per_tile, NW * KA) : NW * KA;   // rows of a tile
    const uint32_t S = std::max(1u, geo.col_splits);
    const uint32_t ntiles = ((nrows + R - 1) / R) * S;                                // workgroup tiles: (row tile, column range)
    const uint32_t nchunks = (ncols + KC - 1) / KC;
    out.geo = geo;
    out.ntiles = ntiles;
    out.nchunks = nchunks;
    out.tiles.assign(ntiles, LdsTile());
    out.rowmap.assign((size_t)ntiles * RS, 0xFFFFFFFFu);

    // per tile: the (wave, k) of every row, the chunk list, per (slot, wave) token counts
    struct TileTmp {
        std::vector<uint32_t> chunk_ids;   // ascending
        std::vector<uint32_t> cnt;         // [nch][NW] tokens (unpadded)
        std::vector<uint8_t> wave_of;      // per row of the tile
        std::vector<uint16_t> k_of;
        uint64_t batches[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    };
    std::vector<TileTmp> tmp(ntiles);
    auto run = [&](auto &&fn) { lds_parallel_for(ntiles, threads, fn); };   // (a worker's exception is carried to the caller)
    run([&](uint32_t t) {
        TileTmp &tt = tmp[t];
        const uint32_t rt = t / S, cs = t % S;
        const uint32_t ch_lo = (uint32_t)((uint64_t)nchunks * cs / S), ch_hi = (uint32_t)((uint64_t)nchunks * (cs + 1) / S);   // this tile's chunk range
        const uint32_t r0 = rt * R, r1 = std::min(nrows, r0 + R), nr = r1 - r0;
        // longest rows first (stable), dealt serpentine over the waves
        std::vector<uint32_t> order(nr);
        for (uint32_t i = 0; i < nr; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return rlen(r0 + a) > rlen(r0 + b);
        });
        tt.wave_of.resize(nr);
        tt.k_of.resize(nr);
        for (uint32_t i = 0; i < nr; i++) {
            const uint32_t round = i / NW, pos = i % NW;
            const uint32_t w = (round & 1) ? NW - 1 - pos : pos;
            tt.wave_of[order[i]] = (uint8_t)w;
            tt.k_of[order[i]] = (uint16_t)round;
            out.rowmap[((size_t)t * NW + w) * KAS + round] = rid(r0 + order[i]) + cs * nrows;   // (row of the partial-sum block of column range cs)
        }
        // token counts per (chunk, wave)
        std::vector<uint32_t> cnt((size_t)nchunks * NW, 0);
        for (uint32_t i = 0; i < nr; i++) {
            const uint32_t w = tt.wave_of[i];
            for (uint32_t e = rowptr[rid(r0 + i)]; e < rowptr[rid(r0 + i) + 1]; e++) {
                const uint32_t ch = col[e] / KC;
                if (ch >= ch_lo && ch < ch_hi) cnt[(size_t)ch * NW + w]++;
            }
        }
        uint32_t nnz_range = 0;
        for (uint32_t c = 0; c < nchunks; c++) {
            uint32_t any = 0;
            for (uint32_t w = 0; w < NW; w++) any |= cnt[(size_t)c * NW + w];
            if (!any) continue;
            tt.chunk_ids.push_back(c);
            for (uint32_t w = 0; w < NW; w++) {
                const uint32_t n = cnt[(size_t)c * NW + w];
                nnz_range += n;
                tt.cnt.push_back(n);
                tt.batches[w] += std::max<uint32_t>(1, (n + B - 1) / B);   // at least the header batch
            }
        }
        out.tiles[t].nch = (uint32_t)tt.chunk_ids.size();
        out.tiles[t].chunk0 = tt.chunk_ids.empty() ? 0 : tt.chunk_ids[0];
        out.tiles[t].row0 = r0;
        out.tiles[t].nnz = nnz_range;
    });
    // offsets
    uint64_t tokb = 0, nbo = 0, cho = 0, slots = 0;
    for (uint32_t t = 0; t < ntiles; t++) {
        LdsTile &d = out.tiles[t];
        d.chunk_off = (uint32_t)cho;
        d.nb_off = (uint32_t)nbo;
        cho += d.nch + 2;
        nbo += (uint64_t)(d.nch + 2) * NW;
        slots += d.nch;
        for (uint32_t w = 0; w < NW; w++) {
            d.tokstart[w] = (uint32_t)tokb;
            tokb += tmp[t].batches[w];
        }
    }
    out.slots = slots;
    out.ntokens = tokb * B;
    out.tok.assign((size_t)out.ntokens + LDS_TOK_SLACK, KA);  // everything starts as padding (column 0 -> the dummy accumulator)
    if (vals) out.wts.assign(out.tok.size(), 0u);
    else out.wts.clear();
    out.nb.assign((size_t)nbo + 256, 0);      // (+ slack: the kernel touches lines a few slots ahead)
    out.chunks.assign((size_t)cho + 128, 0);
    run([&](uint32_t t) {
        TileTmp &tt = tmp[t];
        const LdsTile &d = out.tiles[t];
        const uint32_t rt = t / S, cs = t % S;
        const uint32_t r0 = rt * R, r1 = std::min(nrows, r0 + R), nr = r1 - r0, nch = d.nch;
        for (uint32_t j = 0; j < nch + 2; j++) out.chunks[(size_t)d.chunk_off + j] = nch ? tt.chunk_ids[std::min(j, nch - 1)] : 0;
        // where each (slot, wave) list starts inside the wave's stream
        std::vector<uint32_t> slot_of(nchunks, 0xFFFFFFFFu);
        for (uint32_t j = 0; j < nch; j++) slot_of[tt.chunk_ids[j]] = j;
        std::vector<uint64_t> cursor((size_t)nch * NW);
        for (uint32_t w = 0; w < NW; w++) {
            uint64_t at = (uint64_t)d.tokstart[w] * B;
            for (uint32_t j = 0; j < nch; j++) {
                cursor[(size_t)j * NW + w] = at;
                const uint32_t nbat = std::max<uint32_t>(1, (tt.cnt[(size_t)j * NW + w] + B - 1) / B);
                out.nb[(size_t)d.nb_off + (size_t)j * NW + w] = nbat;
                at += (uint64_t)nbat * B;
            }
        }
        auto batches_of = [&](uint32_t j, uint32_t w) { return j < nch ? std::max<uint32_t>(1, (tt.cnt[(size_t)j * NW + w] + B - 1) / B) : 0u; };
        // rows in accumulator order per wave == any order that visits a wave's rows k = 0, 1, ...: walk k-major
        for (uint32_t k = 0; k < KA; k++)
            for (uint32_t w = 0; w < NW; w++) {
                const uint32_t rowm = out.rowmap[((size_t)t * NW + w) * KAS + k];
                if (rowm == 0xFFFFFFFFu) continue;
                const uint32_t row = rowm - cs * nrows;
                for (uint32_t e = rowptr[row]; e < rowptr[row + 1]; e++) {
                    const uint32_t c = col[e];
                    const uint32_t j = slot_of[c / KC];
                    if (j == 0xFFFFFFFFu) continue;   // (an entry of another column range)
                    const uint64_t at = cursor[(size_t)j * NW + w]++;
                    out.tok[at] = (((c % KC) + (j % geo.NBUF) * KC) << 8) | k;   // slot j reads LDS buffer j % NBUF: row KC * (j % NBUF) + (c % KC)
                    if (vals) out.wts[at] = vals[e];
                }
            }
        // slot headers: upper halves of tokens 0..2 of every list's first batch
        for (uint32_t w = 0; w < NW; w++) {
            uint64_t at = (uint64_t)d.tokstart[w] * B;
            for (uint32_t j = 0; j < nch; j++) {
                const uint32_t nbj = batches_of(j, w), nbn = batches_of(j + 1, w), cidn = j + 1 < nch ? tt.chunk_ids[j + 1] : 0;
                if (nbj > LDS_HDR_MAX || nbn > LDS_HDR_MAX || cidn > LDS_HDR_MAX) out.header_overflow = true;
                out.tok[at] |= nbj << LDS_HDR_SHIFT;
                out.tok[at + 1] |= nbn << LDS_HDR_SHIFT;
                out.tok[at + 2] |= cidn << LDS_HDR_SHIFT;
                at += (uint64_t)nbj * B;
            }
        }
        (void)nr;
    });
    // heaviest tile first
    std::vector<uint32_t> ord(ntiles);
    for (uint32_t t = 0; t < ntiles; t++) ord[t] = t;
    if (!geo.keep_tile_order) std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return out.tiles[a].nnz > out.tiles[b].nnz; });
    std::vector<LdsTile> tiles2(ntiles);
    std::vector<uint32_t> rowmap2(out.rowmap.size());
    for (uint32_t i = 0; i < ntiles; i++) {
        tiles2[i] = out.tiles[ord[i]];
        std::copy(out.rowmap.begin() + (size_t)ord[i] * RS, out.rowmap.begin() + (size_t)(ord[i] + 1) * RS, rowmap2.begin() + (size_t)i * RS);
    }
    out.tiles.swap(tiles2);
    out.rowmap.swap(rowmap2);
}

// Rows per tile such that the tiles x slices workgroups of one product fill whole rounds of the device's compute units
// (one workgroup per CU at a time): 152 tiles x 4 slices on 256 CUs would run as three rounds with the third 37 % full;
// 192 lighter tiles run as three full, shorter ones.  Only worth it for a handful of rounds.
inline uint32_t lds_rows_per_tile(uint32_t nrows, uint32_t rmax, uint32_t nslices, uint32_t cus) {
    if (!nrows || !nslices || !cus) return rmax;
    const uint64_t tiles = (nrows + rmax - 1) / rmax, wgs = tiles * nslices;
    const uint64_t rounds = (wgs + cus - 1) / cus;
    if (rounds > 8) return rmax;
    if (wgs % cus == 0) return (uint32_t)((nrows + tiles - 1) / tiles);   // whole rounds already: tiles of equal height
    // as many tiles as fit the same number of rounds (less than one round: enough tiles to give every CU a workgroup --
    // the caller's reuse rule then decides whether such light tiles are still worth staging X for)
    // (rounded DOWN: 171 tiles x 3 slices = 513 workgroups would start a third round for one workgroup; 170 x 3 = 510 do not)
    const uint64_t tiles2 = rounds * cus / nslices;
    if (tiles2 <= tiles) return rmax;
    return (uint32_t)std::min<uint64_t>(rmax, std::max<uint64_t>(16, (nrows + tiles2 - 1) / tiles2));
}

// Cheap pre-check from the row pointers alone: stored entries per staged column if every tile streamed every chunk
// (the uniform-graph bound; community-structured graphs stream fewer chunks and do better).
inline double lds_plan_uniform_reuse(uint64_t nnz, uint32_t nrows, uint32_t ncols, const LdsGeometry &geo) {
    const uint64_t R = geo.rows_per_tile ? geo.rows_per_tile : (uint64_t)geo.NW * geo.KA;
    const uint64_t ntiles = (nrows + R - 1) / R;
    if (!ntiles || !ncols) return 0;
    return (double)nnz / ((double)ntiles * (double)ncols);
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same schedule compiled into gfx950 MACHINE CODE ("code stream", round 3): per (tile, wave) one straight-line instruction
// stream that the kernel jumps into once.  A stored entry is then
//     ds_read2st64_b32 x[2i : 2i+1], base offset0:row_a offset1:row_b     -- ONE LDS instruction for TWO entries (its two 8-bit
//                                                                            offsets count 256-byte rows: a chunk's row stride)
//     v_add_f32 acc[k], x, acc[k]                                         -- the accumulator is the instruction's register field
// i.e. 1.5 instructions and 8.5 bytes of code per entry, no address arithmetic, no index register, no token loads, no batch
// bookkeeping and no padding (measured in isolation, scripts/micro/codestream.hip: 2.56 CU cycles per entry streamed from memory
// once against 3.1-3.3 for the four-instruction token of the token kernels).  The chunk hand-off (DMA of a later chunk, wait,
// barrier) is inlined with the chunk ids as literals; every 1 KB the stream touches its own lines 2 KiB ahead into the L2.
//
// Round 4: two geometries and two hand-offs.
//   * 16 waves x 96 accumulators (1 536 rows per tile at most, 4 waves per SIMD) -- round 3's -- and 8 waves x 228 accumulators
//     (1 824 rows: the Reddit-shaped product of four slices is then TWO rounds of workgroups on 256 CUs instead of three, i.e. a third
//     less of X streamed through LDS; 2 waves per SIMD with 256 VGPRs each).  The reads run NSETS - 1 groups of GSIZE entries ahead of
//     the adds (x-register sets), and the pipeline is not drained at a slot boundary.
//   * BOUNDARY hand-off (any ring; the only one a ring of two allows; the default): the workgroup meets at the end of slot j -- all its
//     reads of chunk j have returned, everybody's pieces of chunk j + 1 have landed -- and chunk j + NBUF is issued behind the first
//     reads of the next slot, into the buffer of chunk j: NBUF - 1 chunks are in flight while one is read (ring of 5 x 128 columns: four,
//     128 KiB), the last group's adds cross the barrier.
//     MID-SLOT hand-off (rings of three or more buffers, geometry.boundary = 0): ONE barrier in the MIDDLE of slot j says "everybody is
//     done with chunk j - 1 and has landed chunk j + 1"; behind it chunk j + NBUF - 1 is issued into the buffer of chunk j - 1.  Nobody
//     waits at the slot boundary and the reads in flight cross it, but only NBUF - 2 chunks are in flight -- measured 1-2 % behind the
//     boundary form on every shape (profiles/r04_lds_kernel.md).
// Register contract with the kernel (scripts/gen_lds_kernel.py body_code): see LdsCodeRegs.
// ---------------------------------------------------------------------------------------------------------------------------
struct LdsCodeRegs {
    // VGPRs
    uint32_t vbase[3] = {0, 0, 0};   // lane * 4 + 0 / 65536 / 131072 (LDS rows 0..255, 256..511, 512..639)
    uint32_t vl16 = 0;       // lane * 16 (DMA)
    uint32_t vtouch = 0;     // lane offsets of a touch: 8 lines of 128 bytes
    uint32_t vjunk = 0;      // destination of the touches
    uint32_t x0 = 0;         // x registers: NSETS sets of GSIZE
    uint32_t gsize = 0, nsets = 0;
    uint32_t acc0 = 0;       // accumulators v[acc0] .. v[acc0 + KA - 1]
    // SGPRs
    uint32_t s_xs = 80;      // s[80:81]: this slice of X + wave * PIECE (bytes)
    uint32_t s_ldsw = 82;    // LDS byte address of this wave's DMA piece inside a chunk buffer
    uint32_t s_cb = 84;      // s[84:85]: code touch pointer
    uint32_t s_ret = 86;     // s[86:87]: return address
    uint32_t s_pa = 92;      // s[92:93]: DMA source
    uint32_t wide = 0;       // 1: 8-byte elements -- x values and accumulators are register pairs
    constexpr uint32_t nx() const { return gsize * nsets * (wide ? 2 : 1); }   // x REGISTERS
};
// 16 waves: v4 v5 v10 bases, v6 lane * 16, v11 (lane % 8) * 128, v9 junk, x v12..v27, accumulators v28..v123
// 8 waves:  v0 is the lane id (the only register the compiler keeps), v1..v3 bases, v4 lane * 16 (DMA and touch), v5 junk,
//           x v6..v27 (22 registers: two sets of 10 or three of 6), accumulators v28..v255
constexpr LdsCodeRegs lds_code_regs(uint32_t NW, uint32_t gsize = 0, uint32_t nsets = 0, bool wide = false) {
    LdsCodeRegs r;
    r.wide = wide ? 1 : 0;
    if (NW == 16) {
        r.vbase[0] = 4; r.vbase[1] = 5; r.vbase[2] = 10;
        r.vl16 = 6; r.vtouch = 11; r.vjunk = 9; r.x0 = 12; r.acc0 = 28;
        r.gsize = gsize ? gsize : 8;
        r.nsets = nsets ? nsets : 2;
    } else {
        r.vbase[0] = 1; r.vbase[1] = 2; r.vbase[2] = 3;
        r.vl16 = 4; r.vtouch = 4; r.vjunk = 5; r.x0 = 6; r.acc0 = 28;
        r.gsize = gsize ? gsize : (wide ? 5 : 10);
        r.nsets = nsets ? nsets : 2;
    }
    if (!wide && (r.gsize & 1)) r.gsize--;   // (4-byte rows are read two to an instruction)
    if (r.gsize < 2) r.gsize = 2;
    if (r.nsets < 2) r.nsets = 2;
    while (r.x0 + r.nx() > r.acc0 && r.gsize > 2) r.gsize -= wide ? 1 : 2;
    return r;
}
// what build_lds_plan chooses when the tunables leave it open (measured: profiles/r04_lds_kernel.md)
constexpr uint32_t LDS_CODE_AUTO_WAVES = 8, LDS_CODE8_AUTO_NBUF = 5;
constexpr uint32_t LDS_CODE8_KA64 = 114;  // ... of its 8-byte form: a register pair per accumulator
constexpr uint32_t LDS_CODE8_KA = 228;   // accumulators per wave of the 8-wave code-stream geometry (v28..v255)

// a plain array of words that is NOT value-initialised (a std::vector would write the 1 GB of a Reddit-sized code blob once before the
// streams are copied into it)
struct LdsWords {
    std::unique_ptr<uint32_t[]> p;
    size_t n = 0;
    void alloc(size_t words) { p.reset(new uint32_t[words]); n = words; }
    uint32_t *data() { return p.get(); }
    const uint32_t *data() const { return p.get(); }
    size_t size() const { return n; }
    uint32_t &operator[](size_t i) { return p[i]; }
    const uint32_t &operator[](size_t i) const { return p[i]; }
};

struct LdsCodeHost {
    LdsWords code;                       // all streams, each 256-byte aligned, + slack behind the last (the touches read ahead)
    std::vector<uint64_t> start;         // [ntiles][NW]: byte offset of a (tile, wave) stream
    uint64_t entries = 0;                // stored entries compiled (no padding)
    uint64_t pairs = 0;                  // of which read two to an LDS instruction
    uint64_t shared = 0;                 // entries served by the read of an earlier entry of the same column (no read of their own)
    LdsCodeRegs regs;                    // the register map the code was written for
};

// opcode_add: the VOP2 opcode field of the accumulate (v_add_f32 = 0x02000000, v_add_u32 = 0x68000000), or LDS_CODE_PK_ADD_U16 for
// INT16 (two features to a lane: v_pk_add_u16, a VOP3P instruction of 8 bytes)
constexpr uint32_t LDS_CODE_PK_ADD_U16 = 0xFFFFFFFFu;
// 8-byte elements (geometry row_bytes = 512): v_add_f64 on register pairs; v_add_co_u32 + v_addc_co_u32
constexpr uint32_t LDS_CODE_ADD_F64 = 0xFFFFFFFEu, LDS_CODE_ADD_U64 = 0xFFFFFFFDu;
// experiment (timing only, WRONG results): bit 0 = no workgroup barriers, bit 1 = no chunk DMA (what the hand-offs / the fill cost:
// 2.09 -> 1.81 / 1.63 / both 1.49 ms on the bench workload, profiles/r04_lds_kernel.md)
// Valued matrices: FLT32 -- the value is the literal of a v_mul_f32 (VOP2).  INT32 (round 5): gfx9's VOP3 takes no literal, so v_mul_lo_u32 gets the value
// as an INLINE CONSTANT when every value of the matrix lies in [-16, 64] (lds_int_values_inline: 8 bytes of code per multiply), else through an
// SGPR (s_mov_b32 s94, <literal>; v_mul_lo_u32 x, s94, x: 16 bytes).  Products wrap modulo 2^32 like the CPU loop's.
constexpr uint32_t LDS_CODE_S_VAL = 94;   // the SGPR a general integer value travels in (free between the kernel's set-up and its store stage)
inline bool lds_int_values_inline(const uint32_t *vals, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) {
        const int32_t v = (int32_t)vals[i];
        if (v < -16 || v > 64) return false;
    }
    return true;
}
PYGIM_LDS_HD_INLINE uint32_t lds_inline_int(int32_t v) { return v >= 0 ? 128u + (uint32_t)v : 192u + (uint32_t)(-v); }   // VOP3 source field of an inline integer
// INT16 / INT8 (round 5): the value sign-extended in the 4-byte slot; v_pk_mul_lo_u16 x, V, x op_sel_hi:[0,1] (both halves of x times V's low half), V inline or s94.
// DBL64 (round 5): the plan's 4-byte value slot carries the ENTRY INDEX and vals64 the values; per entry s_mov_b32 s94, <low half>; s_mov_b32 s95,
// <high half>; v_mul_f64 x, s[94:95], x (24 bytes), product and sum rounded separately.  (INT64, values that fit int32: v_mul_lo_u32 on the high half + v_mad_u64_u32 -- consume_oldest; wider values: the sweep.)
inline void lds_code_from_plan(const LdsPlanHost &plan, uint32_t opcode_add, LdsCodeHost &out, unsigned threads = 0, uint32_t gsize = 0,
                               uint32_t nsets = 0, uint32_t experiment = 0, const uint64_t *vals64 = nullptr) {
    const LdsGeometry &geo = plan.geo;
    const uint32_t NW = geo.NW, KA = geo.KA, B = geo.BATCH, KC = geo.KC;
    const uint32_t RB = geo.row_bytes;
    const bool wide = RB == 512;
    if (RB != 256 && RB != 512) throw std::runtime_error("lds code: rows of 256 or 512 bytes");
    if (wide != (opcode_add == LDS_CODE_ADD_F64 || opcode_add == LDS_CODE_ADD_U64)) throw std::runtime_error("lds code: 512-byte rows are the 8-byte element types'");
    if (!plan.wts.empty() && (opcode_add == LDS_CODE_ADD_F64 || opcode_add == LDS_CODE_ADD_U64) && !vals64)
        throw std::runtime_error("lds code: valued entries of this element type");
    const LdsCodeRegs R = lds_code_regs(NW, gsize, nsets, wide);
    out.regs = R;
    const uint32_t G = R.gsize, NS = R.nsets;
    if (R.x0 + R.nx() > R.acc0 || R.acc0 + KA * (wide ? 2 : 1) > 256 || G > 12 || (NS - 1) * G > 15)
        throw std::runtime_error("lds code: the geometry does not fit the register map");
    const uint32_t pieces = (KC * RB / 1024) / NW;   // 1 KiB DMA pieces of a chunk per wave
    if (pieces * NW * 1024 != KC * RB) throw std::runtime_error("lds code: a chunk is not a whole number of pieces per wave");
    const uint32_t chunk_bytes = KC * RB;
    const uint32_t RPB = 65536 / RB;                  // rows of a 64 KiB LDS block (one base register each)
    if ((KC * geo.NBUF + RPB - 1) / RPB > 3) throw std::runtime_error("lds code: the ring is larger than three 64 KiB blocks");
    const uint32_t ntiles = plan.ntiles;
    // a touch per 1 KB of code: the 8 lines (1 KB) that start 2 KiB ahead (the kernel's lane offsets repeat every 8 lanes).  A wave
    // consumes ~0.2 bytes of code per cycle: 2 KiB is > 10 000 cycles of lead.  (First version: 64 lines from 8 KiB ahead every 6 KB --
    // 512 waves per XCD x 8 KiB of lead is the whole 4 MiB L2: touched lines were evicted before they were fetched, fabric traffic
    // 17.7 GB per product and an L2 hit rate of 80 %.)
    constexpr uint32_t TOUCH_EVERY_DW = 256;

    // the instruction words of a stream go straight to their place in the final blob; a first pass with out == nullptr only counts them
    // (two passes over the schedule instead of one: cheaper than 1 GB of per-stream vectors, their page faults and the copy)
    struct Emit {
        uint32_t *out = nullptr;
        size_t n = 0;
        uint32_t since_touch = 0;
        uint64_t vm_touch = 0;             // touches emitted (vector loads, counted into vmcnt)
        void put(uint32_t a) {
            if (out) out[n] = a;
            n++;
        }
        void op(uint32_t a) { put(a); since_touch++; }
        void op(uint32_t a, uint32_t b) { put(a); put(b); since_touch += 2; }
    };
    struct Grp {                           // a group of up to GSIZE staged columns whose reads have been issued, and the entries they serve
        uint32_t nx = 0, nlds = 0, xb = 0;
        std::vector<uint32_t> k, xr, v;    // per entry: accumulator, x register, value (valued matrices)
    };
    const LdsCodeRegs Rr = R;
    // (padding tokens carry the value 0: inside the inline range)
    bool int_inline = (opcode_add == 0x68000000u || opcode_add == LDS_CODE_PK_ADD_U16) && !plan.wts.empty() && lds_int_values_inline(plan.wts.data(), plan.wts.size());
    bool i64_full = false;   // INT64: some value needs more than 32 bits -- both halves travel through s[94:95]
    if (opcode_add == LDS_CODE_ADD_U64 && !plan.wts.empty()) {   // INT64: inline when all values lie in [-16, 64], one SGPR when all fit int32
        int_inline = true;
        for (uint64_t i = 0; i < plan.ntokens; i++) {
            if ((plan.tok[i] & 0xFF) >= KA) continue;   // padding
            const int64_t v = (int64_t)vals64[plan.wts[i]];
            if (v != (int64_t)(int32_t)v) i64_full = true;
            if (v < -16 || v > 64) int_inline = false;
        }
        if (i64_full) int_inline = false;
    }
    auto emit_stream = [&](uint32_t ti, uint32_t wv, Emit &e, uint64_t &n_entries, uint64_t &n_pairs, uint64_t &n_shared) {
        const LdsTile &t = plan.tiles[ti];
        auto s_add_lit = [&](uint32_t sdst, uint32_t ssrc, uint32_t lit) { e.op(0x80000000u | (sdst << 16) | (0xFFu << 8) | ssrc, lit); };
        auto s_addc0 = [&](uint32_t sdst, uint32_t ssrc) { e.op(0x82000000u | (sdst << 16) | (0x80u << 8) | ssrc); };
        auto touch = [&]() {
            s_add_lit(Rr.s_cb, Rr.s_cb, (e.since_touch + 5) * 4);   // (+ the five dwords of this touch itself)
            s_addc0(Rr.s_cb + 1, Rr.s_cb + 1);
            e.op(0xDC508000u, (Rr.vjunk << 24) | (Rr.s_cb << 16) | Rr.vtouch);   // global_load_dword vjunk, vtouch, s[cb:cb+1]
            e.since_touch = 0;
            e.vm_touch++;
        };
        auto dma = [&](uint32_t cid, uint32_t buf) {
            s_add_lit(Rr.s_pa, Rr.s_xs, cid * chunk_bytes);
            s_addc0(Rr.s_pa + 1, Rr.s_xs + 1);
            for (uint32_t i = 0; i < pieces; i++) {
                if (i % 4 == 0) {
                    if (i) {
                        s_add_lit(Rr.s_pa, Rr.s_pa, 0x1000);
                        s_addc0(Rr.s_pa + 1, Rr.s_pa + 1);
                    }
                    s_add_lit(124 /* m0 */, Rr.s_ldsw, buf * chunk_bytes + (i / 4) * 0x1000);
                    e.op(0xBF800000u);                              // s_nop 0 (M0 write -> LDS-DMA)
                }
                if (!(experiment & 2)) e.op(0xDDF48000u | ((i % 4) * 1024), (Rr.s_pa << 16) | Rr.vl16);   // global_load_lds_dwordx4 vl16, s[pa:pa+1] offset
            }
        };
        uint64_t at = (uint64_t)t.tokstart[wv] * B;
        const uint32_t NBUF = geo.NBUF;
        const bool mid = NBUF >= 3 && !geo.boundary;                // where the workgroup meets: in the middle of a slot, or at its boundary
        // vector loads issued so far (DMA pieces and touches): the wait for "chunk j + 1 has landed" names how many YOUNGER loads may
        // still be in flight -- known when the code is generated (vmcnt retires in issue order)
        uint64_t dma_issued = 0;
        auto vm_now = [&]() { return dma_issued + e.vm_touch; };    // vector loads issued so far: DMA pieces + touches
        std::vector<uint64_t> landed_mark(t.nch + NBUF + 1, 0);     // vm_now() right after the last piece of chunk j
        auto dma_chunk = [&](uint32_t j) {                          // chunk of slot j -> buffer j % NBUF
            dma(plan.chunks[t.chunk_off + j], j % NBUF);
            dma_issued += (experiment & 2) ? 0 : pieces;
            landed_mark[j] = vm_now();
        };
        // experiment bit 2 (results stay RIGHT): the pieces of a chunk go out one behind every group of reads instead of all behind the slot's
        // first group -- a wave that waits for room in the vector memory queue then waits with reads in flight
        const bool spread = (experiment & 4) && pieces <= 4 && !mid;
        uint32_t dma_left = 0, dma_j = 0;
        auto dma_begin = [&](uint32_t j) {
            const uint32_t cid = plan.chunks[t.chunk_off + j], buf = j % NBUF;
            s_add_lit(Rr.s_pa, Rr.s_xs, cid * chunk_bytes);
            s_addc0(Rr.s_pa + 1, Rr.s_xs + 1);
            s_add_lit(124 /* m0 */, Rr.s_ldsw, buf * chunk_bytes);
            e.op(0xBF800000u);
            dma_left = pieces;
            dma_j = j;
        };
        auto dma_piece = [&]() {
            const uint32_t i = pieces - dma_left;
            e.op(0xDDF48000u | (i * 1024), (Rr.s_pa << 16) | Rr.vl16);
            dma_issued++;
            if (!--dma_left) landed_mark[dma_j] = vm_now();
        };
        auto wait_landed = [&](uint32_t j) {                        // s_waitcnt vmcnt(N): everything up to chunk j's last piece has landed
            const uint64_t younger = vm_now() - landed_mark[j];
            const uint32_t nn = (uint32_t)std::min<uint64_t>(younger, 63);
            e.op(0xBF8C0F70u | (nn & 15) | (((nn >> 4) & 3) << 14));
        };
        auto wait_lgkm = [&](uint32_t n) { e.op(0xBF8CC07Fu | (std::min(n, 15u) << 8)); };
        const bool halves = geo.half_split != 0;                    // (the value slot holds the entry's half, not a value)
        const bool valued = !plan.wts.empty() && !halves;
        // the reads in flight: groups whose LDS instructions have been issued and whose adds have not, oldest first
        std::vector<Grp> ring(NS + 1);                              // (re-used: no allocation per group)
        std::vector<uint32_t> pend;                                 // indices into ring
        uint32_t gcount = 0;                                        // groups issued so far: group g reads into x-set g % NSETS
        uint32_t lds_this_slot = 0;                                 // LDS instructions issued since the slot began
        bool older_reads = false;                                   // a read of an EARLIER slot may still be in flight
        auto consume_oldest = [&]() {                               // wait for the oldest group's reads, then its adds
            uint32_t younger = 0;
            for (size_t q = 1; q < pend.size(); q++) younger += ring[pend[q]].nlds;
            wait_lgkm(younger);
            if (younger <= lds_this_slot) older_reads = false;      // (LDS reads return in order)
            const Grp &g = ring[pend.front()];
            if (valued && opcode_add == LDS_CODE_ADD_U64) {
                // INT64 (round 5), values that fit int32: acc += x * v modulo 2^64 from 32-bit pieces, v_u = v mod 2^32 as the operand V --
                //   hi(acc) += lo32(x_hi * V) - [v < 0] * x_lo ;  acc += x_lo * V (v_mad_u64_u32: the 32 x 32 -> 64-bit product AND the 64-bit sum in one
                //   instruction; the first form -- v_mul_hi_u32, v_mul_lo_u32, v_add_co, v_addc_co -- was 48 bytes of code per entry, this one is 32)
                // Values of more than 32 bits: x * v = x_lo * v_lo + 2^32 * (x_hi * v_lo + x_lo * v_hi) modulo 2^64, every piece unsigned (48 bytes).
                for (size_t q = 0; q < g.k.size(); q++) {
                    const uint64_t v64 = vals64[g.v[q]];
                    const int32_t v = (int32_t)(uint32_t)v64;
                    const uint32_t xl = g.xr[q], xh = g.xr[q] + 1, al = Rr.acc0 + g.k[q] * 2, ah = al + 1;
                    const uint32_t src = int_inline ? lds_inline_int(v) : LDS_CODE_S_VAL;
                    if (!int_inline) e.op(0xBE8000FFu | (LDS_CODE_S_VAL << 16), (uint32_t)v);                    // s_mov_b32 s94, <value / its low half>
                    if (i64_full) e.op(0xBE8000FFu | ((LDS_CODE_S_VAL + 1) << 16), (uint32_t)(v64 >> 32));        // s_mov_b32 s95, <high half>
                    e.op(0xD2850000u | xh, src | ((256 + xh) << 9));                                              // v_mul_lo_u32 x_hi, V, x_hi
                    if (!i64_full) e.op(v < 0 ? (0x6A000000u | (xh << 17) | (xl << 9) | (256 + xh)) : 0xBF800000u);   // v_sub_u32 x_hi, x_hi, x_lo  |  s_nop
                    e.op(0x68000000u | (ah << 17) | (ah << 9) | (256 + xh));                                      // v_add_u32 acc_hi, x_hi, acc_hi
                    if (i64_full) {
                        e.op(0xD2850000u | xh, (LDS_CODE_S_VAL + 1) | ((256 + xl) << 9));                         // v_mul_lo_u32 x_hi, V_hi, x_lo
                        e.op(0x68000000u | (ah << 17) | (ah << 9) | (256 + xh));                                  // v_add_u32 acc_hi, x_hi, acc_hi
                    }
                    e.op(0xD1E86A00u | al, src | ((256 + xl) << 9) | ((256 + al) << 18));                         // v_mad_u64_u32 acc[0:1], vcc, V, x_lo, acc[0:1]
                }
            } else if (valued && opcode_add == LDS_CODE_ADD_F64) {   // DBL64: the value through an SGPR pair
                for (size_t q = 0; q < g.k.size(); q++) {
                    const uint64_t v = vals64[g.v[q]];
                    e.op(0xBE8000FFu | (LDS_CODE_S_VAL << 16), (uint32_t)v);
                    e.op(0xBE8000FFu | ((LDS_CODE_S_VAL + 1) << 16), (uint32_t)(v >> 32));
                    e.op(0xD2810000u | g.xr[q], LDS_CODE_S_VAL | ((256 + g.xr[q]) << 9));              // v_mul_f64 x[0:1], s[94:95], x[0:1]
                }
            } else if (valued && opcode_add == LDS_CODE_PK_ADD_U16) {   // INT16 (and INT8 on its widened features): both halves of x times the value's low half
                for (size_t q = 0; q < g.k.size(); q++) {
                    if (!int_inline) e.op(0xBE8000FFu | (LDS_CODE_S_VAL << 16), g.v[q]);                      // s_mov_b32 s94, <value, sign-extended>
                    e.op(0xD3814000u | g.xr[q], 0x10000000u | ((256 + g.xr[q]) << 9) | (int_inline ? lds_inline_int((int32_t)g.v[q]) : LDS_CODE_S_VAL));   // v_pk_mul_lo_u16 x, V, x op_sel_hi:[0,1]
                }
            } else if (valued && opcode_add == 0x68000000u) {   // INT32: x = val * x (wrapping), the value inline or through an SGPR
                for (size_t q = 0; q < g.k.size(); q++) {
                    if (int_inline) e.op(0xD2850000u | g.xr[q], lds_inline_int((int32_t)g.v[q]) | ((256 + g.xr[q]) << 9));
                    else {
                        e.op(0xBE8000FFu | (LDS_CODE_S_VAL << 16), g.v[q]);                            // s_mov_b32 s94, <literal value>
                        e.op(0xD2850000u | g.xr[q], LDS_CODE_S_VAL | ((256 + g.xr[q]) << 9));          // v_mul_lo_u32 x, s94, x
                    }
                }
            } else if (valued)   // acc += val * x, product and sum rounded separately (no FMA), as the CPU loop: the products first (every entry has
                          // its own x register here: the multiply overwrites it)
                for (size_t q = 0; q < g.k.size(); q++) e.op(0x0A0000FFu | (g.xr[q] << 17) | (g.xr[q] << 9), g.v[q]);   // v_mul_f32 x, <literal value>, x
            // half-split plans: the adds of the entries whose column lies in the lower range under the lower half of EXEC (s[76:77], set by the kernel), then the
            // upper range's under the upper half (s[78:79]); a row's entries of one range keep their order
            for (uint32_t pass = 0; pass < (halves ? 2u : 1u); pass++) {
            if (halves) {
                bool any = false;
                for (size_t q = 0; q < g.k.size() && !any; q++) any = (g.v[q] & 1u) == pass;
                if (!any) continue;
                e.op(0xBEFE0100u | (pass ? 78u : 76u));                                                                      // s_mov_b64 exec, s[76:77] / s[78:79]
            }
            for (size_t q = 0; q < g.k.size() && !(valued && opcode_add == LDS_CODE_ADD_U64); q++) {   // (valued INT64: the sum is part of the v_mad_u64_u32 above)
                if (halves && (g.v[q] & 1u) != pass) continue;
                const uint32_t vk = Rr.acc0 + g.k[q] * (wide ? 2 : 1), vx = g.xr[q];
                if (opcode_add == LDS_CODE_PK_ADD_U16) e.op(0xD38A4000u | vk, 0x18000000u | ((256 + vk) << 9) | (256 + vx));   // v_pk_add_u16 acc, x, acc
                else if (opcode_add == LDS_CODE_ADD_F64) e.op(0xD2800000u | vk, (256 + vx) | ((256 + vk) << 9));             // v_add_f64 acc[0:1], x[0:1], acc[0:1]
                else if (opcode_add == LDS_CODE_ADD_U64) {
                    e.op(0x32000000u | (vk << 17) | (vk << 9) | (256 + vx));                                                  // v_add_co_u32 acc0, vcc, x0, acc0
                    e.op(0x38000000u | ((vk + 1) << 17) | ((vk + 1) << 9) | (256 + vx + 1));                                  // v_addc_co_u32 acc1, vcc, x1, acc1, vcc
                } else e.op(opcode_add | (vk << 17) | (vk << 9) | (256 + vx));
            }
            }
            if (halves) e.op(0xBEFE01C1u);                                                                                     // s_mov_b64 exec, -1
            pend.erase(pend.begin());
        };
        // the first NBUF - 1 chunks, then the slots
        if (t.nch) {
            for (uint32_t j = 0; j + 1 < NBUF && j < t.nch; j++) dma_chunk(j);
            wait_landed(0);
            if (!(experiment & 1)) e.op(0xBF8A0000u);               // s_barrier
        }
        std::vector<uint64_t> toks;   // token | value << 32 (valued FLT32 matrices: the entry's value rides as a literal of its v_mul_f32)
        std::vector<uint64_t> toks2;
        std::vector<uint32_t> keys;
        struct Col { uint32_t row, first, cnt; };
        std::vector<Col> cols;        // (re-used: no allocation per slot -- 256 threads in malloc were most of the creation time)
        for (uint32_t j = 0; j < t.nch; j++) {
            // boundary form (any ring): the DMA of chunk j + NBUF - 1 (into the buffer the boundary barrier just freed) goes behind the
            // slot's FIRST group of reads (their LDS latency covers its issue; in front of them: 2.056 against 2.040 ms).
            // mid-slot form (rings of >= 3 buffers): the hand-off sits in the middle of the slot (below)
            bool dma_due = !mid && j + NBUF - 1 < t.nch;
            bool handoff_due = mid;
            const uint32_t nb = plan.tok[at] >> LDS_HDR_SHIFT;
            toks.clear();
            for (uint32_t b = 0; b < nb * B; b++) {
                const uint32_t tk = plan.tok[at + b] & ((1u << LDS_HDR_SHIFT) - 1);
                if ((tk & 0xFF) < KA) toks.push_back((uint64_t)tk | ((valued || halves) ? (uint64_t)plan.wts[at + b] << 32 : 0));
            }
            at += (uint64_t)nb * B;
            // by staged column (stable): every row's entries are in column order already, so each row's order is kept; entries of
            // different rows of this wave that share a column now sit side by side and share ONE read (8 waves x 228 rows: a fifth of
            // the entries of a uniform Reddit-shaped slot), and neighbours are in the same 256-row LDS block: they pair up
            // (std::stable_sort allocates a merge buffer per call -- 1.9 M calls from 256 threads; the order is made total instead: LDS row,
            // then position in the list, in one 32-bit key, sorted in place)
            if (toks.size() < 65536) {
                keys.resize(toks.size());
                for (uint32_t q = 0; q < toks.size(); q++) keys[q] = (((uint32_t)toks[q] >> 8) << 16) | q;
                std::sort(keys.begin(), keys.end());
                toks2.resize(toks.size());
                for (uint32_t q = 0; q < toks.size(); q++) toks2[q] = toks[keys[q] & 0xFFFFu];
                toks.swap(toks2);
            } else {
                std::stable_sort(toks.begin(), toks.end(), [](uint64_t a, uint64_t b) { return ((uint32_t)a >> 8) < ((uint32_t)b >> 8); });
            }
            n_entries += toks.size();
            lds_this_slot = 0;
            older_reads = !pend.empty();
            // the hand-off of a ring of >= 3 buffers, once per slot and wave: my pieces of chunk j + 1 have landed, my reads of chunk
            // j - 1 have returned; behind the barrier chunk j + NBUF - 1 goes into the buffer of chunk j - 1
            auto handoff = [&]() {
                if (j + 1 < t.nch) wait_landed(j + 1);
                if (older_reads) {
                    wait_lgkm(lds_this_slot);
                    older_reads = false;
                }
                if (!(experiment & 1)) e.op(0xBF8A0000u);           // s_barrier
                // (the pieces go out at once: letting the waves take turns behind the barrier -- wave w first reads w mod 2 / 4 / 8 more
                // groups -- measured 2.15 / 2.13 / 2.11 ms against 2.06: the fill is on the critical path, earlier is better)
                if (j + NBUF - 1 < t.nch) dma_chunk(j + NBUF - 1);
                handoff_due = false;
            };
            // the staged columns of the slot: (LDS row, first entry, entries); valued matrices: one per entry
            cols.clear();
            for (uint32_t q = 0; q < toks.size(); q++) {
                const uint32_t r = (uint32_t)toks[q] >> 8;
                if (!valued && !cols.empty() && cols.back().row == r) cols.back().cnt++;
                else cols.push_back({r, q, 1});
            }
            n_shared += toks.size() - cols.size();
            const size_t ngroups_est = (cols.size() + G - 1) / G;
            size_t gi = 0, i = 0;
            while (i < cols.size()) {
                Grp &g = ring[gcount % (NS + 1)];
                g.nx = g.nlds = 0;
                g.k.clear();
                g.xr.clear();
                g.v.clear();
                const uint32_t XW = wide ? 2 : 1;                   // registers of a staged value
                g.xb = Rr.x0 + G * XW * (gcount % NS);
                while (g.nx < G && i < cols.size()) {
                    const uint32_t r0 = cols[i].row, blk = r0 / RPB;
                    uint32_t took = 1;
                    if (wide) {
                        e.op(0xD8EC0000u | ((r0 % RPB) * RB), ((g.xb + g.nx * 2) << 24) | Rr.vbase[blk]);                    // ds_read_b64 x[0:1], base offset:row * 512
                    } else if (g.nx + 2 <= G && (g.nx & 1) == 0 && i + 1 < cols.size() && (cols[i + 1].row >> 8) == blk) {
                        const uint32_t r1 = cols[i + 1].row;
                        e.op(0xD8700000u | ((r1 & 255) << 8) | (r0 & 255), ((g.xb + g.nx) << 24) | Rr.vbase[blk]);   // ds_read2st64_b32
                        took = 2;
                        n_pairs += cols[i].cnt + cols[i + 1].cnt;
                    } else {
                        e.op(0xD86C0000u | ((r0 & 255) << 8), ((g.xb + g.nx) << 24) | Rr.vbase[blk]);             // ds_read_b32
                    }
                    for (uint32_t u = 0; u < took; u++)
                        for (uint32_t q = cols[i + u].first; q < cols[i + u].first + cols[i + u].cnt; q++) {
                            g.k.push_back((uint32_t)toks[q] & 0xFF);
                            g.xr.push_back(g.xb + (g.nx + u) * XW);
                            g.v.push_back((uint32_t)(toks[q] >> 32));
                        }
                    g.nx += took;
                    i += took;
                    g.nlds++;
                }
                lds_this_slot += g.nlds;
                pend.push_back(gcount % (NS + 1));
                gcount++;
                if (dma_due) {
                    if (spread) dma_begin(j + NBUF - 1);
                    else dma_chunk(j + NBUF - 1);
                    dma_due = false;
                }
                if (dma_left) dma_piece();
                if (pend.size() >= NS) consume_oldest();            // frees the x-set the next group reads into
                if (handoff_due && gi + 1 >= (ngroups_est + 1) / 2) handoff();
                gi++;
                if (e.since_touch >= TOUCH_EVERY_DW) touch();
            }
            if (dma_due) dma_chunk(j + NBUF - 1);                   // (a slot without entries for this wave)
            while (dma_left) dma_piece();
            if (handoff_due) handoff();
            if (!mid) {
                // everybody's reads of this chunk have returned and everybody has landed the next before anybody goes on; the adds of
                // the groups still in flight overlap the wait, the last group's adds cross the barrier
                while (pend.size() > 1) consume_oldest();
                wait_lgkm(0);
                older_reads = false;
                if (j + 1 < t.nch) wait_landed(j + 1);              // my pieces of the NEXT chunk have landed (younger loads may still fly)
                if (!(experiment & 1)) e.op(0xBF8A0000u);           // s_barrier (timing experiments: only every 2nd / 4th / none: 1.97 / 1.93 / 1.81 ms against 2.04)
            }
        }
        while (!pend.empty()) consume_oldest();
        e.op(0xBF8C0F70u);                                          // s_waitcnt vmcnt(0): no touch is left in flight
        e.op(0xBE801D00u | Rr.s_ret);                               // s_setpc_b64 s[ret:ret+1]
        while (e.n % 64) e.put(0xBF800000u);                         // streams start on 256-byte lines
    };

    const uint32_t nstreams = ntiles * NW;
    std::vector<uint64_t> ne(nstreams, 0), np(nstreams, 0), nsh(nstreams, 0), words(nstreams, 0);
    // (one pass into per-stream vectors and a copy was measured at 1.8-2.0 s for the Reddit-sized graph against 0.72-0.78 s for the two
    // passes below, on the pool's boxes: 16 CPUs of quota, and page faults on 1 GB of fresh vectors)
    lds_parallel_for(nstreams, threads, [&](uint32_t s) {   // pass 1: sizes
        Emit e;
        uint64_t a = 0, b = 0, c = 0;
        emit_stream(s / NW, s % NW, e, a, b, c);
        words[s] = e.n;
    });
    out.start.assign(nstreams, 0);
    uint64_t total = 0;
    for (uint32_t s = 0; s < nstreams; s++) {
        out.start[s] = total * 4;
        total += words[s];
    }
    out.code.alloc((size_t)total + 8192);                           // (+ 32 KB of s_nop behind the last stream: touches read ahead)
    std::fill(out.code.data() + total, out.code.data() + total + 8192, 0xBF800000u);
    lds_parallel_for(nstreams, threads, [&](uint32_t s) {   // pass 2: the words, in place
        Emit e;
        e.out = out.code.data() + out.start[s] / 4;
        emit_stream(s / NW, s % NW, e, ne[s], np[s], nsh[s]);
        if (e.n != words[s]) throw std::runtime_error("lds code: the two passes over a stream disagree");
    });
    for (uint32_t s = 0; s < nstreams; s++) {
        out.entries += ne[s];
        out.pairs += np[s];
        out.shared += nsh[s];
    }
}

}  // namespace pygim
