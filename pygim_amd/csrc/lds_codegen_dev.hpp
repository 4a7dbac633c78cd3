// lds_codegen_dev.hpp -- the data-parallel code-stream encoder (lds_codegen.hpp) on the device: its bodies as HIP kernels, a prefix sum
// and a stable LSD radix sort (hand-written: one-time set-up steps, bandwidth-bound passes over the entries), and the driver that strings
// them together against the resident CSR.  The instruction words go straight into executable memory; the host sees the row pointers
// (it has them), the per-tile chunk flags and the per-stream sizes -- kilobytes -- instead of the graph.
// Reference cost being matched: the reference's one-time step is a partition walk and a copy (spmm_default/spmm_mul_csr.c:118-330).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <string>

#include "lds_codegen.hpp"

namespace pygim {

// ------------------------------------------------------------------------------------------------------------------------------------
// exclusive prefix sum of uint32 (n up to 2^32 - 1 elements, sums modulo 2^32): blocks of 2048, recursive over the block sums
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t CG_SCAN_ITEMS = 8, CG_SCAN_TILE = 256 * CG_SCAN_ITEMS;

__device__ inline uint32_t cg_block_exclusive(uint32_t v, uint32_t *total) {   // 256 threads; returns the exclusive prefix of v over the block
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if ((int)lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void k_cg_scan_block(const uint32_t *in, uint32_t *out, uint32_t *block_sums, uint64_t n) {
    const uint64_t i0 = (uint64_t)blockIdx.x * CG_SCAN_TILE + (uint64_t)threadIdx.x * CG_SCAN_ITEMS;
    uint32_t v[CG_SCAN_ITEMS], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < CG_SCAN_ITEMS; k++) {
        v[k] = i0 + k < n ? in[i0 + k] : 0u;
        sum += v[k];
    }
    uint32_t total;
    uint32_t run = cg_block_exclusive(sum, &total);
#pragma unroll
    for (uint32_t k = 0; k < CG_SCAN_ITEMS; k++) {
        if (i0 + k < n) out[i0 + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void k_cg_scan_add(uint32_t *out, const uint32_t *block_offsets, uint64_t n) {
    const uint32_t add = block_offsets[blockIdx.x];
    const uint64_t i0 = (uint64_t)blockIdx.x * CG_SCAN_TILE + (uint64_t)threadIdx.x * CG_SCAN_ITEMS;
#pragma unroll
    for (uint32_t k = 0; k < CG_SCAN_ITEMS; k++)
        if (i0 + k < n) out[i0 + k] += add;
}
// scratch: at least cg_scan_scratch_words(n) uint32.  in == out allowed.
inline uint64_t cg_scan_scratch_words(uint64_t n) {
    uint64_t w = 0;
    while (n > 1) {
        n = (n + CG_SCAN_TILE - 1) / CG_SCAN_TILE;
        w += n + 1;
    }
    return w + 2;
}
inline void cg_scan_exclusive(const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *scratch, hipStream_t st) {
    if (n == 0) return;
    const uint64_t nb = (n + CG_SCAN_TILE - 1) / CG_SCAN_TILE;
    hipLaunchKernelGGL(k_cg_scan_block, dim3((unsigned)nb), dim3(256), 0, st, in, out, scratch, n);
    if (nb > 1) {
        cg_scan_exclusive(scratch, scratch, nb, scratch + nb + 1, st);
        hipLaunchKernelGGL(k_cg_scan_add, dim3((unsigned)nb), dim3(256), 0, st, out, (const uint32_t *)scratch, n);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// stable LSD radix sort of 64-bit keys (+ optional 32-bit values), 8 bits per pass.  A WAVE owns a tile of 2048 consecutive keys: per
// pass a histogram, a prefix sum over [digit][tile], and a scatter in which the lanes of a wave that hold the same digit find each other
// with eight ballots -- rank = number of lower lanes with the same digit, so the order of equal digits is kept (stable)
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t CG_SORT_ITEMS = 32, CG_SORT_TILE = 64 * CG_SORT_ITEMS;

__global__ __launch_bounds__(256) void k_cg_radix_hist(const uint64_t *keys, uint32_t *hist, uint64_t n, uint32_t ntiles, uint32_t shift) {
    __shared__ uint32_t h[4][256];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t d = lane; d < 256; d += 64) h[wave][d] = 0;
    __syncthreads();
    const uint64_t tile = (uint64_t)blockIdx.x * 4 + wave;
    if (tile < ntiles) {
        const uint64_t base = tile * CG_SORT_TILE;
        for (uint32_t it = 0; it < CG_SORT_ITEMS; it++) {
            const uint64_t i = base + (uint64_t)it * 64 + lane;
            if (i < n) atomicAdd(&h[wave][(uint32_t)(keys[i] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (tile < ntiles)
        for (uint32_t d = lane; d < 256; d += 64) hist[(uint64_t)d * ntiles + tile] = h[wave][d];
}

template <bool HAS_VALS>
__global__ __launch_bounds__(256) void k_cg_radix_scatter(const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout, const uint32_t *offsets,
                                                          uint64_t n, uint32_t ntiles, uint32_t shift) {
    __shared__ uint32_t off_s[4][256];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t tile = (uint64_t)blockIdx.x * 4 + wave;
    if (tile < ntiles)
        for (uint32_t d = lane; d < 256; d += 64) off_s[wave][d] = offsets[(uint64_t)d * ntiles + tile];
    __syncthreads();
    if (tile >= ntiles) return;
    volatile uint32_t *off = off_s[wave];
    const uint64_t base = tile * CG_SORT_TILE;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (uint32_t it = 0; it < CG_SORT_ITEMS; it++) {
        const uint64_t i = base + (uint64_t)it * 64 + lane;
        const bool valid = i < n;
        const uint64_t key = valid ? kin[i] : 0ull;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        uint64_t same = __ballot(valid);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(valid && bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(same & below);
        uint32_t pos = 0;
        if (valid) pos = off[d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) off[d] = off[d] + (uint32_t)__popcll(same);   // (one lane per digit present)
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            kout[pos] = key;
            if (HAS_VALS) vout[pos] = vin[i];
        }
    }
}

// sorts n keys by bits [0, key_bits); the result ends in (*keys_a, *vals_a) -- the pointers are swapped as the passes ping-pong.
// hist: 256 * ntiles + 1 words; scan_scratch: cg_scan_scratch_words(256 * ntiles)
inline uint32_t cg_sort_tiles(uint64_t n) { return (uint32_t)((n + CG_SORT_TILE - 1) / CG_SORT_TILE); }
inline void cg_radix_sort(uint64_t **keys_a, uint64_t **keys_b, uint32_t **vals_a, uint32_t **vals_b, uint64_t n, uint32_t key_bits, uint32_t *hist,
                          uint32_t *scan_scratch, hipStream_t st) {
    if (n == 0) return;
    const uint32_t ntiles = cg_sort_tiles(n);
    const unsigned blocks = (ntiles + 3) / 4;
    for (uint32_t shift = 0; shift < key_bits; shift += 8) {
        hipLaunchKernelGGL(k_cg_radix_hist, dim3(blocks), dim3(256), 0, st, (const uint64_t *)*keys_a, hist, n, ntiles, shift);
        cg_scan_exclusive(hist, hist, (uint64_t)256 * ntiles, scan_scratch, st);
        if (vals_a && *vals_a)
            hipLaunchKernelGGL((k_cg_radix_scatter<true>), dim3(blocks), dim3(256), 0, st, (const uint64_t *)*keys_a, *keys_b, (const uint32_t *)*vals_a, *vals_b,
                               (const uint32_t *)hist, n, ntiles, shift);
        else
            hipLaunchKernelGGL((k_cg_radix_scatter<false>), dim3(blocks), dim3(256), 0, st, (const uint64_t *)*keys_a, *keys_b, (const uint32_t *)nullptr,
                               (uint32_t *)nullptr, (const uint32_t *)hist, n, ntiles, shift);
        std::swap(*keys_a, *keys_b);
        if (vals_a && *vals_a) std::swap(*vals_a, *vals_b);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// the bodies of lds_codegen.hpp as kernels
// ------------------------------------------------------------------------------------------------------------------------------------
template <int WHAT>   // 0 = mark chunks, 1 = keys, 2 = entries per (row tile, column range): a wave per row, lanes over its entries
__global__ __launch_bounds__(256) void k_cg_rows(CgParams P, CgTables T) {
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= P.nrows) return;
    const uint32_t e0 = T.rowptr[row], e1 = T.rowptr[row + 1], lane = threadIdx.x & 63;
    if (WHAT == 2) {
        // one atomic per (row, range) and wave step: the lanes that hold the same tile find each other (a row's ranges are runs: sorted columns)
        for (uint32_t base = e0; base < e1; base += 64) {
            const uint32_t e = base + lane;
            const bool valid = e < e1;
            const uint32_t t = valid ? cg_tile_of_entry(P, T, (uint32_t)row, e) : 0u;
            uint64_t todo = __ballot(valid);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t tl = (uint32_t)__shfl((int)t, leader, 64);
                const uint64_t same = __ballot(valid && t == tl);
                if ((int)lane == leader) atomicAdd(&T.tnnz[tl], (uint32_t)__popcll(same));
                todo &= ~same;
            }
        }
        return;
    }
    for (uint32_t e = e0 + lane; e < e1; e += 64) {
        if (WHAT == 0) cg_mark_entry(P, T, (uint32_t)row, e);
        else cg_key_entry(P, T, (uint32_t)row, e);
    }
}
template <int WHAT>   // per sorted entry: 0 = column flags, 1 = column fill, 2 = the entry's instructions
__global__ __launch_bounds__(256) void k_cg_entries(CgParams P, CgTables T) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T.nnz) return;
    if (WHAT == 0) cg_colflag(P, T, i);
    else if (WHAT == 1) cg_col_fill(P, T, i);
    else cg_emit_entry(P, T, i);
}
template <int WHAT>   // per (stream, slot): 0 = first column, 1 = group count, 2 = groups
__global__ __launch_bounds__(256) void k_cg_slots(CgParams P, CgTables T) {
    const uint64_t sj = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (WHAT == 0) {
        if (sj > T.nsj) return;
        uint32_t s = 0, j = 0;
        if (sj < T.nsj) cg_sj_decode(P, T, (uint32_t)sj, &s, &j);
        cg_slot_bounds(P, T, (uint32_t)sj, s, j);
    } else if (WHAT == 1) {
        if (sj > T.nsj) return;
        if (sj < T.nsj) cg_slot_ngroups(P, T, (uint32_t)sj);
        else T.slot_ng[sj] = 0;
    } else {
        if (sj >= T.nsj) return;
        const unsigned long long pairs = cg_slot_groups(P, T, (uint32_t)sj);
        if (pairs) atomicAdd(&T.stats[0], pairs);
    }
}
__global__ __launch_bounds__(64) void k_cg_streams(CgParams P, CgTables T, int write) {   // one lane per stream: a sequential walk over its groups
    if (threadIdx.x != 0 || blockIdx.x >= P.nstreams) return;
    cg_stream_pass(P, T, blockIdx.x, write != 0);
}
__global__ __launch_bounds__(256) void k_cg_reads(CgParams P, CgTables T) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g < T.ngroups) cg_emit_reads(P, T, (uint32_t)g);
}
// valued INT32: does every value lie in the inline-constant range [-16, 64]?  flag[0] is set when one does not (lds_int_values_inline)
__global__ __launch_bounds__(256) void k_cg_int_outside(const uint32_t *vals, uint64_t n, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int32_t v = (int32_t)vals[i];
        if (v < -16 || v > 64) flag[0] = 1u;
    }
}
// valued INT64: flag[0] |= 1 when a value lies outside the inline-constant range [-16, 64], |= 2 when one does not fit 32 bits
__global__ __launch_bounds__(256) void k_cg_i64_range(const uint64_t *vals, uint64_t n, uint32_t *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int64_t v = (int64_t)vals[i];
        uint32_t f = 0;
        if (v < -16 || v > 64) f |= 1u;
        if (v != (int64_t)(int32_t)v) f |= 2u;
        if (f) atomicOr(flag, f);
    }
}
// 0 = every value inline, 1 = every value fits int32, 2 = some value needs more than 32 bits; -1 = the check failed
inline int cg_i64_value_class(const uint64_t *d_vals64, uint64_t n, hipStream_t st) {
    uint32_t *d_flag = nullptr, h_flag = 2;
    if (hipMalloc((void **)&d_flag, 64) != hipSuccess) { (void)hipGetLastError(); return -1; }
    (void)hipMemsetAsync(d_flag, 0, 64, st);
    if (n) hipLaunchKernelGGL(k_cg_i64_range, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_vals64, n, d_flag);
    const bool got = hipMemcpyAsync(&h_flag, d_flag, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    (void)hipFree(d_flag);
    if (!got) { (void)hipGetLastError(); return -1; }
    return (h_flag & 2u) ? 2 : (h_flag & 1u) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_cg_fill_words(uint32_t *p, uint64_t n, uint32_t word) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = word;
}
__global__ __launch_bounds__(256) void k_cg_tail2(const uint32_t *colflag, const uint32_t *colx, uint64_t nnz, uint32_t *out) {   // out[0] = staged columns in all
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = nnz ? colx[nnz - 1] + colflag[nnz - 1] : 0u;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// the driver
// ------------------------------------------------------------------------------------------------------------------------------------
struct CgDeviceResult {
    void *code = nullptr;              // executable memory (from alloc_exec), code_bytes long
    size_t code_bytes = 0;
    uint64_t *d_start = nullptr;       // hipMalloc: [nstreams]
    uint32_t *d_rowmap = nullptr;      // hipMalloc
    LdsTile *d_tiles = nullptr;        // hipMalloc (nch, chunk0, row0, nnz filled; the token fields are not used by the code-stream kernels)
    uint32_t ntiles = 0;
    uint64_t slots = 0, entries = 0, pairs = 0, shared = 0;
    LdsCodeRegs regs;
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // phase times (PYGIM_PLAN_TIMING): prep, mark+lists, keys+sort, columns, slots+groups, sizes, emit, total
};

// alloc_exec(bytes) returns executable device memory or nullptr; free_exec(ptr) releases it.  Returns "" on success, else why not
// (nothing is left allocated then).  h_rowptr: the row pointers on the host (the caller has them); h_rorder: the tile order of the rows
// (similarity tiles, lds_reorder_dev.hpp) or nullptr for consecutive rows.
static bool g_cg_fail_big_alloc = false;   // lds_fail bit 8 (rt_plans.inc)

inline std::string cg_run_on_device(const uint32_t *d_rowptr, const uint32_t *d_col, const uint32_t *d_vals, const uint64_t *d_vals64, const uint32_t *h_rowptr,
                                    const uint32_t *h_rorder,
                                    uint32_t nrows, uint32_t ncols,
                                    const LdsGeometry &geo, uint32_t opcode_add, uint32_t gsize, uint32_t nsets, hipStream_t st,
                                    const std::function<void *(size_t)> &alloc_exec, const std::function<void(void *)> &free_exec, CgDeviceResult &out,
                                    const std::function<double()> &now_ms, uint32_t half_H = 0) {   // half_H: a half-split plan -- d_col holds the STORED columns, ncols = H
    const uint64_t nnz = h_rowptr[nrows];
    if (nnz == 0 || nnz >= (1ull << 31)) return "lds codegen: no entries, or 2^31 and more";
    bool int_inline = false;
    if (d_vals && (opcode_add == 0x68000000u || opcode_add == LDS_CODE_PK_ADD_U16)) {   // valued INT32 / INT16: inline constants when every value allows it (one pass over the values)
        uint32_t *d_flag = nullptr, h_flag = 1;
        if (hipMalloc((void **)&d_flag, 64) != hipSuccess) { (void)hipGetLastError(); return "lds codegen: out of device memory"; }
        (void)hipMemsetAsync(d_flag, 0, 64, st);
        hipLaunchKernelGGL(k_cg_int_outside, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, d_vals, nnz, d_flag);
        const bool got = hipMemcpyAsync(&h_flag, d_flag, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        (void)hipFree(d_flag);
        if (!got) { (void)hipGetLastError(); return "lds codegen: value range check failed"; }
        int_inline = h_flag == 0;
    }
    bool i64_full = false;
    if (d_vals64 && opcode_add == LDS_CODE_ADD_U64) {   // valued INT64: inline when every value allows it, one SGPR when all fit int32, else both halves
        const int cls = cg_i64_value_class(d_vals64, nnz, st);
        if (cls < 0) return "lds codegen: value range check failed";
        int_inline = cls == 0;
        i64_full = cls == 2;
    }
    CgParams P;
    try {
        P = cg_params(geo, opcode_add, d_vals != nullptr || d_vals64 != nullptr, nrows, ncols, gsize, nsets, int_inline, i64_full, half_H);
    } catch (const std::exception &e) {
        return e.what();
    }
    std::vector<void *> tmp;           // device temporaries, freed on every way out
    bool failed = false;
    auto dalloc = [&](size_t bytes) -> void * {
        void *p = nullptr;
        // (g_cg_fail_big_alloc: tests -- the first transient of 1 MiB and more "runs out of device memory", as the sort keys of a large graph may)
        if (failed || (g_cg_fail_big_alloc && bytes >= (1u << 20)) || hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess) {
            (void)hipGetLastError();
            failed = true;
            return nullptr;
        }
        tmp.push_back(p);
        return p;
    };
    auto release = [&](void *p) {   // a temporary that is no longer needed (waits for the work that uses it)
        for (auto &q : tmp)
            if (q == p && p) { (void)hipFree(p); q = nullptr; }
    };
    void *keep[3] = {nullptr, nullptr, nullptr};   // results handed to the caller
    auto bail = [&](const std::string &why) {
        (void)hipStreamSynchronize(st);
        for (void *p : tmp)
            if (p) (void)hipFree(p);
        for (void *p : keep)
            if (p) (void)hipFree(p);
        if (out.code) free_exec(out.code);
        out = CgDeviceResult();
        return why.empty() ? std::string("lds codegen: failed") : why;
    };
    auto ok = [&](hipError_t e) { if (e != hipSuccess) { (void)hipGetLastError(); failed = true; } return e == hipSuccess; };
    const double t0 = now_ms();
    double t_mark = t0;
    int phase = 0;
    auto lap = [&]() { const double t = now_ms(); out.ms[phase++] = t - t_mark; t_mark = t; };

    // host: rows -> tiles / waves / accumulators (from the row pointers alone); column-split plans: the entries of every (row tile, column
    // range) are counted on the device before the tiles are put in launch order (heaviest first)
    CgRows rows;
    try {
        cg_deal_rows_a(h_rowptr, geo, P, rows, h_rorder);
    } catch (const std::exception &e) {
        return bail(e.what());
    }
    CgTables T;
    T.rowptr = d_rowptr; T.colind = d_col; T.vals_in = d_vals; T.vals_in64 = d_vals64; T.nnz = nnz;
    uint32_t *d_rowinfo = (uint32_t *)dalloc((size_t)nrows * 4);
    uint8_t *d_flags = (uint8_t *)dalloc((size_t)P.ntiles * P.nchunks + 1);
    uint32_t *d_tile_pos = (uint32_t *)dalloc((size_t)P.ntiles * 4), *d_tnnz = (uint32_t *)dalloc((size_t)P.ntiles * 4);
    if (failed) return bail("lds codegen: out of device memory (row tables)");
    if (!ok(hipMemcpyAsync(d_rowinfo, rows.rowinfo.data(), (size_t)nrows * 4, hipMemcpyHostToDevice, st)) ||
        !ok(hipMemsetAsync(d_flags, 0, (size_t)P.ntiles * P.nchunks + 1, st)) || !ok(hipMemsetAsync(d_tnnz, 0, (size_t)P.ntiles * 4, st)))
        return bail("lds codegen: row table upload");
    T.rowinfo = d_rowinfo;
    T.flags = d_flags;
    T.tnnz = d_tnnz;
    const unsigned row_blocks = (unsigned)(((uint64_t)nrows + 3) / 4);
    if (P.S > 1) {
        hipLaunchKernelGGL((k_cg_rows<2>), dim3(row_blocks), dim3(256), 0, st, P, T);
        if (!ok(hipMemcpyAsync(rows.tnnz.data(), d_tnnz, (size_t)P.ntiles * 4, hipMemcpyDeviceToHost, st)) || !ok(hipStreamSynchronize(st)))
            return bail("lds codegen: entries per (row tile, column range)");
    }
    cg_deal_rows_b(geo, P, rows);
    if (!ok(hipMemcpyAsync(d_tile_pos, rows.tile_pos.data(), (size_t)P.ntiles * 4, hipMemcpyHostToDevice, st))) return bail("lds codegen: tile order upload");
    T.tile_pos = d_tile_pos;
    lap();
    // D0: chunks present per tile -> chunk lists (host)
    hipLaunchKernelGGL((k_cg_rows<0>), dim3(row_blocks), dim3(256), 0, st, P, T);
    std::vector<uint8_t> h_flags((size_t)P.ntiles * P.nchunks + 1);
    if (!ok(hipMemcpyAsync(h_flags.data(), d_flags, h_flags.size(), hipMemcpyDeviceToHost, st)) || !ok(hipStreamSynchronize(st))) return bail("lds codegen: chunk flags");
    CgChunks ch;
    cg_chunk_lists(h_flags.data(), P, ch);
    if ((uint64_t)P.NW * ch.slots + 2 >= (1ull << 32)) return bail("lds codegen: more than 2^32 (stream, slot) pairs");
    uint32_t *d_nch = (uint32_t *)dalloc((size_t)P.ntiles * 4), *d_choff = (uint32_t *)dalloc(((size_t)P.ntiles + 1) * 4);
    uint32_t *d_chunks = (uint32_t *)dalloc(ch.chunks.size() * 4);
    if (failed) return bail("lds codegen: out of device memory (chunk lists)");
    if (!ok(hipMemcpyAsync(d_nch, ch.nch.data(), (size_t)P.ntiles * 4, hipMemcpyHostToDevice, st)) ||
        !ok(hipMemcpyAsync(d_choff, ch.choff.data(), ((size_t)P.ntiles + 1) * 4, hipMemcpyHostToDevice, st)) ||
        !ok(hipMemcpyAsync(d_chunks, ch.chunks.data(), ch.chunks.size() * 4, hipMemcpyHostToDevice, st)))
        return bail("lds codegen: chunk list upload");
    T.nch = d_nch; T.choff = d_choff; T.chunks = d_chunks;
    T.nsj = (uint32_t)(P.NW * ch.slots);
    release(d_flags);
    lap();
    // D1 + D2: keys, stable sort
    uint64_t *keys_a = (uint64_t *)dalloc((nnz + 1) * 8), *keys_b = (uint64_t *)dalloc((nnz + 1) * 8);
    uint32_t *vals_a = nullptr, *vals_b = nullptr;
    const bool payload = P.valued || P.half_H;   // something rides with the keys through the sort (values; the entries' halves)
    if (payload) {
        vals_a = (uint32_t *)dalloc((nnz + 1) * 4);
        vals_b = (uint32_t *)dalloc((nnz + 1) * 4);
    }
    const uint32_t sort_tiles = cg_sort_tiles(nnz);
    uint32_t *d_hist = (uint32_t *)dalloc(((size_t)256 * sort_tiles + 1) * 4);
    const uint64_t scan_words = std::max(cg_scan_scratch_words((uint64_t)256 * sort_tiles), std::max(cg_scan_scratch_words(nnz), cg_scan_scratch_words((uint64_t)T.nsj + 1)));
    uint32_t *d_scan = (uint32_t *)dalloc(scan_words * 4);
    if (failed) return bail("lds codegen: out of device memory (keys)");
    T.keys = keys_a;
    T.vals = vals_a;
    hipLaunchKernelGGL((k_cg_rows<1>), dim3(row_blocks), dim3(256), 0, st, P, T);
    cg_radix_sort(&keys_a, &keys_b, payload ? &vals_a : nullptr, payload ? &vals_b : nullptr, nnz, cg_key_bits(P), d_hist, d_scan, st);
    T.keys = keys_a;
    T.vals = vals_a;
    if (!ok(hipGetLastError())) return bail("lds codegen: sort launch");
    release(keys_b);
    release(vals_b);
    release(d_hist);
    lap();
    // D3 - D5: staged columns
    const unsigned ent_blocks = (unsigned)((nnz + 255) / 256);
    uint32_t *d_colflag = (uint32_t *)dalloc((nnz + 1) * 4), *d_colx = (uint32_t *)dalloc((nnz + 1) * 4), *d_word = (uint32_t *)dalloc(64);
    if (failed) return bail("lds codegen: out of device memory (column flags)");
    T.colflag = d_colflag;
    T.colx = d_colx;
    hipLaunchKernelGGL((k_cg_entries<0>), dim3(ent_blocks), dim3(256), 0, st, P, T);
    cg_scan_exclusive(d_colflag, d_colx, nnz, d_scan, st);
    hipLaunchKernelGGL(k_cg_tail2, dim3(1), dim3(64), 0, st, (const uint32_t *)d_colflag, (const uint32_t *)d_colx, nnz, d_word);
    uint32_t ncols_total = 0;
    if (!ok(hipMemcpyAsync(&ncols_total, d_word, 4, hipMemcpyDeviceToHost, st)) || !ok(hipStreamSynchronize(st))) return bail("lds codegen: column count");
    T.ncols_total = ncols_total;
    uint32_t *d_col_first = (uint32_t *)dalloc(((size_t)ncols_total + 3) * 4), *d_col_g = (uint32_t *)dalloc(((size_t)ncols_total + 1) * 4);
    uint16_t *d_col_lrow = (uint16_t *)dalloc(((size_t)ncols_total + 2) * 2);
    if (failed) return bail("lds codegen: out of device memory (columns)");
    T.col_first = d_col_first; T.col_g = d_col_g; T.col_lrow = d_col_lrow;
    hipLaunchKernelGGL(k_cg_fill_words, dim3(1), dim3(256), 0, st, d_col_first + ncols_total, (uint64_t)3, (uint32_t)nnz);   // sentinels
    hipLaunchKernelGGL((k_cg_entries<1>), dim3(ent_blocks), dim3(256), 0, st, P, T);
    lap();
    // D6 - D8: slots and groups
    const unsigned sj_blocks = (unsigned)(((uint64_t)T.nsj + 1 + 255) / 256);
    uint32_t *d_sfc = (uint32_t *)dalloc(((size_t)T.nsj + 2) * 4), *d_sng = (uint32_t *)dalloc(((size_t)T.nsj + 2) * 4), *d_sfg = (uint32_t *)dalloc(((size_t)T.nsj + 2) * 4);
    unsigned long long *d_stats = (unsigned long long *)dalloc(64);
    if (failed) return bail("lds codegen: out of device memory (slots)");
    T.slot_firstcol = d_sfc; T.slot_ng = d_sng; T.slot_firstgroup = d_sfg; T.stats = d_stats;
    (void)hipMemsetAsync(d_stats, 0, 64, st);
    hipLaunchKernelGGL((k_cg_slots<0>), dim3(sj_blocks), dim3(256), 0, st, P, T);
    hipLaunchKernelGGL((k_cg_slots<1>), dim3(sj_blocks), dim3(256), 0, st, P, T);
    cg_scan_exclusive(d_sng, d_sfg, (uint64_t)T.nsj + 1, d_scan, st);
    uint32_t ngroups = 0;
    if (!ok(hipMemcpyAsync(&ngroups, d_sfg + T.nsj, 4, hipMemcpyDeviceToHost, st)) || !ok(hipStreamSynchronize(st))) return bail("lds codegen: group count");
    if (ngroups >= (1u << 28)) return bail("lds codegen: 2^28 and more groups");
    T.ngroups = ngroups;
    uint32_t *d_gn = (uint32_t *)dalloc(((size_t)ngroups + 1) * 4), *d_gf = (uint32_t *)dalloc(((size_t)ngroups + 1) * 4), *d_gc = (uint32_t *)dalloc(((size_t)ngroups + 1) * 4);
    uint32_t *d_gr = (uint32_t *)dalloc(((size_t)ngroups + 1) * 4), *d_ga = (uint32_t *)dalloc(((size_t)ngroups + 1) * 4);
    uint8_t *d_g8 = (uint8_t *)dalloc(((size_t)ngroups + 1) * 3);
    uint32_t *d_gn0 = P.half_H ? (uint32_t *)dalloc(((size_t)ngroups + 1) * 4) : nullptr;
    if (failed) return bail("lds codegen: out of device memory (groups)");
    T.g_n0 = d_gn0;
    T.g_nent = d_gn; T.g_first = d_gf; T.g_firstcol = d_gc; T.g_rpos = d_gr; T.g_apos = d_ga;
    T.g_nlds = d_g8; T.g_ncols = d_g8 + ((size_t)ngroups + 1); T.g_xset = d_g8 + 2 * ((size_t)ngroups + 1);
    if (T.nsj) hipLaunchKernelGGL((k_cg_slots<2>), dim3((unsigned)(((uint64_t)T.nsj + 255) / 256)), dim3(256), 0, st, P, T);
    lap();
    // D9: stream sizes -> offsets (host: nstreams numbers) -> executable memory
    uint32_t *d_sdw = (uint32_t *)dalloc(((size_t)P.nstreams + 1) * 4);
    if (failed) return bail("lds codegen: out of device memory (streams)");
    T.stream_dw = d_sdw;
    hipLaunchKernelGGL(k_cg_streams, dim3(P.nstreams), dim3(64), 0, st, P, T, 0);
    std::vector<uint32_t> sdw(P.nstreams);
    unsigned long long h_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!ok(hipMemcpyAsync(sdw.data(), d_sdw, (size_t)P.nstreams * 4, hipMemcpyDeviceToHost, st)) ||
        !ok(hipMemcpyAsync(h_stats, d_stats, 64, hipMemcpyDeviceToHost, st)) || !ok(hipStreamSynchronize(st)))
        return bail("lds codegen: the size pass failed");
    std::vector<uint64_t> start(P.nstreams);
    uint64_t total = 0;
    for (uint32_t s = 0; s < P.nstreams; s++) {
        start[s] = total * 4;
        total += sdw[s];
    }
    if (total + 8192 >= (1ull << 32)) return bail("lds codegen: 16 GiB of code and more");
    out.code_bytes = (size_t)(total + 8192) * 4;
    out.code = alloc_exec(out.code_bytes);
    if (!out.code) return bail("lds codegen: no executable memory for the code");
    uint64_t *d_start = nullptr;
    if (!ok(hipMalloc((void **)&d_start, std::max<size_t>((size_t)P.nstreams * 8, 256)))) return bail("lds codegen: out of device memory (stream offsets)");
    keep[0] = d_start;
    if (!ok(hipMemcpyAsync(d_start, start.data(), (size_t)P.nstreams * 8, hipMemcpyHostToDevice, st))) return bail("lds codegen: stream offset upload");
    T.start = d_start;
    T.code = (uint32_t *)out.code;
    lap();
    // D9 (writing) + D10
    hipLaunchKernelGGL(k_cg_fill_words, dim3(32), dim3(256), 0, st, T.code + total, (uint64_t)8192, 0xBF800000u);   // s_nop behind the last stream: the touches read ahead
    hipLaunchKernelGGL(k_cg_streams, dim3(P.nstreams), dim3(64), 0, st, P, T, 1);
    if (ngroups) hipLaunchKernelGGL(k_cg_reads, dim3((ngroups + 255) / 256), dim3(256), 0, st, P, T);
    hipLaunchKernelGGL((k_cg_entries<2>), dim3(ent_blocks), dim3(256), 0, st, P, T);
    // the tables the product kernels read
    const uint32_t KAS = geo.ka_stride();
    std::vector<LdsTile> tiles(P.ntiles);
    for (uint32_t ti = 0; ti < P.ntiles; ti++) {
        tiles[ti] = LdsTile();
        tiles[ti].nch = ch.nch[ti];
        tiles[ti].chunk0 = ch.nch[ti] ? ch.chunks[ch.choff[ti]] : 0;
        tiles[ti].row0 = rows.tile_row0[ti];
        tiles[ti].nnz = rows.tile_nnz[ti];
    }
    uint32_t *d_rowmap = nullptr;
    LdsTile *d_tiles = nullptr;
    if (!ok(hipMalloc((void **)&d_rowmap, std::max<size_t>(rows.rowmap.size() * 4, 256)))) return bail("lds codegen: out of device memory (row map)");
    keep[1] = d_rowmap;
    if (!ok(hipMalloc((void **)&d_tiles, std::max<size_t>(tiles.size() * sizeof(LdsTile), 256)))) return bail("lds codegen: out of device memory (tile table)");
    keep[2] = d_tiles;
    (void)KAS;
    if (!ok(hipMemcpyAsync(d_rowmap, rows.rowmap.data(), rows.rowmap.size() * 4, hipMemcpyHostToDevice, st)) ||
        !ok(hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(LdsTile), hipMemcpyHostToDevice, st)) || !ok(hipStreamSynchronize(st)) ||
        !ok(hipGetLastError()))
        return bail("lds codegen: emission failed");
    lap();
    for (void *p : tmp)
        if (p) (void)hipFree(p);
    out.d_start = d_start;
    out.d_rowmap = d_rowmap;
    out.d_tiles = d_tiles;
    out.ntiles = P.ntiles;
    out.slots = ch.slots;
    out.entries = nnz;
    out.pairs = h_stats[0];
    out.shared = nnz - ncols_total;
    out.regs = lds_code_regs(geo.NW, gsize, nsets, geo.row_bytes == 512);
    out.ms[7] = now_ms() - t0;
    return std::string();
}

}  // namespace pygim
