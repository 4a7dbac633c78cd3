// Round 5: the DENSITY SPLIT of a community-structured matrix whose node ids carry no locality (a dataset with arbitrary ids).
//
// The LDS-staged product pays 32 KiB of staging per (tile of rows, chunk of 128 columns, slice) whatever the chunk holds for the tile; the L2 sweep pays
// one gathered row of X per stored entry.  With the rows AND the columns taken in the order label propagation found (lds_reorder_dev.hpp) the entries inside
// a community fall into a few (tile, chunk) cells with thousands of entries each, the rest is spread one entry to a cell.  So the matrix is split by cell:
//     A = A_dense (cells of at least `min_cell` entries; column ids renamed to positions of that order, sorted inside each row)  -> LDS-staged product over a
//                  copy of X whose rows are in that order (k_slice_pack with an index), tiles skip every chunk they have no entry in
//       + A_sparse (all other entries, ids and stored order untouched)                                                          -> the L2 sweep, adding into C
// The reference has no counterpart (it deals consecutive rows to DPUs, support/partition.c:51-99, and gathers from MRAM); the split changes the ORDER in which
// a row's products are summed, so integers stay exact and floats stay inside the 1e-5 the north star asks for, not bit-identical: off for floats unless asked.
//
// Everything here runs on the device from the resident CSR; the scan and the radix sort are lds_codegen_dev.hpp's.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "lds_codegen_dev.hpp"

namespace pygim {

__global__ __launch_bounds__(256) void k_hy_pos(const uint32_t *order, uint32_t n, uint32_t *pos) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) pos[order[i]] = i;
}

// a wave per row, lanes over its entries.  WHAT 0: count the entries of every (row tile, column chunk) cell; 1: flag[e] = the entry's cell holds at least
// min_cell entries; 2: scatter -- dense entries as keys (row << col_bits | position of the column) at dpos[e], the others' column ids at e - dpos[e]
template <int WHAT>
__global__ __launch_bounds__(256) void k_hy_rows(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, const uint32_t *pos, uint32_t rows_per_tile, uint32_t kc,
                                                 uint32_t nchunks, uint32_t *cnt, uint32_t min_cell, uint32_t *flag, const uint32_t *dpos, uint32_t col_bits,
                                                 uint64_t *keys, uint32_t *col_s) {
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const uint32_t e0 = rowptr[row], e1 = rowptr[row + 1], lane = threadIdx.x & 63;
    const uint64_t cell0 = (uint64_t)(pos[row] / rows_per_tile) * nchunks;
    for (uint32_t e = e0 + lane; e < e1; e += 64) {
        const uint32_t c = col[e], pc = pos[c];
        if (WHAT == 0) atomicAdd(&cnt[cell0 + pc / kc], 1u);
        else if (WHAT == 1) flag[e] = cnt[cell0 + pc / kc] >= min_cell ? 1u : 0u;
        else if (flag[e]) keys[dpos[e]] = ((uint64_t)row << col_bits) | pc;
        else col_s[e - dpos[e]] = c;
    }
}

__global__ __launch_bounds__(256) void k_hy_rowptr(const uint32_t *rowptr, uint32_t nrows, const uint32_t *dpos, uint32_t *rowptr_d, uint32_t *rowptr_s) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r > nrows) return;
    const uint32_t e = rowptr[r], d = dpos[e];   // (dpos has nnz + 1 elements: dpos[nnz] = dense entries in all)
    rowptr_d[r] = d;
    rowptr_s[r] = e - d;
}

__global__ __launch_bounds__(256) void k_hy_unpack(const uint64_t *keys, uint64_t n, uint32_t col_bits, uint32_t *col_d) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) col_d[i] = (uint32_t)(keys[i] & ((1ull << col_bits) - 1));
}

struct HybridSplit {
    uint32_t *d_rowptr_d = nullptr, *d_col_d = nullptr;   // A_dense: columns = positions in the order, sorted inside each row
    uint32_t *d_rowptr_s = nullptr, *d_col_s = nullptr;   // A_sparse: stored ids, stored order
    uint32_t *d_order = nullptr;                          // order[position] = node: row `position` of the staged copy holds X[order[position]]
    uint64_t nnz_d = 0, nnz_s = 0;
    void release() {
        for (uint32_t *q : {d_rowptr_d, d_col_d, d_rowptr_s, d_col_s, d_order})
            if (q) (void)hipFree(q);
        d_rowptr_d = d_col_d = d_rowptr_s = d_col_s = d_order = nullptr;
    }
};

// h_order: the rows (= columns: a square part) in similarity order.  Returns "" or why not (nothing is left allocated then).
inline std::string hybrid_split_on_device(const uint32_t *d_rowptr, const uint32_t *d_col, uint32_t n, uint64_t nnz, const std::vector<uint32_t> &h_order, uint32_t rows_per_tile,
                                          uint32_t kc, uint32_t min_cell, hipStream_t st, HybridSplit &out) {
    if (h_order.size() != n || n == 0 || nnz == 0 || nnz >= (1ull << 31)) return "hybrid: no order, no entries, or 2^31 and more";
    const uint32_t nchunks = (n + kc - 1) / kc, ntiles = (n + rows_per_tile - 1) / rows_per_tile;
    const uint64_t ncells = (uint64_t)ntiles * nchunks;
    if (ncells >= (1ull << 29)) return "hybrid: more than 2^29 (tile, chunk) cells";
    uint32_t col_bits = 1;
    while ((1ull << col_bits) < (uint64_t)n) col_bits++;
    std::vector<void *> tmp;
    bool failed = false;
    auto dalloc = [&](size_t bytes) -> void * {
        void *p = nullptr;
        if (failed || hipMalloc(&p, std::max<size_t>(bytes, 256)) != hipSuccess) {
            (void)hipGetLastError();
            failed = true;
            return nullptr;
        }
        tmp.push_back(p);
        return p;
    };
    auto drop = [&]() {
        for (void *q : tmp) (void)hipFree(q);
        tmp.clear();
    };
    uint32_t *d_pos = (uint32_t *)dalloc((size_t)n * 4);
    uint32_t *d_cnt = (uint32_t *)dalloc((size_t)ncells * 4);
    uint32_t *d_flag = (uint32_t *)dalloc((size_t)(nnz + 1) * 4);
    uint32_t *d_dpos = (uint32_t *)dalloc((size_t)(nnz + 1) * 4);
    uint32_t *d_scan = (uint32_t *)dalloc((size_t)cg_scan_scratch_words(nnz + 1) * 4);
    if (hipMalloc((void **)&out.d_order, (size_t)n * 4) != hipSuccess) failed = true;
    if (failed) { drop(); out.release(); (void)hipGetLastError(); return "hybrid: out of device memory"; }
    const unsigned row_blocks = (unsigned)(((uint64_t)n + 3) / 4);
    bool ok = hipMemcpyAsync(out.d_order, h_order.data(), (size_t)n * 4, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemsetAsync(d_cnt, 0, (size_t)ncells * 4, st) == hipSuccess && hipMemsetAsync(d_flag + nnz, 0, 4, st) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_hy_pos, dim3((n + 255) / 256), dim3(256), 0, st, (const uint32_t *)out.d_order, n, d_pos);
        hipLaunchKernelGGL((k_hy_rows<0>), dim3(row_blocks), dim3(256), 0, st, d_rowptr, d_col, n, (const uint32_t *)d_pos, rows_per_tile, kc, nchunks, d_cnt, min_cell,
                           (uint32_t *)nullptr, (const uint32_t *)nullptr, col_bits, (uint64_t *)nullptr, (uint32_t *)nullptr);
        hipLaunchKernelGGL((k_hy_rows<1>), dim3(row_blocks), dim3(256), 0, st, d_rowptr, d_col, n, (const uint32_t *)d_pos, rows_per_tile, kc, nchunks, d_cnt, min_cell, d_flag,
                           (const uint32_t *)nullptr, col_bits, (uint64_t *)nullptr, (uint32_t *)nullptr);
        cg_scan_exclusive(d_flag, d_dpos, nnz + 1, d_scan, st);
        uint32_t total = 0;
        ok = hipMemcpyAsync(&total, d_dpos + nnz, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
        out.nnz_d = total;
        out.nnz_s = nnz - total;
    }
    if (!ok) { drop(); out.release(); (void)hipGetLastError(); return "hybrid: counting the cells failed"; }
    if (out.nnz_d == 0 || out.nnz_s == 0) { drop(); out.release(); return out.nnz_d == 0 ? "hybrid: no cell is dense" : "hybrid: every cell is dense"; }
    uint64_t *d_keys_a = (uint64_t *)dalloc((size_t)out.nnz_d * 8), *d_keys_b = (uint64_t *)dalloc((size_t)out.nnz_d * 8);
    const uint32_t stiles = cg_sort_tiles(out.nnz_d);
    uint32_t *d_hist = (uint32_t *)dalloc(((size_t)256 * stiles + 1) * 4);
    uint32_t *d_scan2 = (uint32_t *)dalloc((size_t)cg_scan_scratch_words((uint64_t)256 * stiles) * 4);
    if (hipMalloc((void **)&out.d_rowptr_d, ((size_t)n + 1) * 4) != hipSuccess || hipMalloc((void **)&out.d_rowptr_s, ((size_t)n + 1) * 4) != hipSuccess ||
        hipMalloc((void **)&out.d_col_d, (size_t)out.nnz_d * 4) != hipSuccess || hipMalloc((void **)&out.d_col_s, (size_t)out.nnz_s * 4) != hipSuccess)
        failed = true;
    if (failed) { drop(); out.release(); (void)hipGetLastError(); return "hybrid: out of device memory"; }
    hipLaunchKernelGGL((k_hy_rows<2>), dim3(row_blocks), dim3(256), 0, st, d_rowptr, d_col, n, (const uint32_t *)d_pos, rows_per_tile, kc, nchunks, d_cnt, min_cell, d_flag,
                       (const uint32_t *)d_dpos, col_bits, d_keys_a, out.d_col_s);
    hipLaunchKernelGGL(k_hy_rowptr, dim3((n + 1 + 255) / 256), dim3(256), 0, st, d_rowptr, n, (const uint32_t *)d_dpos, out.d_rowptr_d, out.d_rowptr_s);
    // rows are in place already (the scatter keeps the stored order): sorting by the whole key orders the columns inside each row
    uint32_t row_bits = 1;
    while ((1ull << row_bits) < (uint64_t)n) row_bits++;
    uint32_t *no_vals = nullptr;
    cg_radix_sort(&d_keys_a, &d_keys_b, &no_vals, &no_vals, out.nnz_d, col_bits + row_bits, d_hist, d_scan2, st);
    hipLaunchKernelGGL(k_hy_unpack, dim3((unsigned)((out.nnz_d + 255) / 256)), dim3(256), 0, st, (const uint64_t *)d_keys_a, out.nnz_d, col_bits, out.d_col_d);
    ok = hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
    drop();
    if (!ok) { out.release(); (void)hipGetLastError(); return "hybrid: the split failed on the device"; }
    return std::string();
}

}  // namespace pygim
