// lds_reorder_dev.hpp -- SIMILARITY TILES (round 5): which rows share a tile of the LDS-staged product.
//
// The reference hands each DPU a range of CONSECUTIVE rows (support/partition.c:51-99).  Here a tile's rows need not be consecutive -- the
// store stage writes every row by index (the row map) and every row is still summed by one wave in stored order, so the result does not
// depend on the assignment -- and rows with similar neighbourhoods side by side pay off twice: entries of one wave that hit the same column
// share ONE LDS read, and a tile whose rows live in one community skips the chunks of X it never touches.  Node ids of real datasets
// carry no such locality in general (pygim_amd/synth.py: SBM / R-MAT stand-ins with shuffled ids), so the order is found from the graph:
// label propagation with hashed votes -- seed: the min-hash of a row's column ids; a few rounds of "adopt the label most of your
// neighbours hold" (a wave per row, 512 vote buckets in LDS, ties broken by a per-round hash: deterministic) -- then rows sorted by
// (label, id).  Square matrices only (a column's label is its row's).  On a graph without structure the labels collapse to a few and
// the order degenerates to the consecutive one: nothing is lost.
#pragma once
#include <hip/hip_runtime.h>

#include "lds_codegen_dev.hpp"

namespace pygim {

__device__ inline uint32_t lp_hash(uint32_t x, uint32_t salt) {
    const uint64_t v = ((uint64_t)x + salt) * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(v >> 29);
}
constexpr uint32_t LP_BUCKETS = 512;

// seed: min-hash of the row's column ids (rows that share many neighbours tend to share it); empty rows: a hash of their id
__global__ __launch_bounds__(256) void k_lp_seed(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t *lab) {
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const uint32_t lane = threadIdx.x & 63, e0 = rowptr[row], e1 = rowptr[row + 1];
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t e = e0 + lane; e < e1; e += 64) m = min(m, lp_hash(col[e], 0));
    for (int d = 32; d; d >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0) lab[row] = e1 > e0 ? m : lp_hash((uint32_t)row, 0);
}
// one round: the label most of the row's neighbours hold (votes counted in hashed buckets; a bucket answers with the largest label it saw)
__global__ __launch_bounds__(256) void k_lp_iter(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, const uint32_t *lab_in, uint32_t *lab_out,
                                                 uint32_t round) {
    __shared__ uint32_t cnt[4][LP_BUCKETS], rep[4][LP_BUCKETS];
    __shared__ unsigned long long best[4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + wave;
    for (uint32_t b = lane; b < LP_BUCKETS; b += 64) { cnt[wave][b] = 0; rep[wave][b] = 0; }
    if (lane == 0) best[wave] = 0;
    __syncthreads();
    if (row < nrows) {
        const uint32_t e0 = rowptr[row], e1 = rowptr[row + 1];
        for (uint32_t e = e0 + lane; e < e1; e += 64) {
            const uint32_t l = lab_in[col[e]], b = lp_hash(l, 17) % LP_BUCKETS;
            atomicAdd(&cnt[wave][b], 1u);
            atomicMax(&rep[wave][b], l);
        }
    }
    __syncthreads();
    if (row < nrows) {
        unsigned long long mine = 0;
        for (uint32_t b = lane; b < LP_BUCKETS; b += 64) {
            const uint32_t c = cnt[wave][b];
            if (c) mine = max(mine, ((unsigned long long)c << 32) | lp_hash(rep[wave][b], 1000 + round));
        }
        // (score = votes, then a per-round hash of the bucket's label; the winning bucket is found again below by its score)
        atomicMax(&best[wave], mine);
    }
    __syncthreads();
    if (row < nrows) {
        const uint32_t e0 = rowptr[row], e1 = rowptr[row + 1];
        const unsigned long long win = best[wave];
        uint32_t out = 0;
        for (uint32_t b = lane; b < LP_BUCKETS; b += 64) {
            const uint32_t c = cnt[wave][b];
            if (c && (((unsigned long long)c << 32) | lp_hash(rep[wave][b], 1000 + round)) == win) out = max(out, rep[wave][b]);
        }
        for (int d = 32; d; d >>= 1) out = max(out, (uint32_t)__shfl_xor((int)out, d, 64));
        if (lane == 0) lab_out[row] = e1 > e0 ? out : lab_in[row];
    }
}
// how good the labels are: the number of stored entries whose column carries its row's label (of nnz: ~p_in for a planted partition, ~0
// for labels that mean nothing)
__global__ __launch_bounds__(256) void k_lp_agree(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, const uint32_t *lab, unsigned long long *acc) {
    // a wave walks rows wave, wave + (waves of the grid), ...: ONE atomic per wave at the end (one per row on a single address cost 29 ms for 2.4 M rows)
    unsigned long long same = 0;
    for (uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += (uint64_t)gridDim.x * 4) {
        const uint32_t mine = lab[row], e1 = rowptr[row + 1];
        for (uint32_t e = rowptr[row] + (threadIdx.x & 63); e < e1; e += 64) same += lab[col[e]] == mine ? 1u : 0u;
    }
    for (int d = 32; d; d >>= 1) same += __shfl_xor((long long)same, d, 64);
    if ((threadIdx.x & 63) == 0 && same) atomicAdd(acc, same);
}
__global__ __launch_bounds__(256) void k_lp_keys(const uint32_t *lab, uint32_t nrows, uint64_t *keys) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < nrows) keys[r] = ((uint64_t)lab[r] << 32) | r;
}

// how local the stored column ids already are: the mean over non-empty rows of (last column - first column) / ncols (rows are column-sorted).
// ~1 for ids without locality (uniform, shuffled), small when a row's neighbours sit near each other in id order
__global__ __launch_bounds__(256) void k_row_span(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, unsigned long long *acc) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long span = 0, cnt = 0;
    if (r < nrows) {
        const uint32_t e0 = rowptr[r], e1 = rowptr[r + 1];
        if (e1 > e0) { span = col[e1 - 1] - col[e0]; cnt = 1; }
    }
    for (int d = 32; d; d >>= 1) {
        span += __shfl_xor((long long)span, d, 64);
        cnt += __shfl_xor((long long)cnt, d, 64);
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        atomicAdd(&acc[0], span);
        atomicAdd(&acc[1], cnt);
    }
}
inline double lds_mean_row_span(const uint32_t *d_rowptr, const uint32_t *d_col, uint32_t nrows, uint32_t ncols, hipStream_t st) {
    unsigned long long *acc = nullptr, h[2] = {0, 0};
    if (hipMalloc((void **)&acc, 64) != hipSuccess) { (void)hipGetLastError(); return 1.0; }
    (void)hipMemsetAsync(acc, 0, 64, st);
    hipLaunchKernelGGL(k_row_span, dim3((nrows + 255) / 256), dim3(256), 0, st, d_rowptr, d_col, nrows, acc);
    const bool ok = hipMemcpyAsync(h, acc, 16, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    (void)hipFree(acc);
    if (!ok || !h[1] || !ncols) { (void)hipGetLastError(); return 1.0; }
    return (double)h[0] / (double)h[1] / (double)ncols;
}

// rows ordered by (label, id) after `rounds` rounds; "" on success.  labels_out (optional): number of distinct labels, largest label's rows
inline std::string lds_similarity_order(const uint32_t *d_rowptr, const uint32_t *d_col, uint32_t nrows, int rounds, hipStream_t st, std::vector<uint32_t> &rorder,
                                        uint32_t *n_labels = nullptr, uint32_t *largest = nullptr, uint64_t *agree = nullptr) {
    if (nrows == 0) return "no rows";
    uint32_t *lab_a = nullptr, *lab_b = nullptr, *hist = nullptr, *scan = nullptr;
    uint64_t *keys_a = nullptr, *keys_b = nullptr;
    const uint32_t tiles = cg_sort_tiles(nrows);
    auto drop = [&]() {
        for (void *p : {(void *)lab_a, (void *)lab_b, (void *)hist, (void *)scan, (void *)keys_a, (void *)keys_b})
            if (p) (void)hipFree(p);
    };
    if (hipMalloc((void **)&lab_a, (size_t)nrows * 4 + 256) != hipSuccess || hipMalloc((void **)&lab_b, (size_t)nrows * 4 + 256) != hipSuccess ||
        hipMalloc((void **)&keys_a, (size_t)nrows * 8 + 256) != hipSuccess || hipMalloc((void **)&keys_b, (size_t)nrows * 8 + 256) != hipSuccess ||
        hipMalloc((void **)&hist, ((size_t)256 * tiles + 1) * 4) != hipSuccess ||
        hipMalloc((void **)&scan, cg_scan_scratch_words((uint64_t)256 * tiles) * 4) != hipSuccess) {
        (void)hipGetLastError();
        drop();
        return "out of device memory";
    }
    const unsigned rb = (unsigned)(((uint64_t)nrows + 3) / 4);
    hipLaunchKernelGGL(k_lp_seed, dim3(rb), dim3(256), 0, st, d_rowptr, d_col, nrows, lab_a);
    for (int r = 0; r < rounds; r++) {
        hipLaunchKernelGGL(k_lp_iter, dim3(rb), dim3(256), 0, st, d_rowptr, d_col, nrows, (const uint32_t *)lab_a, lab_b, (uint32_t)r);
        std::swap(lab_a, lab_b);
    }
    unsigned long long h_agree = 0;
    if (agree) {   // (keys_b is free until the sort: its first word takes the count)
        (void)hipMemsetAsync(keys_b, 0, 8, st);
        hipLaunchKernelGGL(k_lp_agree, dim3(std::min<unsigned>(rb, 4096u)), dim3(256), 0, st, d_rowptr, d_col, nrows, (const uint32_t *)lab_a, (unsigned long long *)keys_b);
        (void)hipMemcpyAsync(&h_agree, keys_b, 8, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        *agree = h_agree;
    }
    hipLaunchKernelGGL(k_lp_keys, dim3((nrows + 255) / 256), dim3(256), 0, st, (const uint32_t *)lab_a, nrows, keys_a);
    uint32_t bits = 1;
    while (bits < 32 && (1ull << bits) < (uint64_t)nrows) bits++;
    // (all 32 label bits + the row bits; the row ids are ascending already, so the LSD passes over them keep that order: only the label passes matter)
    for (uint32_t shift = 32; shift < 64; shift += 8) {
        hipLaunchKernelGGL(k_cg_radix_hist, dim3((tiles + 3) / 4), dim3(256), 0, st, (const uint64_t *)keys_a, hist, (uint64_t)nrows, tiles, shift);
        cg_scan_exclusive(hist, hist, (uint64_t)256 * tiles, scan, st);
        hipLaunchKernelGGL((k_cg_radix_scatter<false>), dim3((tiles + 3) / 4), dim3(256), 0, st, (const uint64_t *)keys_a, keys_b, (const uint32_t *)nullptr,
                           (uint32_t *)nullptr, (const uint32_t *)hist, (uint64_t)nrows, tiles, shift);
        std::swap(keys_a, keys_b);
    }
    (void)bits;
    std::vector<uint64_t> h((size_t)nrows);
    const bool ok = hipMemcpyAsync(h.data(), keys_a, (size_t)nrows * 8, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
                    hipGetLastError() == hipSuccess;
    drop();
    if (!ok) return "label propagation failed on the device";
    rorder.resize(nrows);
    uint32_t nl = 0, run = 0, big = 0;
    for (uint32_t i = 0; i < nrows; i++) {
        rorder[i] = (uint32_t)h[i];
        if (i == 0 || (h[i] >> 32) != (h[i - 1] >> 32)) { nl++; run = 0; }
        run++;
        big = std::max(big, run);
    }
    if (n_labels) *n_labels = nl;
    if (largest) *largest = big;
    return std::string();
}

}  // namespace pygim
