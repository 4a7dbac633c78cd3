/*
 * oracle/spmm_oracle.c -- CPU restatement of PyGim's SpMM / SpMV aggregation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pygim_amd/ may link, import or call
 * this file; it is the checker used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY STATUS: PINNED.  The reference holds no golden vectors / known-answer tests
 * for this path and its version=cpu path lives in the un-vendored,
 * version-unpinned third-party package torch_sparse (Libs/install_libs.sh:13).  What the
 * reference does hold is its own host oracle: spmm_host_coo
 * (spmm_default/spmm_mul_coo.c:40-51), spmm_host_csr (valued: spmm_grande/spmm_mul_csr.c:119-136;
 * unit weights: spmm_default/spmm_mul_csr.c:100-113), spmm_host (spmv_sparseP/spmv_mul_coo.c:92-103),
 * the merge helpers (spmm_default/spmm_mul_csr.c:41-86) and the group drivers
 * (spmm_default/ops.hpp:42-62,97-118; spmv_sparseP/spmv_mul_coo.c:128-148).  The FILES that hold
 * them include the UPMEM SDK's dpu.h (absent here) for their device code, but those FUNCTIONS need
 * only the reference's own support/common.h + support/matrix.h: oracle/build_ref_host.sh cuts them
 * out of the reference files BY NAME at build time, pipes them to gcc/g++ together with the
 * reference's headers (read in place; no stand-in header, nothing copied into the repo) and
 * builds oracle/_ref/libref_host_<variant>_<DTYPE>.so for the six val_dt types.
 * tests/test_oracle.py compares every arithmetic entry point below with those libraries BYTE FOR
 * BYTE (floats included, INT8/INT16 wrap included) on random valued / overflow cases and on every
 * golden vector; tests/golden/make_golden.py accepts a vector only when the reference build
 * produced the same bytes and stamps `pinned_by` into the fixture.
 * For the record, the absent third-party dependency: torch_sparse (rusty1s/pytorch_sparse), whatever wheel
 * `pip install ... -f https://data.pyg.org/whl/torch-1.13.1+cpu.html` resolved to (Libs/install_libs.sh:13,
 * no version given; the 0.6.x line for torch 1.13).  Its published algorithm for the call the
 * driver makes -- torch_sparse.matmul(adj_t, x), reduce = "sum" (spmm_test.py:25) -- is
 * csrc/cpu/spmm_cpu.cpp: rows in parallel, and per row `for e in rowptr[r] .. rowptr[r+1]:
 * out[r, k] += value[e] * mat[col[e], k]` in the element type of `mat`: the stored-order row sum
 * that oracle_spmm_csr_* below (and the reference's own spmm_host_csr) computes.
 * tests/golden/make_golden.py also checks the vectors against the real package whenever it can be imported.
 * PINNED as well: the partition functions (oracle_partition_*) against the reference's own
 * support/partition.c, and the MatrixMarket reader against utils.hpp (oracle/Makefile targets `ref`, `ref_utils`).
 *
 * Element type: the reference builds one library per val_dt
 * (spmm_default/support/common.h:39-60); here every function is instantiated for
 * the six types with a suffix: i8 i16 i32 i64 f32 f64.  Integer products and sums
 * are two's-complement modular at the width of val_dt, exactly what C's
 * promote-then-truncate `y += v * x` does in the reference.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* One instantiation per element type. */
#define ORACLE_FOR_EACH_TYPE(X) \
    X(i8, int8_t)               \
    X(i16, int16_t)             \
    X(i32, int32_t)             \
    X(i64, int64_t)             \
    X(f32, float)               \
    X(f64, double)

/*
 * CSR product, reference loop order.
 * Follows spmm_grande/spmm_mul_csr.c:119-136 (row -> feature -> stored entry,
 * `y[row*ncols+k] += value * x[col*ncols_pad+k]`), which is the valued form of
 * spmm_default/spmm_mul_csr.c:100-113 (that one drops `value`: its callers only
 * pass all-ones adjacency).  values == NULL means unit weights
 * (backend_pim/spmm.py:48-49 supplies ones when the SparseTensor has no value).
 * Accumulates INTO y (caller zero-fills), like the reference.
 */
#define DEF_SPMM_CSR(S, T)                                                              \
    void oracle_spmm_csr_##S(T *y, uint32_t nrows, const uint32_t *rowptr,              \
                             const uint32_t *colind, const T *values, const T *x,       \
                             uint32_t ncols, uint32_t ldx) {                            \
        for (uint32_t r = 0; r < nrows; r++)                                            \
            for (uint32_t k = 0; k < ncols; k++)                                        \
                for (uint32_t e = rowptr[r]; e < rowptr[r + 1]; e++) {                  \
                    T v = values ? values[e] : (T)1;                                    \
                    y[(size_t)r * ncols + k] += v * x[(size_t)colind[e] * ldx + k];     \
                }                                                                       \
    }

/*
 * COO product, reference loop order.
 * Follows spmm_default/spmm_mul_coo.c:40-51 and spmv_sparseP/spmv_mul_coo.c:92-103
 * (stored entry -> feature, `y[row*ncols+k] += x[col*ncols+k] * value`).
 */
#define DEF_SPMM_COO(S, T)                                                              \
    void oracle_spmm_coo_##S(T *y, uint32_t nnz, const uint32_t *rowind,                \
                             const uint32_t *colind, const T *values, const T *x,       \
                             uint32_t ncols) {                                          \
        for (uint32_t n = 0; n < nnz; n++) {                                            \
            T v = values ? values[n] : (T)1;                                            \
            for (uint32_t k = 0; k < ncols; k++)                                        \
                y[(size_t)rowind[n] * ncols + k] += x[(size_t)colind[n] * ncols + k] * v; \
        }                                                                               \
    }

/*
 * Row-parallel CSR product (entry -> feature inside a row, rows over OpenMP
 * threads).  Same per-(row,k) summation order as the two functions above, so the
 * result is bit-identical; this is the shape of the reference's version=cpu path
 * (torch_sparse.matmul, call site spmm_test.py:25 -- row-parallel loop with a
 * per-row accumulator, recalled from upstream csrc/cpu/spmm_cpu.cpp, source not
 * in /root/reference) and is what bench.py times as cpu_baseline (kind "port").
 */
#define DEF_SPMM_CSR_ROWPAR(S, T)                                                       \
    void oracle_spmm_csr_rowpar_##S(T *y, uint32_t nrows, const uint32_t *rowptr,       \
                                    const uint32_t *colind, const T *values,            \
                                    const T *x, uint32_t ncols, int nthreads) {         \
        _Pragma("omp parallel for schedule(dynamic, 64) num_threads(nthreads)")         \
        for (int64_t r = 0; r < (int64_t)nrows; r++) {                                  \
            T *yr = y + (size_t)r * ncols;                                              \
            for (uint32_t e = rowptr[r]; e < rowptr[r + 1]; e++) {                      \
                T v = values ? values[e] : (T)1;                                        \
                const T *xr = x + (size_t)colind[e] * ncols;                            \
                for (uint32_t k = 0; k < ncols; k++) yr[k] += v * xr[k];                \
            }                                                                           \
        }                                                                               \
    }

/*
 * Window accumulate: A[off_x+i, off_y+j] += B[i, j].
 * Follows add_2D, spmm_default/spmm_mul_csr.c:41-50.
 */
#define DEF_ADD_2D(S, T)                                                                \
    void oracle_add_2d_##S(T *A, const T *B, uint32_t A_ncols, uint32_t B_ncols,        \
                           uint32_t off_x, uint32_t off_y, uint32_t len_x,              \
                           uint32_t len_y) {                                            \
        for (uint32_t i = 0; i < len_x; i++)                                            \
            for (uint32_t j = 0; j < len_y; j++)                                        \
                A[(size_t)(off_x + i) * A_ncols + off_y + j] += B[(size_t)i * B_ncols + j]; \
    }

/*
 * Group product: sparse parts are column blocks of A (local column ids), dense
 * parts are feature blocks of X, each dense part a separate row-major array of
 * all total_cols rows.  out[total_rows, h] accumulates every (i, j) block:
 * rows of dense part j used by sparse part i start at the running sum of the
 * previous parts' widths; its output columns start at the running sum of the
 * previous dense widths.
 * Follows spmm_host_csr_group / spmm_host_coo_group, spmm_default/ops.hpp:42-62
 * and :97-118 (and spmv_sparseP/spmv_mul_coo.c:125-148).
 * is_coo selects which index array `ptr_or_row[i]` is (rowptr or row indices).
 */
#define DEF_GROUP(S, T)                                                                 \
    void oracle_group_##S(T *out, int is_coo, uint32_t n_parts,                         \
                          const uint32_t *const *ptr_or_row,                            \
                          const uint32_t *const *colind, const T *const *values,        \
                          const uint32_t *nrows, const uint32_t *ncols,                 \
                          const uint32_t *nnz, uint32_t dense_parts,                    \
                          const T *const *x_parts, const uint32_t *dense_ncols,         \
                          uint32_t h) {                                                 \
        uint32_t brow = 0;                                                              \
        for (uint32_t i = 0; i < n_parts; i++) {                                        \
            uint32_t acol = 0;                                                          \
            for (uint32_t j = 0; j < dense_parts; j++) {                                \
                uint32_t hj = dense_ncols[j];                                           \
                T *tmp = (T *)calloc((size_t)nrows[i] * hj + 1, sizeof(T));             \
                const T *xj = x_parts[j] + (size_t)brow * hj;                           \
                if (is_coo)                                                             \
                    oracle_spmm_coo_##S(tmp, nnz[i], ptr_or_row[i], colind[i],          \
                                        values ? values[i] : NULL, xj, hj);             \
                else                                                                    \
                    oracle_spmm_csr_##S(tmp, nrows[i], ptr_or_row[i], colind[i],        \
                                        values ? values[i] : NULL, xj, hj, hj);         \
                oracle_add_2d_##S(out, tmp, h, hj, 0, acol, nrows[i], hj);              \
                acol += hj;                                                             \
                free(tmp);                                                              \
            }                                                                           \
            brow += ncols[i];                                                           \
        }                                                                               \
    }

#define DEF_ALL(S, T) DEF_SPMM_CSR(S, T) DEF_SPMM_COO(S, T) DEF_SPMM_CSR_ROWPAR(S, T) DEF_ADD_2D(S, T) DEF_GROUP(S, T)
ORACLE_FOR_EACH_TYPE(DEF_ALL)

/* ------------------------------------------------------------------------- */
/* Partitioning (pinned against oracle/_ref built from support/partition.c)    */
/* ------------------------------------------------------------------------- */

/*
 * Equal row counts per part, remainder spread over the first parts.
 * Follows partition_by_row_csr, spmm_default/support/partition.c:14-46.
 */
void oracle_partition_by_row(uint32_t nrows, uint32_t *split, int nparts) {
    if (nparts == 1) {
        split[0] = 0;
        split[1] = nrows;
        return;
    }
    uint32_t base = nrows / (uint32_t)nparts, rest = nrows % (uint32_t)nparts, cur = 0;
    split[0] = 0;
    for (int p = 0; p < nparts; p++) {
        cur += base + ((uint32_t)p < rest ? 1u : 0u);
        if (cur > nrows) cur = nrows;
        split[p + 1] = cur;
    }
}

/*
 * Greedy nnz balance at row granularity: close a part as soon as its running
 * nnz reaches floor(nnz / nparts); leftovers merge into / pad the tail.
 * Follows partition_by_nnz_csr, spmm_default/support/partition.c:51-99
 * (partition_by_nnz_rgrn_coo :106-147 is the same walk over a row histogram).
 * `rownnz(r)` is rowptr[r+1]-rowptr[r].
 */
void oracle_partition_by_nnz(uint32_t nrows, const uint32_t *rowptr, uint32_t *split, int nparts) {
    if (nparts == 1) {
        split[0] = 0;
        split[1] = nrows;
        return;
    }
    uint32_t total = rowptr[nrows] - rowptr[0];
    uint32_t target = total / (uint32_t)nparts;
    uint32_t run = 0;
    uint32_t closed = 0;
    split[0] = 0;
    for (uint32_t r = 0; r < nrows; r++) {
        run += rowptr[r + 1] - rowptr[r];
        if (run >= target) {
            closed++;
            if (closed <= (uint32_t)nparts) split[closed] = r + 1;
            run = 0;
        }
    }
    if (run < target && closed <= (uint32_t)nparts) {
        /* the reference stores at [++split_cnt] even when that is nparts+1 (one past
         * the array its callers allocate); the next statement makes that slot moot */
        closed++;
        if (closed <= (uint32_t)nparts) split[closed] = nrows;
    }
    if (closed > (uint32_t)nparts) split[nparts] = nrows;
    for (uint32_t p = closed + 1; p <= (uint32_t)nparts; p++) split[p] = nrows;
}

/*
 * Equal-nnz split of a contiguous nnz range (rows may straddle parts).
 * Follows partition_tsklt_by_nnz_coo, spmm_default/support/partition.c:231-262.
 */
void oracle_partition_equal_nnz(uint32_t nnz, uint32_t *split, int nparts) {
    uint32_t base = nnz / (uint32_t)nparts, rest = nnz % (uint32_t)nparts;
    split[0] = 0;
    for (int p = 0; p < nparts; p++) split[p + 1] = split[p] + base + ((uint32_t)p < rest ? 1u : 0u);
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
