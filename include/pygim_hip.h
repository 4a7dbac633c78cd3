/*
 * pygim_hip.h -- C ABI of the MI355X (gfx950) aggregation backend.
 *
 * This is the drop-in boundary for PyGim's backend_pim SpMM/SpMV path: every
 * entry point replaces one function the reference reaches through
 * torch.ops.pim_ops.* (registered in backend_pim/<variant>/pytorch_api.cpp).
 * Plain pointers and sizes only; no torch types.  All functions return 0 on
 * success and a non-zero pygim_status otherwise; pygim_last_error() then holds a
 * message (thread-local).
 *
 * Pointer convention: every data pointer may be a HOST pointer or a DEVICE (HIP)
 * pointer; the library asks the HIP runtime which.  Host data is staged through
 * device buffers owned by the group (one-time for the sparse arrays -- the
 * reference's copy_sparse_csr/copy_sparse_coo -- and per call for dense parts).
 * Device data is used in place: the caller keeps it alive for the lifetime of
 * the group, the same ownership rule the reference has for the raw data_ptr()s
 * it stores (spmm_default/pytorch_api.cpp:230-232, 312-314).
 *
 * Threading: like the reference (one process-global `dpus` singleton,
 * spmm_default/pytorch_api.cpp:152) the library holds one device context per
 * process; calls are not re-entrant for the same group handle.  The tunables are
 * process-global.  The slice-major copy of X that the panel sweep gathers from lives in
 * one buffer per (device, launch stream), so products on different streams do not
 * disturb each other; at most four are kept (least recently used goes first) and all
 * are freed with the last group.  A caller that runs several products on the SAME
 * unchanged X says so per call (the x_unchanged argument of pygim_spmm_run_group_x /
 * pygim_block_run_x): the copy made by the most recent product on exactly that operand
 * (pointer, stride, shape) is then shared instead of repeated; the caller orders the
 * streams itself (an event after the first product, as bench.py does) and must not
 * pass the flag after writing to X.
 * Scratch buffers are sized on first use (hipMalloc), so capture a product into a hipGraph only after one warm-up call.
 * A captured graph bakes in the addresses of the slice-major buffer of its capture stream and of the group's scratch
 * buffers: while such a graph may still be replayed, do not run products on four or more OTHER streams of the same
 * device (the least recently used slice-major buffer is freed), do not run a LARGER product on the capture stream
 * (its buffer is re-allocated) and do not free the group -- inference.py --graph 1 keeps to one stream and one shape.
 */
#ifndef PYGIM_HIP_H
#define PYGIM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PYGIM_OK = 0,
    PYGIM_ERR_INVALID = 1,   /* bad argument / shape / dtype / handle          */
    PYGIM_ERR_NO_DEVICE = 2, /* no HIP device, or pygim_init not called         */
    PYGIM_ERR_HIP = 3,       /* a HIP runtime call failed                       */
    PYGIM_ERR_UNSORTED = 4   /* COO input not sorted by row (must be coalesced) */
} pygim_status;

/* val_dt of the reference (spmm_default/support/common.h:39-60), chosen at run
 * time here instead of one library per type. */
typedef enum {
    PYGIM_INT8 = 0,
    PYGIM_INT16 = 1,
    PYGIM_INT32 = 2,
    PYGIM_INT64 = 3,
    PYGIM_FLT32 = 4,
    PYGIM_DBL64 = 5
} pygim_dtype;

typedef enum { PYGIM_CSR = 0, PYGIM_COO = 1 } pygim_format;

/* ---- device set-up --------------------------------------------------------
 * pygim_init_ranks replaces dpu_init_ranks (spmm_default/pytorch_api.cpp:154,
 * spmv_sparseP/pytorch_api.cpp:132; grande's int[]-returning form
 * spmm_grande/pytorch_api.cpp:157-166).  nr_ranks = number of (sparse part x
 * dense part) partitions the caller will create.  If units_per_rank is not NULL
 * it receives nr_ranks entries: the number of feature windows the grande layout
 * should cut per sparse part (grande's "DPUs per rank"; here one window per XCD).
 * pygim_init_units replaces dpu_init_dpus (:158 / grande :168-177).
 * pygim_release replaces dpu_release (:162): frees every live group.          */
int pygim_init_ranks(int64_t nr_ranks, int64_t *units_per_rank);
int pygim_init_units(int64_t nr_units, int64_t *units_per_rank, int64_t *nr_ranks_out);
int pygim_release(void);
/* number of pygim_release calls so far: a handle made under an earlier generation is dead (every group was freed),
 * even when the allocator hands the same address out again -- wrappers that cache a handle compare this. */
int64_t pygim_generation(void);
/* creation serial of a live group (1, 2, 3 ... over the life of the process, never reused): a handle is an address, and the allocator may hand
 * a freed group's address to the next one -- a wrapper that frees what it created (the reference's spmm_free_group,
 * spmm_default/pytorch_api.cpp:198-201, is the caller's to call) remembers the serial and frees only while it still matches. */
int pygim_group_serial(int64_t handle, int64_t *out);
int pygim_is_initialized(void);
const char *pygim_last_error(void);
/* name / CU count / bytes of HBM of the device in use */
int pygim_device_info(char *name, int name_len, int *cu_count, int64_t *hbm_bytes);

/* ---- sparse group: create / free -------------------------------------------
 * Replaces spmm_csr_to_device_group / spmm_coo_to_device_group
 * (spmm_default/pytorch_api.cpp:204-243, 286-329), grande's
 * spmm_csr_to_device_group (spmm_grande/pytorch_api.cpp:221-264) and
 * spmv_coo_to_device_group (spmv_sparseP/pytorch_api.cpp:184-228).
 *
 *  n_parts          sparse column blocks of A (backend_pim/spmm.py:127-136);
 *                   block i is nrows[i] x ncols[i] with LOCAL column ids.
 *  idx0[i]          CSR: rowptr, nrows[i]+1 int32;  COO: row index, nnz[i] int32,
 *                   sorted by (row, col) as torch's coalesce() leaves it.
 *  colind[i]        nnz[i] int32.
 *  values[i]        nnz[i] elements of `dtype`; values == NULL or values[i] ==
 *                   NULL means all-ones (backend_pim/spmm.py:36-37, 48-49).
 *  n_dense[i]       number of dense (feature) parts paired with sparse part i.
 *                   Default/spmv layout: the same list for every part -- pass the
 *                   count in n_dense[0..n_parts) and the widths once per part.
 *  dense_cols       concatenated widths, sum_i n_dense[i] entries; for each part
 *                   they must add up to h_size.
 *  out_handle       opaque handle (the reference returns its csr_info_group*
 *                   as an int64, pytorch_api.cpp:240).
 */
int pygim_group_create(int format, int dtype, int n_parts,
                       const int32_t *const *idx0, const int32_t *const *colind,
                       const void *const *values, const int64_t *nrows, const int64_t *ncols,
                       const int64_t *nnz, const int64_t *n_dense, const int64_t *dense_cols,
                       int64_t h_size, int64_t *out_handle);
/* Replaces spmm_free_group (spmm_default/pytorch_api.cpp:198-201). */
int pygim_group_free(int64_t handle);

/* ---- run -----------------------------------------------------------------
 * pygim_spmm_run_group replaces spmm_csr_run_group / spmm_coo_run_group
 * (spmm_default/pytorch_api.cpp:248-280, 332-367): B_parts[j] is a row-major
 * [total_cols, dense_cols[j]] array holding feature block j of X for ALL sparse
 * parts (sparse part i reads the rows that start at the sum of the previous
 * parts' ncols, spmm_mul_csr.c:356-363); out is row-major [nrows[0], h_size] and
 * is fully overwritten (the reference returns a fresh torch::zeros + merge).
 *
 * pygim_grande_run_group replaces grande's spmm_csr_run_group
 * (spmm_grande/pytorch_api.cpp:269-321): B_windows holds, part after part, one
 * window per dense part of that sparse part, each row-major
 * [ncols[i], window_ld[k]] of which the first dense_cols[k] columns are used
 * (backend_pim/grande.py:12-23 pads windows to 8 bytes).
 *
 * pygim_spmv_run_group replaces spmv_coo_run_group
 * (spmv_sparseP/pytorch_api.cpp:231-266): B_vectors[j] is vector j ([ncols,1]);
 * out is [nrows[0], n_dense] -- one SpMV result per column.
 *
 * stream: a hipStream_t (NULL = the default stream).  With device pointers the
 * call only enqueues work; with host pointers it returns after the result has
 * been copied back.                                                           */
int pygim_spmm_run_group(int64_t handle, const void *const *B_parts, void *out, void *stream);
/* the same with x_unchanged (see "Threading" above): 1 = B_parts hold exactly what the most
 * recent product on the same pointers held, its slice-major copy may be reused.   */
int pygim_spmm_run_group_x(int64_t handle, const void *const *B_parts, void *out, int x_unchanged,
                           void *stream);
int pygim_grande_run_group(int64_t handle, const void *const *B_windows, const int64_t *window_ld,
                           void *out, void *stream);
int pygim_spmv_run_group(int64_t handle, const void *const *B_vectors, void *out, void *stream);

/* One block product on resident data, the building brick of the three calls
 * above and of the multi-GPU layer: C[0:nrows, c_col0 : c_col0+width] (+)=
 * A_part . X[:, 0:width] with row strides ldx / ldc in elements.  X and C must
 * be device pointers.                                                         */
int pygim_block_run(int64_t handle, int part, const void *X, int64_t ldx, void *C, int64_t ldc,
                    int64_t width, int accumulate, void *stream);
int pygim_block_run_x(int64_t handle, int part, const void *X, int64_t ldx, void *C, int64_t ldc,
                      int64_t width, int accumulate, int x_unchanged, void *stream);

/* ---- quantise -> aggregate -> dequantise (the step either side of the product in every conv layer,
 * models/pyg_gcn_conv.py:130-137 with models/quantize.py:20-42), on device in one call:
 *   scale = max|X| * 2 / 2^k (k = 5 / 10 / 20 for INT8 / INT16 / INT32 groups; 20 for FLT32),
 *   X_q = round_half_even(X / scale) in the group's type, out_q = A . X_q, out = float(out_q) * scale.
 * X: float32 [total_cols, h] (row stride ldx), out: float32 [nrows, h]; device pointers.
 * scale_out (device float, may be NULL) receives the scale.                     */
int pygim_quant_spmm_run(int64_t handle, const float *X, int64_t ldx, float *out, float *scale_out, void *stream);
/* the same with a per-column epilogue applied in the last store of every row (the sweep's, or the LDS-staged kernel's):
 *   out[r, f] = col_mul[f] * out[r, f] + col_add[f], then max(., 0) when relu != 0
 * (what follows the aggregation in a GCN layer -- + bias, eval-mode BatchNorm, ReLU, models/models.py:30-38 -- folded into
 * one affine map per feature; col_mul / col_add: device float[h]; both NULL = no epilogue).                              */
int pygim_quant_spmm_run_post(int64_t handle, const float *X, int64_t ldx, float *out, float *scale_out,
                              const float *col_mul, const float *col_add, int relu, void *stream);

/* The same quantiser in its three steps, for callers that exchange the QUANTISED features between them
 * (row-sharded multi-GPU inference: every rank quantises its row block with the global scale, the blocks
 * are all-gathered in the narrow type, each rank aggregates its rows and dequantises them).
 *   absmax_bits: device uint32 holding the bit pattern of max|X| (a non-negative float orders like an
 *   unsigned integer, so ranks combine it with an integer MAX all-reduce); pygim_quant_absmax
 *   max-accumulates into it, the caller zeroes it first.
 *   pygim_quantize: Xq[rows, width] contiguous in `dtype` (INT8/INT16/INT32/FLT32) = round_half_even(X / scale).
 *   pygim_dequantize: out[i] = float(Q[i]) * scale.                                     */
int pygim_quant_absmax(const float *X, int64_t ldx, int64_t rows, int64_t width, uint32_t *absmax_bits, void *stream);
int pygim_quantize(int dtype, const float *X, int64_t ldx, int64_t rows, int64_t width, const uint32_t *absmax_bits,
                   void *Xq, float *scale_out, void *stream);
int pygim_dequantize(int dtype, const void *Q, int64_t n, const uint32_t *absmax_bits, float *out, void *stream);
/* the last two of those steps in one sweep: out = float(A . Xq) * scale, Xq [total_cols, h] already quantised in the
 * group's type (row stride ldx), the dequantisation done by the sweep's last store per row (no integer result matrix). */
int pygim_spmm_run_dequant(int64_t handle, const void *Xq, int64_t ldx, float *out, const uint32_t *absmax_bits, void *stream);

/* ---- introspection -----------------------------------------------------------
 * Milliseconds of the last host-pointer run, in the reference's Timer buckets
 * (support/timer.h; printed as [DATA] lines, spmm_mul_csr.c:563-580):
 * [0] load_dense (H2D)  [1] kernel  [2] retrieve (D2H)  [3] merge (always 0: no
 * host merge here)  [4] one-time sparse upload + analysis of the group.       */
int pygim_group_timers(int64_t handle, double out_ms[5]);
/* shape / plan of a group: total_rows, total_cols, h, n_parts, n_long_rows (rows cut into
 * segments over several waves), all_ones flag, column panels of part 0's L2-blocked plan
 * (0 = no plan), work items of that plan.                                       */
int pygim_group_info(int64_t handle, int64_t out[8]);
/* With tunable "kernel_events" = 1 every block product brackets its dominant kernel (the
 * row-gather kernel, not the long-row tail kernels) with HIP events on the launch stream.
 * This call waits for the pending pairs and returns the accumulated milliseconds and the
 * number of launches since the last reset.                                      */
int pygim_group_kernel_ms(int64_t handle, double *sum_ms, int64_t *count, int reset);
/* the same switch for ONE group (what bench.py uses; the tunable is the process-wide default for A/B scripts) */
int pygim_group_kernel_events(int64_t handle, int on);
/* plan of the matrix the run entry points sweep (the merged matrix of a multi-part group, else part 0):
 * column panels (0 = no panel plan), columns per panel, work items, 16-bit panel-local column ids built (0/1),
 * wave-cooperative (long) items, segment-kernel tasks, merged (0/1), has a non-unit-weight correction part (0/1) */
int pygim_group_plan(int64_t handle, int64_t out[8]);
/* schedule of the LDS-staged product (k_lds_spmm; the reference's scratchpad loop spmm_default/dpu_kernels/
 * spmm_mul_csr_dpu.c:108-126 with X chunks in LDS and the running sums of a tile of rows in registers) of the same matrix:
 * row tiles (0 = no such plan), 80 KiB (320-column) chunk fills per 64-feature slice and product, tokens incl. padding, stored entries */
int pygim_group_lds_plan(int64_t handle, int64_t out[4]);
/* the same schedule compiled into gfx950 machine code (the code-stream form of the product, k_lds_code_*: one straight-line
 * instruction stream per (row tile, wave), 1.5 instructions per stored entry): bytes of code (0 = none), stored entries that share
 * an LDS instruction with a neighbour, 1 when products take this form (tunable "lds_code"), 1 when the stream was generated on the
 * device from the resident CSR (round 5, tunable "lds_codegen") rather than by the host encoder */
int pygim_group_lds_code(int64_t handle, int64_t out[4]);
/* which rows share a tile of that schedule (round 5; the reference hands a DPU consecutive rows, support/partition.c:51-99): 1 when the
 * rows were ordered by similarity (label propagation over the graph, tunable "lds_tile_order") instead of by index, the number of labels
 * the propagation ended with, the rows of the largest one; and whether the L2 sweep's work items are in locality order (0 = by length alone,
 * 1 = blocks of consecutive rows: the stored ids are local, 2 = blocks of the propagated order; tunable "panel_locality") */
int pygim_group_lds_tiles(int64_t handle, int64_t out[4]);
/* how many products (or blocks of pygim_block_run) the LDS-staged kernels have served for this group since it was created: a plan existing
 * (pygim_group_lds_plan) does not say that a given call took it -- widths, alignment and accumulation decide per call (the reference has one
 * kernel per build, spmm_default/dpu_kernels/spmm_mul_csr_dpu.c, so nothing to ask there) */
int pygim_group_lds_runs(int64_t handle, int64_t *out);
/* host operands (the reference driver's default call, spmm_test.py:29-35 with CPU tensors; spmm_default/pytorch_api.cpp:269-271 returns one): how many
 * feature windows the LAST run of this group moved as a pipeline -- window k + 1 uploaded while window k is multiplied and its result travels back
 * (tunable "host_windows"); 1 = upload, product and download one after the other; 0 = no run with host operands yet.  A window is a column range of X
 * and of C: every row's sum keeps its stored order.  *direct (may be NULL): 1 when the windows' products stored their rows straight into the caller's
 * page-locked result (tunable "host_direct"), 0 when the result was staged in device memory and copied down */
int pygim_group_host_windows(int64_t handle, int64_t *windows, int64_t *direct);
/* geometry of that schedule, as the library planned it (callers price staged bytes from THIS, not from assumed constants):
 * waves per workgroup, accumulators (rows) per wave, columns per chunk (chunk bytes = 256 x this), chunk buffers of the LDS ring,
 * staged columns per group of reads and x-register sets of a code stream (0 0 for a token plan), stored entries served by another
 * entry's LDS read (code streams: entries of different rows of one wave that share a column of a chunk), column ranges per row tile */
int pygim_group_lds_geometry(int64_t handle, int64_t out[8]);
/* which form of the product the group got at creation and, when it is not the fastest one, why (text, NUL-terminated, at most cap - 1
 * characters): "code-stream form" | "code-stream form not available: <reason>; products take the token form ..." | "... the L2 sweep
 * serves this group".  The ladder is code stream -> token kernels -> sweep; nothing falls back silently. */
int pygim_group_lds_note(int64_t handle, char *out, int64_t cap);
/* Kernel tunables (for A/B runs): name in {"long_row_threshold", "long_segment", "force_vec_bytes",
 * "csr_kernel", "coo_chunk", "coo_via_rowptr", "panel_mode", "panel_bytes", "panel_min_seg",
 * "panel_coop", "panel_block", "panel_lds_pad", "panel_pack", "panel_locality", "panel_col16", "slice_group_bytes", "fuse_windows",
 * "split_unit_pattern", "narrow_vals" (INT64 / DBL64 values that all fit int32 / float exactly are streamed in 4 bytes), "merge_parts", "vec_kernel", "vec_lds", "vec_lds_min_seg", "kernel_events",
 * "lds_mode" (LDS-staged product: 0 = by the reuse rule, 1 = whenever planned, 2 = never), "lds_min_reuse_x100",
 * "lds_min_width", "lds_threads", "lds_waves" (8 | 16 waves per workgroup of the kernel the plan is made for),
 * "lds_round_tiles", "lds_code" (1 = FLT32 (valued too) / INT32 / INT16 unit-weight plans are compiled into machine code at creation and run by k_lds_code_*,
 * 0 = the token kernels), "lds_col_split" (short row shares: tiles split into column ranges, partial sums reduced in range order; 0 = automatic,
 * 1 = never, S), "lds_col_split_f32" (FLT32 shares too -- their sums are then sums of per-range sums, inside the path's 1e-5 of |A|.|x| but not the CPU loop's bits:
 * 1 = parts of 2^20 entries and more (the default since round 6), 2 = any part, 0 = never), "lds_row_tail" (percent of a share's rows that may stay outside its LDS plan -- summed by the
 * tail kernels from the same staged copy -- when that leaves exactly one workgroup per compute unit; 0 = never), "lds_fill_tiles", "lds_split_order" (round 6: tile counts / tile order of
 * column-split plans), "lds_half_split" (round 6: 1 = products of at most 32 lanes of a 4-byte type fold two column ranges into the halves of a wave -- half the staged bytes; the plan is
 * written by the host encoder, so 0 is the default), "lds_code_nbuf" (its LDS ring: 0 = by width and geometry, 2 = two buffers of 320 columns with a barrier at every slot boundary,
 * 3 / 4 / 5 ... 10 = 192 / 160 / 128 ... 64 columns; the default is 5 x 128 for the 8-wave geometry), "lds_code_boundary" (rings of three or more
 * buffers: 0 / 1 = the workgroup meets at every slot boundary with nbuf - 1 chunks in flight, 2 = once in the middle of a slot with nbuf - 2),
 * "lds_code_waves" (waves per workgroup of a code-stream plan: 16 x 96 accumulators, 8 x 228 = taller tiles and fewer rounds of workgroups, 0 = automatic),
 * "lds_code_kc" (columns per chunk, 0 = by the ring), "lds_code_gsize" / "lds_code_nsets" (staged columns per group of LDS reads / x-register sets: the
 * reads run nsets - 1 groups ahead of the adds; 0 = default), "lds_codegen" (code streams: 1 = generated on the device from the resident CSR -- the default --, 0 = by
 * the host encoder, 2 = on the device and checked word for word against the host encoder), "lds_tile_order" (which rows share a tile: 0 = consecutive
 * rows, 1 = rows ordered by similarity, 2 = automatic: similarity for square parts of a million entries and more), "lds_lp_rounds", "lds_hybrid" (density split of a community-structured part whose ids carry no locality: the dense (tile, chunk) cells through the LDS-staged kernel, the rest through the sweep; 0 = off, 1 = integer types, 2 = floats too -- their sums are then reordered), "lds_hybrid_min" (entries a cell must hold), "lds_xcd_slices" (code-stream products: slices of X per XCD -- 0 = automatic (the default), 1 = an XCD streams one slice through
 * its L2; 2 / 4 = the workgroups an XCD runs side by side are slices of the same tile and share its code stream in L2), "lds_long_slots", "lds_ablate" (timing experiments, wrong results)};
 * returns the previous value, or -1 for an unknown name (pygim_last_error() says which).
 * READ AT GROUP CREATION (they shape the plan; changing them afterwards does not touch existing groups, and switching "lds_code" off
 * after a code-stream group was created sends that group's products to the sweep): panel_*, long_*, split_unit_pattern, narrow_vals,
 * merge_parts, lds_code, lds_codegen, lds_tile_order, lds_lp_rounds, lds_hybrid, lds_hybrid_min, lds_code_waves, lds_code_nbuf, lds_code_kc, lds_code_gsize, lds_code_nsets, lds_code_boundary, lds_waves, lds_col_split,
 * lds_col_split_f32, lds_row_tail, lds_fill_tiles, lds_split_order, lds_half_split, lds_min_width, lds_round_tiles, lds_long_slots, lds_min_reuse_x100 and lds_mode (whether a plan is made at all), lds_threads.
 * The others are read per product.  */
int64_t pygim_set_tunable(const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* PYGIM_HIP_H */
