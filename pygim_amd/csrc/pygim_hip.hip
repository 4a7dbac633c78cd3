// pygim_hip.hip -- host runtime + C ABI (include/pygim_hip.h) of the MI355X
// aggregation backend.  Device code lives in kernels.hpp.
//
// Reference boundary this file replaces, function by function:
//   dpu_init_ranks / dpu_init_dpus / dpu_release   spmm_default/pytorch_api.cpp:154-164
//   spmm_{csr,coo}_to_device_group                 spmm_default/pytorch_api.cpp:204-243, 286-329
//     prepare_pim_{csr,coo} + copy_sparse_{csr,coo}  spmm_mul_csr.c:118-330, spmm_mul_coo.c:83-318
//   spmm_{csr,coo}_run_group                       spmm_default/pytorch_api.cpp:248-280, 332-367
//     spmm_pim_{csr,coo}                             spmm_mul_csr.c:335-561, spmm_mul_coo.c:323-592
//   grande / spmv twins                            spmm_grande/pytorch_api.cpp:221-321,
//                                                  spmv_sparseP/pytorch_api.cpp:184-266
// No CPU compute fallback exists in this file: without a HIP device every entry
// point fails with PYGIM_ERR_NO_DEVICE.
#include "../../include/pygim_hip.h"
#include "kernels.hpp"
#ifdef PYGIM_LDS_ABLATE
#include "lds_kernel_gen_ablate.hpp"   // (make ablate: the same kernels + round 3's timing-experiment variants; not committed)
#else
#include "lds_kernel_gen.hpp"
#include "lds_codegen_dev.hpp"
#include "lds_reorder_dev.hpp"
#include "lds_hybrid_dev.hpp"
#endif
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

// rows of a slice of the slice-major copy: whole chunks of either kernel family (320-column chunks of the token kernels, 192 of the
// code-stream kernels), so that the DMA of a tile's last chunk stays inside the copy
// rows of a slice of the staged copy: whole chunks of the part's ring (the DMA of a tile's last chunk reads KC rows whatever the
// matrix's width), 64-row aligned
static inline uint64_t lds_rows_pad(uint64_t ncols, uint32_t kc) {
    const uint64_t k = kc ? kc : 320;
    return ((ncols + k - 1) / k * k + 63) / 64 * 64;
}

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <map>
#include <memory>
#include <vector>

using namespace pygim;

namespace {

#include "rt_state.inc"
#include "rt_launch.inc"
#include "rt_plans.inc"

}  // namespace

#include "rt_run.inc"

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

const char *pygim_last_error(void) { return g_err.c_str(); }

int pygim_is_initialized(void) { return g_ctx.inited ? 1 : 0; }

int pygim_init_ranks(int64_t nr_ranks, int64_t *units_per_rank) {
    if (nr_ranks <= 0) return fail(PYGIM_ERR_INVALID, "nr_ranks must be positive");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(PYGIM_ERR_NO_DEVICE, "no HIP device visible (this backend has no CPU fallback)");
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    g_ctx.device = dev;
    g_ctx.cu_count = prop.multiProcessorCount;
    g_ctx.nr_ranks = nr_ranks;
    g_ctx.inited = true;
    if (units_per_rank)
        for (int64_t i = 0; i < nr_ranks; i++) units_per_rank[i] = 8;  // one feature window per XCD
    return 0;
}

int pygim_init_units(int64_t nr_units, int64_t *units_per_rank, int64_t *nr_ranks_out) {
    // the reference rounds a DPU count up to whole ranks of 64 (dpu_alloc); here a
    // "unit" is a feature window and a rank holds 8 of them (one per XCD)
    if (nr_units <= 0) return fail(PYGIM_ERR_INVALID, "nr_units must be positive");
    const int64_t ranks = (nr_units + 7) / 8;
    if (nr_ranks_out) *nr_ranks_out = ranks;
    return pygim_init_ranks(ranks, units_per_rank);
}

int pygim_release(void) {
    std::vector<Group *> gs;
    {
        std::lock_guard<std::mutex> lk(g_ctx.mu);
        gs.assign(g_ctx.groups.begin(), g_ctx.groups.end());
        g_ctx.groups.clear();
    }
    if (g_ctx.inited) (void)hipDeviceSynchronize();
    for (Group *g : gs) free_group(g);
    {
        std::lock_guard<std::mutex> lk(g_ctx.mu);
        free_xs_buffers_locked();
    }
    g_ctx.inited = false;
    g_ctx.generation++;
    return 0;
}

int64_t pygim_generation(void) { return g_ctx.generation; }

int pygim_device_info(char *name, int name_len, int *cu_count, int64_t *hbm_bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(PYGIM_ERR_NO_DEVICE, "no HIP device visible");
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (name && name_len > 0) {
        std::snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return 0;
}

int64_t pygim_set_tunable(const char *name, int64_t value) {
    int64_t *slot = nullptr;
    std::string n = name ? name : "";
    if (n == "long_row_threshold") slot = &g_tune.long_row_threshold;
    else if (n == "long_segment") slot = &g_tune.long_segment;
    else if (n == "force_vec_bytes") slot = &g_tune.force_vec_bytes;
    else if (n == "csr_kernel") slot = &g_tune.csr_kernel;
    else if (n == "coo_chunk") slot = &g_tune.coo_chunk;
    else if (n == "coo_via_rowptr") slot = &g_tune.coo_via_rowptr;
    else if (n == "kernel_events") slot = &g_tune.kernel_events;
    else if (n == "host_windows") slot = &g_tune.host_windows;
    else if (n == "host_direct") slot = &g_tune.host_direct;
    else if (n == "panel_mode") slot = &g_tune.panel_mode;
    else if (n == "panel_bytes") slot = &g_tune.panel_bytes;
    else if (n == "panel_min_seg") slot = &g_tune.panel_min_seg;
    else if (n == "panel_pack") slot = &g_tune.panel_pack;
    else if (n == "panel_coop") slot = &g_tune.panel_coop;
    else if (n == "panel_block") slot = &g_tune.panel_block;
    else if (n == "slice_group_bytes") slot = &g_tune.slice_group_bytes;
    else if (n == "fuse_windows") slot = &g_tune.fuse_windows;
    else if (n == "panel_col16") slot = &g_tune.panel_col16;
    else if (n == "panel_locality") slot = &g_tune.panel_locality;
    else if (n == "split_unit_pattern") slot = &g_tune.split_unit_pattern;
    else if (n == "narrow_vals") slot = &g_tune.narrow_vals;
    else if (n == "merge_parts") slot = &g_tune.merge_parts;
    else if (n == "vec_kernel") slot = &g_tune.vec_kernel;
    else if (n == "vec_lds") slot = &g_tune.vec_lds;
    else if (n == "panel_lds_pad") slot = &g_tune.panel_lds_pad;
    else if (n == "vec_lds_min_seg") slot = &g_tune.vec_lds_min_seg;
    else if (n == "lds_mode") slot = &g_tune.lds_mode;
    else if (n == "lds_min_reuse_x100") slot = &g_tune.lds_min_reuse_x100;
    else if (n == "lds_min_width") slot = &g_tune.lds_min_width;
    else if (n == "lds_min_width8") slot = &g_tune.lds_min_width8;
    else if (n == "lds_min_reuse_narrow_x100") slot = &g_tune.lds_min_reuse_narrow_x100;
    else if (n == "lds_threads") slot = &g_tune.lds_threads;
    else if (n == "lds_waves") slot = &g_tune.lds_waves;
    else if (n == "lds_ablate") slot = &g_tune.lds_ablate;
    else if (n == "lds_round_tiles") slot = &g_tune.lds_round_tiles;
    else if (n == "lds_code") slot = &g_tune.lds_code;
    else if (n == "lds_code_nbuf") slot = &g_tune.lds_code_nbuf;
    else if (n == "lds_code_waves") slot = &g_tune.lds_code_waves;
    else if (n == "lds_fail") slot = &g_tune.lds_fail;
    else if (n == "lds_stamp") slot = &g_tune.lds_stamp;
    else if (n == "lds_code_exp") slot = &g_tune.lds_code_exp;
    else if (n == "lds_hybrid") slot = &g_tune.lds_hybrid;
    else if (n == "lds_hybrid_min") slot = &g_tune.lds_hybrid_min;
    else if (n == "lds_hybrid_contig") slot = &g_tune.lds_hybrid_contig;
    else if (n == "lds_code_boundary") slot = &g_tune.lds_code_boundary;
    else if (n == "lds_xcd_slices") slot = &g_tune.lds_xcd_slices;
    else if (n == "lds_codegen") slot = &g_tune.lds_codegen;
    else if (n == "lds_tile_order") slot = &g_tune.lds_tile_order;
    else if (n == "lds_lp_rounds") slot = &g_tune.lds_lp_rounds;
    else if (n == "lds_code_kc") slot = &g_tune.lds_code_kc;
    else if (n == "lds_code_gsize") slot = &g_tune.lds_code_gsize;
    else if (n == "lds_code_nsets") slot = &g_tune.lds_code_nsets;
    else if (n == "lds_col_split") slot = &g_tune.lds_col_split;
    else if (n == "lds_col_split_f32") slot = &g_tune.lds_col_split_f32;
    else if (n == "lds_fill_tiles") slot = &g_tune.lds_fill_tiles;
    else if (n == "lds_row_tail") slot = &g_tune.lds_row_tail;
    else if (n == "lds_split_order") slot = &g_tune.lds_split_order;
    else if (n == "lds_half_split") slot = &g_tune.lds_half_split;
    else if (n == "lds_direct_x") slot = &g_tune.lds_direct_x;
    else if (n == "lds_long_slots") slot = &g_tune.lds_long_slots;
    if (!slot) {
        fail(PYGIM_ERR_INVALID, "unknown tunable: " + n);
        return -1;
    }
    const int64_t old = *slot;
    if (slot == &g_tune.panel_lds_pad) value = std::min<int64_t>(std::max<int64_t>(value, 0), 64 << 10);  // goes into a launch's dynamic LDS size
    *slot = value;
    return old;
}

int pygim_group_create(int format, int dtype, int n_parts, const int32_t *const *idx0,
                       const int32_t *const *colind, const void *const *values, const int64_t *nrows,
                       const int64_t *ncols, const int64_t *nnz, const int64_t *n_dense,
                       const int64_t *dense_cols, int64_t h_size, int64_t *out_handle) {
    if (int rc = need_init()) return rc;
    if (format != PYGIM_CSR && format != PYGIM_COO) return fail(PYGIM_ERR_INVALID, "format must be CSR(0) or COO(1)");
    const size_t es = dtype_size(dtype);
    if (es == 0) return fail(PYGIM_ERR_INVALID, "unknown dtype");
    if (n_parts <= 0 || !idx0 || !colind || !nrows || !ncols || !nnz || !n_dense || !dense_cols || !out_handle)
        return fail(PYGIM_ERR_INVALID, "null argument or n_parts <= 0");
    if (h_size <= 0) return fail(PYGIM_ERR_INVALID, "h_size must be positive");
    const double t0 = now_ms();
    Group *g = new Group;
    g->format = format;
    g->dtype = dtype;
    t_plan_dtype = dtype;
    g->h = h_size;
    g->total_rows = nrows[0];
    g->parts.resize(n_parts);
    hipStream_t st = nullptr;
    int rc = 0;
    auto bail = [&](int code) {
        (void)hipDeviceSynchronize();
        free_group(g);
        return code;
    };
    if (hipMalloc((void **)&g->d_flags, 8 * sizeof(int)) != hipSuccess) return bail(fail(PYGIM_ERR_HIP, "hipMalloc flags"));
    if (hipMemsetAsync(g->d_flags, 0, 8 * sizeof(int), st) != hipSuccess) return bail(fail(PYGIM_ERR_HIP, "memset flags"));
    if (hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming) != hipSuccess)
        return bail(fail(PYGIM_ERR_HIP, "side stream / events"));
    size_t dpos = 0;
    for (int i = 0; i < n_parts; i++) {
        Part &p = g->parts[i];
        p.nrows = nrows[i];
        p.ncols = ncols[i];
        p.nnz = nnz[i];
        if (p.nrows != g->total_rows) return bail(fail(PYGIM_ERR_INVALID, "all sparse parts must have the same number of rows"));
        if (p.nrows < 0 || p.ncols < 0 || p.nnz < 0 || p.nnz > 0xFFFFFFFFll || p.nrows >= 0xFFFFFFFFll ||
            p.ncols > 0xFFFFFFFFll)
            return bail(fail(PYGIM_ERR_INVALID, "part sizes must fit 32-bit indices (the reference's uint32 matrices)"));
        if (n_dense[i] <= 0) return bail(fail(PYGIM_ERR_INVALID, "n_dense must be positive"));
        int64_t sum = 0;
        for (int64_t j = 0; j < n_dense[i]; j++) {
            if (dense_cols[dpos + j] < 0) return bail(fail(PYGIM_ERR_INVALID, "negative dense width"));
            p.dense_cols.push_back(dense_cols[dpos + j]);
            sum += dense_cols[dpos + j];
        }
        dpos += (size_t)n_dense[i];
        if (sum != h_size) return bail(fail(PYGIM_ERR_INVALID, "dense widths of a part must add up to h_size"));
        g->total_cols += p.ncols;
        if (!idx0[i] && (format == PYGIM_CSR || p.nnz > 0)) return bail(fail(PYGIM_ERR_INVALID, "null index array"));
        if (!colind[i] && p.nnz > 0) return bail(fail(PYGIM_ERR_INVALID, "null colind"));

        if ((rc = to_device<uint32_t>(colind[i], (size_t)p.nnz * 4, &p.colind, &p.own_colind, st))) return bail(rc);
        const void *v = values ? values[i] : nullptr;
        if (v) {
            if ((rc = to_device<void>(v, (size_t)p.nnz * es, &p.vals, &p.own_vals, st))) return bail(rc);
        }
        if (format == PYGIM_CSR) {
            if ((rc = to_device<uint32_t>(idx0[i], (size_t)(p.nrows + 1) * 4, &p.rowptr, &p.own_rowptr, st))) return bail(rc);
            hipLaunchKernelGGL(k_check_csr, dim3((unsigned)((std::max(p.nrows + 1, p.nnz) + 255) / 256)), dim3(256), 0,
                               st, p.rowptr, p.colind, (uint32_t)p.nrows, (uint32_t)p.nnz, (uint32_t)p.ncols,
                               g->d_flags);
        } else {
            if ((rc = to_device<uint32_t>(idx0[i], (size_t)p.nnz * 4, &p.rowind, &p.own_rowind, st))) return bail(rc);
            if (p.nnz > 0)
                hipLaunchKernelGGL(k_check_coo, dim3((unsigned)((p.nnz + 255) / 256)), dim3(256), 0, st, p.rowind,
                                   p.colind, (uint32_t)p.nnz, (uint32_t)p.nrows, (uint32_t)p.ncols, g->d_flags);
            // derived rowptr: the reference also builds a per-row histogram of a COO part at
            // to_device time (spmm_default/pytorch_api.cpp:315-318)
            if (hipMalloc((void **)&p.rowptr, (size_t)(p.nrows + 1) * 4) != hipSuccess)
                return bail(fail(PYGIM_ERR_HIP, "hipMalloc rowptr"));
            p.own_rowptr = true;
            hipLaunchKernelGGL(k_coo_rowptr, dim3((unsigned)((p.nnz + 1 + 255) / 256)), dim3(256), 0, st, p.rowind,
                               (uint32_t)p.nnz, (uint32_t)p.nrows, p.rowptr);
        }
        if (p.vals && p.nnz > 0) {
            int *f = g->d_flags + 2;
            switch (dtype) {
                case PYGIM_INT8: launch_check_ones<int8_t>(p.vals, (uint32_t)p.nnz, f, st); break;
                case PYGIM_INT16: launch_check_ones<int16_t>(p.vals, (uint32_t)p.nnz, f, st); break;
                case PYGIM_INT32: launch_check_ones<int32_t>(p.vals, (uint32_t)p.nnz, f, st); break;
                case PYGIM_INT64: launch_check_ones<int64_t>(p.vals, (uint32_t)p.nnz, f, st); break;
                case PYGIM_FLT32: launch_check_ones<float>(p.vals, (uint32_t)p.nnz, f, st); break;
                case PYGIM_DBL64: launch_check_ones<double>(p.vals, (uint32_t)p.nnz, f, st); break;
            }
        }
    }
    int flags[4] = {0, 0, 0, 0};
    if (hipMemcpy(flags, g->d_flags, sizeof(flags), hipMemcpyDeviceToHost) != hipSuccess)
        return bail(fail(PYGIM_ERR_HIP, std::string("group validation: ") + hipGetErrorString(hipGetLastError())));
    if (flags[1]) return bail(fail(PYGIM_ERR_INVALID, "index out of range in a sparse part"));
    if (flags[0]) {
        if (format == PYGIM_COO) return bail(fail(PYGIM_ERR_UNSORTED, "COO row indices are not sorted (pass a coalesced tensor)"));
        return bail(fail(PYGIM_ERR_INVALID, "rowptr is not a non-decreasing prefix array ending at nnz"));
    }
    g->all_ones = (flags[2] == 0);
    // several sparse parts: also one merged matrix (built from the parts as given, before values are dropped or split)
    if (n_parts > 1 && g_tune.merge_parts && (rc = build_merged(g, es, st))) return bail(rc);
    // long-row plan needs rowptr on the host
    for (int i = 0; i < n_parts; i++) {
        Part &p = g->parts[i];
        if (g->all_ones && p.vals) {
            // unit weights: drop the value array from the hot loop (legitimate: 1*x == x exactly)
            if (p.own_vals) (void)hipFree(p.vals);
            p.vals = nullptr;
            p.own_vals = false;
        }
        if (!g->all_ones && (rc = split_unit_pattern(p, dtype, es, g->d_flags + 4, (uint32_t *)(g->d_flags + 5), st))) return bail(rc);
        if ((rc = build_plans(p, es, g->d_flags + 4, st, g->h, /*allow_lds=*/!(g->merged && g_tune.merge_parts)))) return bail(rc);
        if (!(g->merged && g_tune.merge_parts) && (rc = narrow_values(p, dtype, g->d_flags + 6, st))) return bail(rc);
        if (!(g->merged && g_tune.merge_parts) && (rc = build_hybrid(p, dtype, es, g->d_flags + 4, st, g->h))) return bail(rc);
    }
    if (hipDeviceSynchronize() != hipSuccess) return bail(fail(PYGIM_ERR_HIP, "sync after create"));
    g->timers[4] = now_ms() - t0;
    {
        std::lock_guard<std::mutex> lk(g_ctx.mu);
        g->serial = ++g_ctx.group_serial;
        g_ctx.groups.insert(g);
    }
    *out_handle = (int64_t) reinterpret_cast<uintptr_t>(g);
    return 0;
}

int pygim_group_free(int64_t handle) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(g_ctx.mu);
        g_ctx.groups.erase(g);
        last = g_ctx.groups.empty();
    }
    (void)hipDeviceSynchronize();
    free_group(g);
    if (last) {  // the slice-major copies belong to the products of live groups: none left, none kept
        std::lock_guard<std::mutex> lk(g_ctx.mu);
        free_xs_buffers_locked();
    }
    return 0;
}

int pygim_quant_spmm_run(int64_t handle, const float *X, int64_t ldx, float *out, float *scale_out, void *stream) {
    return pygim_quant_spmm_run_post(handle, X, ldx, out, scale_out, nullptr, nullptr, 0, stream);
}

int pygim_quant_spmm_run_post(int64_t handle, const float *X, int64_t ldx, float *out, float *scale_out, const float *col_mul,
                              const float *col_add, int relu, void *stream) {
    if (int rc = need_init()) return rc;
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    if ((col_mul == nullptr) != (col_add == nullptr)) return fail(PYGIM_ERR_INVALID, "col_mul and col_add come together");
    if (col_mul && (!is_device_ptr(col_mul) || !is_device_ptr(col_add))) return fail(PYGIM_ERR_INVALID, "epilogue vectors must be device memory");
    struct PostGuard {
        Group *g;
        ~PostGuard() { g->post_mul = g->post_add = nullptr; g->post_relu = 0; }
    } guard{g};
    g->post_mul = col_mul;
    g->post_add = col_add;
    g->post_relu = col_mul ? relu : 0;
    if (!X || !out || ldx < g->h) return fail(PYGIM_ERR_INVALID, "bad X / out / ldx");
    if (!is_device_ptr(X) || !is_device_ptr(out) || (scale_out && !is_device_ptr(scale_out)))
        return fail(PYGIM_ERR_INVALID, "pygim_quant_spmm_run needs device pointers");
    for (auto &p : g->parts)
        if (p.dense_cols != g->parts[0].dense_cols) return fail(PYGIM_ERR_INVALID, "needs one dense split for all parts");
    hipStream_t st = (hipStream_t)stream;
    switch (g->dtype) {  // ranges of models/quantize.py:22-30
        case PYGIM_INT8: return quant_run_t<int8_t>(g, X, ldx, out, scale_out, 5, st);
        case PYGIM_INT16: return quant_run_t<int16_t>(g, X, ldx, out, scale_out, 10, st);
        case PYGIM_INT32: return quant_run_t<int32_t>(g, X, ldx, out, scale_out, 20, st);
        case PYGIM_FLT32: return quant_run_t<float>(g, X, ldx, out, scale_out, 20, st);
        default: return fail(PYGIM_ERR_INVALID, "quantised run: group type must be INT8/INT16/INT32/FLT32");
    }
}

int pygim_spmm_run_dequant(int64_t handle, const void *Xq, int64_t ldx, float *out, const uint32_t *absmax_bits, void *stream) {
    if (int rc = need_init()) return rc;
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    if (!Xq || !out || !absmax_bits || ldx < g->h) return fail(PYGIM_ERR_INVALID, "bad Xq / out / absmax / ldx");
    if (!is_device_ptr(Xq) || !is_device_ptr(out) || !is_device_ptr(absmax_bits))
        return fail(PYGIM_ERR_INVALID, "pygim_spmm_run_dequant needs device pointers");
    for (auto &p : g->parts)
        if (p.dense_cols != g->parts[0].dense_cols) return fail(PYGIM_ERR_INVALID, "needs one dense split for all parts");
    const int k = quant_log2_range(g->dtype);
    hipStream_t st = (hipStream_t)stream;
    switch (g->dtype) {
        case PYGIM_INT8: return dequant_run_t<int8_t>(g, Xq, ldx, out, absmax_bits, k, st);
        case PYGIM_INT16: return dequant_run_t<int16_t>(g, Xq, ldx, out, absmax_bits, k, st);
        case PYGIM_INT32: return dequant_run_t<int32_t>(g, Xq, ldx, out, absmax_bits, k, st);
        case PYGIM_FLT32: return dequant_run_t<float>(g, Xq, ldx, out, absmax_bits, k, st);
        default: return fail(PYGIM_ERR_INVALID, "dequantising run: group type must be INT8/INT16/INT32/FLT32");
    }
}

int pygim_quant_absmax(const float *X, int64_t ldx, int64_t rows, int64_t width, uint32_t *absmax_bits, void *stream) {
    if (int rc = need_init()) return rc;
    if (rows < 0 || width < 0 || ldx < width || !absmax_bits) return fail(PYGIM_ERR_INVALID, "bad absmax arguments");
    if ((rows * width > 0 && (!X || !is_device_ptr(X))) || !is_device_ptr(absmax_bits))
        return fail(PYGIM_ERR_INVALID, "pygim_quant_absmax needs device pointers");
    return launch_absmax(X, ldx, (uint64_t)rows, (uint32_t)width, absmax_bits, (hipStream_t)stream);
}

int pygim_quantize(int dtype, const float *X, int64_t ldx, int64_t rows, int64_t width, const uint32_t *absmax_bits,
                   void *Xq, float *scale_out, void *stream) {
    if (int rc = need_init()) return rc;
    const int k = quant_log2_range(dtype);
    if (k < 0) return fail(PYGIM_ERR_INVALID, "quantise: type must be INT8/INT16/INT32/FLT32");
    if (rows < 0 || width < 0 || ldx < width || !absmax_bits) return fail(PYGIM_ERR_INVALID, "bad quantise arguments");
    if (rows * width > 0 && (!X || !Xq || !is_device_ptr(X) || !is_device_ptr(Xq)))
        return fail(PYGIM_ERR_INVALID, "pygim_quantize needs device pointers");
    hipStream_t st = (hipStream_t)stream;
    const uint64_t r = (uint64_t)rows;
    const uint32_t w = (uint32_t)width;
    switch (dtype) {
        case PYGIM_INT8: return launch_quantize<int8_t>(X, ldx, r, w, absmax_bits, k, (int8_t *)Xq, scale_out, st);
        case PYGIM_INT16: return launch_quantize<int16_t>(X, ldx, r, w, absmax_bits, k, (int16_t *)Xq, scale_out, st);
        case PYGIM_INT32: return launch_quantize<int32_t>(X, ldx, r, w, absmax_bits, k, (int32_t *)Xq, scale_out, st);
        default: return launch_quantize<float>(X, ldx, r, w, absmax_bits, k, (float *)Xq, scale_out, st);
    }
}

int pygim_dequantize(int dtype, const void *Q, int64_t n, const uint32_t *absmax_bits, float *out, void *stream) {
    if (int rc = need_init()) return rc;
    const int k = quant_log2_range(dtype);
    if (k < 0) return fail(PYGIM_ERR_INVALID, "dequantise: type must be INT8/INT16/INT32/FLT32");
    if (n < 0 || !absmax_bits) return fail(PYGIM_ERR_INVALID, "bad dequantise arguments");
    if (n > 0 && (!Q || !out || !is_device_ptr(Q) || !is_device_ptr(out)))
        return fail(PYGIM_ERR_INVALID, "pygim_dequantize needs device pointers");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case PYGIM_INT8: return launch_dequantize<int8_t>((const int8_t *)Q, (uint64_t)n, absmax_bits, k, out, st);
        case PYGIM_INT16: return launch_dequantize<int16_t>((const int16_t *)Q, (uint64_t)n, absmax_bits, k, out, st);
        case PYGIM_INT32: return launch_dequantize<int32_t>((const int32_t *)Q, (uint64_t)n, absmax_bits, k, out, st);
        default: return launch_dequantize<float>((const float *)Q, (uint64_t)n, absmax_bits, k, out, st);
    }
}

int pygim_group_timers(int64_t handle, double out_ms[5]) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    for (int i = 0; i < 5; i++) out_ms[i] = g->timers[i];
    return 0;
}

int pygim_group_kernel_ms(int64_t handle, double *sum_ms, int64_t *count, int reset) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    for (auto &e : g->ev_pending) {
        float ms = 0;
        if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) {
            g->ev_ms += ms;
            g->ev_count++;
        }
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    g->ev_pending.clear();
    if (sum_ms) *sum_ms = g->ev_ms;
    if (count) *count = g->ev_count;
    if (reset) {
        g->ev_ms = 0;
        g->ev_count = 0;
    }
    return 0;
}

int pygim_group_kernel_events(int64_t handle, int on) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    g->kernel_events = on != 0;
    return 0;
}

int pygim_group_plan(int64_t handle, int64_t out[8]) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    const Part &p = (g->merged && g_tune.merge_parts && g->parts.size() > 1) ? *g->merged : g->parts[0];
    int64_t ncoop = 0;
    for (auto c : p.panel_coop) ncoop += c;
    out[0] = p.d_items ? (int64_t)p.npanels : 0;
    out[1] = p.panel_cols;
    out[2] = (int64_t)p.n_items;
    out[3] = p.col16 ? 1 : 0;
    out[4] = ncoop;
    out[5] = p.lp_panel.n_tasks;
    out[6] = (g->merged && g->parts.size() > 1) ? 1 : 0;
    out[7] = p.extra ? 1 : 0;
    return 0;
}

int pygim_group_lds_plan(int64_t handle, int64_t out[4]) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    const Part &p = (g->merged && g_tune.merge_parts && g->parts.size() > 1) ? *g->merged : g->parts[0];
    out[0] = p.lds_tiles ? (int64_t)p.lds_ntiles : 0;
    out[1] = (int64_t)p.lds_slots;
    out[2] = (int64_t)p.lds_tokens;
    out[3] = p.nnz;
    return 0;
}

int pygim_group_lds_code(int64_t handle, int64_t out[4]) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    const Part &p = (g->merged && g_tune.merge_parts && g->parts.size() > 1) ? *g->merged : g->parts[0];
    out[0] = p.lds_code ? (int64_t)p.lds_code_bytes : 0;
    out[1] = p.lds_code ? (int64_t)p.lds_code_pairs : 0;
    bool all_code = p.lds_is_code && g_tune.lds_code;
    if (!(g->merged && g_tune.merge_parts && g->parts.size() > 1))
        for (const Part &q : g->parts) all_code = all_code && q.lds_is_code;   // (1 only when EVERY part that serves products is a code stream)
    out[2] = all_code ? 1 : 0;
    out[3] = (p.lds_code && p.lds_codegen_device) ? 1 : 0;   // generated on the device (round 5), not by the host encoder
    return 0;
}

int pygim_group_lds_tiles(int64_t handle, int64_t out[4]) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    const Part &p = (g->merged && g_tune.merge_parts && g->parts.size() > 1) ? *g->merged : g->parts[0];
    out[0] = p.lds_tile_labels ? 1 : 0;
    out[1] = p.lds_tile_labels;
    out[2] = p.lds_tile_largest;
    out[3] = p.panel_locality_used ? p.sim_kind : 0;   // the sweep's items in locality order: 1 = by id (local ids), 2 = by propagated label
    return 0;
}

int pygim_group_serial(int64_t handle, int64_t *out) {
    Group *g = lookup(handle);
    if (!g || !out) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    *out = g->serial;
    return 0;
}

int pygim_group_lds_runs(int64_t handle, int64_t *out) {
    Group *g = lookup(handle);
    if (!g || !out) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    *out = g->lds_runs;
    return 0;
}

int pygim_group_host_windows(int64_t handle, int64_t *windows, int64_t *direct) {
    Group *g = lookup(handle);
    if (!g || !windows) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    *windows = g->host_windows_used;
    if (direct) *direct = g->host_direct_used;
    return 0;
}

int pygim_group_lds_geometry(int64_t handle, int64_t out[8]) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    const Part &p = (g->merged && g_tune.merge_parts && g->parts.size() > 1) ? *g->merged : g->parts[0];
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (!p.lds_tiles) return 0;
    out[0] = p.lds_nw;
    out[1] = p.lds_ka;
    out[2] = p.lds_kc;
    out[3] = p.lds_nbuf;
    out[4] = p.lds_is_code ? p.lds_code_gsize : 0;
    out[5] = p.lds_is_code ? p.lds_code_nsets : 0;
    out[6] = p.lds_is_code ? (int64_t)p.lds_code_shared : 0;
    out[7] = p.lds_col_splits;
    return 0;
}

int pygim_group_lds_note(int64_t handle, char *out, int64_t cap) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    if (!out || cap <= 0) return fail(PYGIM_ERR_INVALID, "no buffer");
    // the parts that serve the group's products: the merged matrix, or -- several parts run one by one -- every part: a later part that
    // took a lower rung of the ladder than part 0 is named too (nothing falls back silently)
    std::string text;
    if ((g->merged && g_tune.merge_parts && g->parts.size() > 1) || g->parts.size() == 1) {
        text = ((g->merged && g_tune.merge_parts && g->parts.size() > 1) ? *g->merged : g->parts[0]).lds_note;
    } else {
        text = g->parts[0].lds_note;
        for (size_t i = 1; i < g->parts.size(); i++)
            if (g->parts[i].lds_note != g->parts[0].lds_note) text += "; part " + std::to_string(i) + ": " + g->parts[i].lds_note;
    }
    for (size_t i = 0; i < g->parts.size(); i++)
        if (!g->parts[i].hy_note.empty()) text += "; part " + std::to_string(i) + ": " + g->parts[i].hy_note;   // (a split that was tried and dropped says why)
    const size_t n = std::min<size_t>(text.size(), (size_t)cap - 1);
    std::memcpy(out, text.data(), n);
    out[n] = 0;
    return 0;
}

int pygim_group_info(int64_t handle, int64_t out[8]) {
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    out[0] = g->total_rows;
    out[1] = g->total_cols;
    out[2] = g->h;
    out[3] = (int64_t)g->parts.size();
    int64_t nl = 0;
    for (auto &p : g->parts) nl += p.lp_base.n_long;
    out[4] = nl;
    out[5] = g->all_ones ? 1 : 0;
    out[6] = g->parts[0].d_items ? (int64_t)g->parts[0].npanels : 0;
    out[7] = (int64_t)g->parts[0].n_items;
    return 0;
}

int pygim_block_run_x(int64_t handle, int part, const void *X, int64_t ldx, void *C, int64_t ldc, int64_t width,
                      int accumulate, int x_unchanged, void *stream) {
    if (int rc = need_init()) return rc;
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    if (part < 0 || part >= (int)g->parts.size()) return fail(PYGIM_ERR_INVALID, "part index out of range");
    if (!X || !C || width <= 0 || ldx < width || ldc < width) return fail(PYGIM_ERR_INVALID, "bad X/C/width/stride");
    if (!is_device_ptr(X) || !is_device_ptr(C)) return fail(PYGIM_ERR_INVALID, "pygim_block_run needs device pointers");
    g->x_unchanged = x_unchanged != 0;
    const int rc = launch_block_any(g, g->parts[part], X, ldx, C, ldc, width, accumulate != 0, (hipStream_t)stream);
    g->x_unchanged = false;
    return rc;
}

int pygim_block_run(int64_t handle, int part, const void *X, int64_t ldx, void *C, int64_t ldc, int64_t width,
                    int accumulate, void *stream) {
    return pygim_block_run_x(handle, part, X, ldx, C, ldc, width, accumulate, 0, stream);
}

int pygim_spmm_run_group_x(int64_t handle, const void *const *B_parts, void *out, int x_unchanged, void *stream) {
    if (int rc = need_init()) return rc;
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    for (auto &p : g->parts)
        if (p.dense_cols != g->parts[0].dense_cols)
            return fail(PYGIM_ERR_INVALID, "spmm_run_group needs the same dense split for every sparse part");
    g->x_unchanged = x_unchanged != 0;
    const int rc = run_group_common(g, B_parts, nullptr, false, out, (hipStream_t)stream);
    g->x_unchanged = false;
    return rc;
}

int pygim_spmm_run_group(int64_t handle, const void *const *B_parts, void *out, void *stream) {
    return pygim_spmm_run_group_x(handle, B_parts, out, 0, stream);
}

int pygim_grande_run_group(int64_t handle, const void *const *B_windows, const int64_t *window_ld, void *out,
                           void *stream) {
    if (int rc = need_init()) return rc;
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    if (!window_ld) return fail(PYGIM_ERR_INVALID, "window_ld is required");
    return run_group_common(g, B_windows, window_ld, true, out, (hipStream_t)stream);
}

int pygim_spmv_run_group(int64_t handle, const void *const *B_vectors, void *out, void *stream) {
    if (int rc = need_init()) return rc;
    Group *g = lookup(handle);
    if (!g) return fail(PYGIM_ERR_INVALID, "unknown group handle");
    hipStream_t st = (hipStream_t)stream;
    const size_t es = dtype_size(g->dtype);
    const size_t nvec = g->parts[0].dense_cols.size();
    for (auto &p : g->parts) {
        if (p.dense_cols.size() != nvec) return fail(PYGIM_ERR_INVALID, "spmv group: dense split differs between parts");
        for (auto wdt : p.dense_cols)
            if (wdt != 1) return fail(PYGIM_ERR_INVALID, "spmv group: every dense part must be one column wide");
    }
    if (!B_vectors || !out) return fail(PYGIM_ERR_INVALID, "null vectors or output");
    for (size_t j = 0; j < nvec; j++)
        if (!B_vectors[j]) return fail(PYGIM_ERR_INVALID, "null vector");
    const bool dev_in = is_device_ptr(B_vectors[0]);
    const bool dev_out = is_device_ptr(out);
    for (size_t j = 1; j < nvec; j++)
        if (is_device_ptr(B_vectors[j]) != dev_in) return fail(PYGIM_ERR_INVALID, "vectors mix host and device memory");
    if (dev_in != dev_out) return fail(PYGIM_ERR_INVALID, "vectors and output must both be host or both be device memory");
    const uint64_t n = (uint64_t)g->total_cols;
    // layout of stage_in: [packed panel n x nvec][raw vectors (host mode only)]
    const size_t panel_bytes = ((size_t)n * nvec * es + 255) & ~(size_t)255;
    const size_t vec_bytes = ((size_t)n * es + 255) & ~(size_t)255;
    const size_t need_in = panel_bytes + (dev_in ? 0 : vec_bytes * nvec);
    if (int rc = ensure(&g->stage_in, &g->stage_in_bytes, std::max<size_t>(need_in, 256))) return rc;
    if (g->d_ptrs_n < nvec) {
        if (g->d_ptrs) HIP_TRY(hipFree(g->d_ptrs));  // (waits for earlier packs)
        g->d_ptrs = nullptr;
        HIP_TRY(hipMalloc((void **)&g->d_ptrs, 2 * nvec * sizeof(void *)));
        g->d_ptrs_n = nvec;
    }
    // pointer table of the pack kernel: page-locked host memory, two slots used in turn, each released by an event
    // recorded behind the pack kernel that read it -- device-pointer calls only enqueue work (no host sync)
    if (g->h_ptrs_n < nvec) {
        if (g->h_ptrs) HIP_TRY(hipHostFree(g->h_ptrs));
        g->h_ptrs = nullptr;
        HIP_TRY(hipHostMalloc((void **)&g->h_ptrs, 2 * nvec * sizeof(void *), hipHostMallocDefault));
        g->h_ptrs_n = nvec;
        for (hipEvent_t &e : g->ev_ptrs)
            if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int slot = g->ptr_slot;
    g->ptr_slot ^= 1;
    HIP_TRY(hipEventSynchronize(g->ev_ptrs[slot]));  // (a never-recorded event is complete)
    const void **src = (const void **)(g->h_ptrs + (size_t)slot * g->h_ptrs_n);
    const double t0 = now_ms();
    for (size_t j = 0; j < nvec; j++) {
        if (dev_in) {
            src[j] = B_vectors[j];
        } else {
            void *d = (char *)g->stage_in + panel_bytes + j * vec_bytes;
            if (n) HIP_TRY(hipMemcpyAsync(d, B_vectors[j], (size_t)n * es, hipMemcpyHostToDevice, st));
            src[j] = d;
        }
    }
    if (!dev_in) HIP_TRY(hipStreamSynchronize(st));  // host mode: the upload time is reported
    HIP_TRY(hipMemcpyAsync(g->d_ptrs + (size_t)slot * nvec, src, nvec * sizeof(void *), hipMemcpyHostToDevice, st));
    const double t1 = now_ms();
    const void *const *tab = (const void *const *)(g->d_ptrs + (size_t)slot * nvec);
    switch (es) {
        case 1: launch_pack<int8_t>(tab, (uint32_t)nvec, n, g->stage_in, st); break;
        case 2: launch_pack<int16_t>(tab, (uint32_t)nvec, n, g->stage_in, st); break;
        case 4: launch_pack<int32_t>(tab, (uint32_t)nvec, n, g->stage_in, st); break;
        case 8: launch_pack<int64_t>(tab, (uint32_t)nvec, n, g->stage_in, st); break;
    }
    HIP_TRY(hipEventRecord(g->ev_ptrs[slot], st));
    void *dout = out;
    if (!dev_out) {
        if (int rc = ensure(&g->stage_out, &g->stage_out_bytes, std::max<size_t>((size_t)g->total_rows * nvec * es, 256))) return rc;
        dout = g->stage_out;
    }
    if (g->merged && g_tune.merge_parts && g->parts.size() > 1) {
        if (int rc = launch_block_any(g, *g->merged, g->stage_in, (int64_t)nvec, dout, (int64_t)nvec, (int64_t)nvec, false, st))
            return rc;
    } else {
        int64_t brow = 0;
        for (size_t i = 0; i < g->parts.size(); i++) {
            Part &p = g->parts[i];
            const char *x = (const char *)g->stage_in + (size_t)brow * nvec * es;
            if (int rc = launch_block_any(g, p, x, (int64_t)nvec, dout, (int64_t)nvec, (int64_t)nvec, i > 0, st)) return rc;
            brow += p.ncols;
        }
    }
    if (!dev_out) {
        HIP_TRY(hipStreamSynchronize(st));
        const double t2 = now_ms();
        const size_t bytes = (size_t)g->total_rows * nvec * es;
        if (bytes) HIP_TRY(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        g->timers[0] = t1 - t0;
        g->timers[1] = t2 - t1;
        g->timers[2] = now_ms() - t2;
        g->timers[3] = 0;
    }
    return 0;
}

}  // extern "C"
