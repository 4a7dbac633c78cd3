// pim_ops_torch.cpp -- TORCH_LIBRARY(pim_ops) shim over the C ABI (include/pygim_hip.h), so that the
// reference's `torch.ops.load_library(args.lib_path)` (spmm_test.py:111, inference.py:134) works
// verbatim.  One shared object per variant, like the reference (its schemas differ per variant):
//   -DPYGIM_VARIANT=0  spmm_default  (spmm_default/pytorch_api.cpp:372-389)
//   -DPYGIM_VARIANT=1  spmm_grande   (spmm_grande/pytorch_api.cpp:326-343)
//   -DPYGIM_VARIANT=2  spmv_sparseP  (spmv_sparseP/pytorch_api.cpp:271-287)
// No arithmetic here: tensors in, raw pointers to the HIP library, tensor out.
#include <torch/library.h>
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pygim_hip.h"

#ifndef PYGIM_VARIANT
#define PYGIM_VARIANT 0
#endif

namespace {

struct Meta {
    at::ScalarType dtype;
    int64_t rows, cols, h;
    std::vector<int64_t> n_dense, dense_cols, part_cols;
    std::vector<at::Tensor> keep;  // device arrays are used in place by the library
};
std::unordered_map<int64_t, Meta> g_meta;

void chk(int rc) { TORCH_CHECK(rc == 0, "pygim_hip error ", rc, ": ", pygim_last_error()); }

int dt_code(at::ScalarType t) {
    switch (t) {
        case at::kChar: return PYGIM_INT8;
        case at::kShort: return PYGIM_INT16;
        case at::kInt: return PYGIM_INT32;
        case at::kLong: return PYGIM_INT64;
        case at::kFloat: return PYGIM_FLT32;
        case at::kDouble: return PYGIM_DBL64;
        default: TORCH_CHECK(false, "unsupported value dtype ", t);
    }
    return -1;
}

void *stream_of(const at::Tensor &t) {
    return t.is_cuda() ? (void *)c10::hip::getCurrentHIPStream(t.get_device()).stream() : nullptr;
}

int64_t to_device_group(int format, std::vector<at::Tensor> idx0, std::vector<at::Tensor> colind,
                        std::vector<at::Tensor> values, std::vector<int64_t> nrows, std::vector<int64_t> ncols,
                        std::vector<int64_t> n_dense, std::vector<int64_t> dense_cols, int64_t h_size) {
    const size_t n = ncols.size();
    TORCH_CHECK(n > 0 && idx0.size() == n && colind.size() == n && values.size() == n && nrows.size() == n,
                "part lists differ in length");
    Meta m;
    m.dtype = values[0].scalar_type();
    std::vector<const int32_t *> a(n), b(n);
    std::vector<const void *> v(n);
    std::vector<int64_t> nnz(n);
    for (size_t i = 0; i < n; i++) {
        // the reference reinterprets int32 tensors (pytorch_api.cpp:230-231); data_ptr<int32_t>() throws otherwise
        idx0[i] = idx0[i].contiguous();
        colind[i] = colind[i].contiguous();
        values[i] = values[i].contiguous();
        TORCH_CHECK(values[i].scalar_type() == m.dtype, "expected scalar type ", m.dtype, " but found ",
                    values[i].scalar_type());
        a[i] = idx0[i].data_ptr<int32_t>();
        b[i] = colind[i].data_ptr<int32_t>();
        v[i] = values[i].data_ptr();
        nnz[i] = values[i].numel();
        if (values[i].is_cuda()) {
            m.keep.push_back(idx0[i]);
            m.keep.push_back(colind[i]);
            m.keep.push_back(values[i]);
        }
    }
    int64_t handle = 0;
    chk(pygim_group_create(format, dt_code(m.dtype), (int)n, a.data(), b.data(), v.data(), nrows.data(), ncols.data(),
                           nnz.data(), n_dense.data(), dense_cols.data(), h_size, &handle));
    m.rows = nrows[0];
    m.cols = 0;
    for (auto c : ncols) m.cols += c;
    m.h = h_size;
    m.n_dense = n_dense;
    m.dense_cols = dense_cols;
    m.part_cols = ncols;
    g_meta[handle] = std::move(m);
    return handle;
}

const Meta &meta(int64_t handle) {
    auto it = g_meta.find(handle);
    TORCH_CHECK(it != g_meta.end(), "unknown sparse group handle ", handle);
    return it->second;
}

std::vector<int64_t> repeat(const std::vector<int64_t> &v, size_t n) {
    std::vector<int64_t> out;
    for (size_t i = 0; i < n; i++) out.insert(out.end(), v.begin(), v.end());
    return out;
}

void dpu_release() {
    g_meta.clear();
    chk(pygim_release());
}
void spmm_free_group(int64_t handle) {
    g_meta.erase(handle);
    chk(pygim_group_free(handle));
}

// the reference prints its Timer buckets after every run (spmm_default/spmm_mul_csr.c:563-580; parsed by
// utils/experiment.py:466-491): same keys, milliseconds, for host-tensor calls when PYGIM_DATA_LOG=1
void log_timers(int64_t handle, bool on_host) {
    const char *e = std::getenv("PYGIM_DATA_LOG");
    if (!on_host || !e || e[0] != '1') return;
    double t[5] = {0, 0, 0, 0, 0};
    if (pygim_group_timers(handle, t) != 0) return;
    std::printf("[DATA]load_sparse_time: %.3f\n[DATA]load_dense_time: %.3f\n[DATA]kernel_time: %.3f\n"
                "[DATA]retrieve_result_time: %.3f\n[DATA]alignment_time: %.3f\n", t[4], t[0], t[1], t[2], t[3]);
    std::fflush(stdout);
}

at::Tensor run_common(int64_t handle, std::vector<at::Tensor> &parts, int kind) {
    const Meta &m = meta(handle);
    // same checks as the Python registration (pim_ops.py _check_dense and the per-op shape checks): the C side indexes
    // its window table by the group's own counts and gathers rows of B up to the group's column count
    int64_t expect = 0;
    if (kind == 1)
        for (auto v : m.n_dense) expect += v;
    else
        expect = m.n_dense.empty() ? 0 : m.n_dense[0];
    TORCH_CHECK((int64_t)parts.size() == expect, "expected ", expect, " dense parts, got ", parts.size());
    TORCH_CHECK(!parts.empty(), "no dense parts");
    std::vector<const void *> ptrs;
    std::vector<int64_t> lds;
    size_t k = 0, part = 0;
    int64_t in_part = 0;
    for (auto &p : parts) {
        TORCH_CHECK(p.scalar_type() == m.dtype, "expected scalar type ", m.dtype, " but found ", p.scalar_type());
        TORCH_CHECK(p.device() == parts[0].device(), "dense parts must live on one device");
        if (kind == 2) {
            TORCH_CHECK(p.numel() == m.cols, "vector ", k, " has ", p.numel(), " elements, expected ", m.cols);
        } else if (kind == 0) {
            TORCH_CHECK(p.dim() == 2 && p.size(0) == m.cols && p.size(1) == m.dense_cols[k], "dense part ", k, " has shape ",
                        p.sizes(), ", expected (", m.cols, ", ", m.dense_cols[k], ")");
        } else {
            // grande: window k belongs to sparse part `part` and holds that part's rows of B
            TORCH_CHECK(p.dim() == 2 && p.size(0) == m.part_cols[part] && p.size(1) >= m.dense_cols[k], "window ", k,
                        " has shape ", p.sizes(), ", expected (", m.part_cols[part], ", >= ", m.dense_cols[k], ")");
            if (++in_part == m.n_dense[part]) {
                part++;
                in_part = 0;
            }
        }
        p = p.contiguous();
        ptrs.push_back(p.data_ptr());
        lds.push_back(p.dim() == 2 ? p.size(1) : 1);
        k++;
    }
    const int64_t out_cols = kind == 2 ? m.n_dense[0] : m.h;
    // host operands: the result lives in page-locked host memory (torch's caching host allocator), so the device-to-host
    // copy runs at PCIe speed instead of faulting fresh pages in; PYGIM_PINNED_OUT=0 gives pageable memory
    const char *po = std::getenv("PYGIM_PINNED_OUT");
    const bool pinned = !parts[0].is_cuda() && !(po && po[0] == '0');
    at::Tensor out = at::empty({m.rows, out_cols}, parts[0].options().pinned_memory(pinned));
    void *st = stream_of(out);
    if (kind == 0) chk(pygim_spmm_run_group(handle, ptrs.data(), out.data_ptr(), st));
    else if (kind == 1) chk(pygim_grande_run_group(handle, ptrs.data(), lds.data(), out.data_ptr(), st));
    else chk(pygim_spmv_run_group(handle, ptrs.data(), out.data_ptr(), st));
    log_timers(handle, !out.is_cuda());
    return out;
}

#if PYGIM_VARIANT == 1
// ---- grande --------------------------------------------------------------------------------
std::vector<int64_t> dpu_init_ranks(int64_t nr_ranks) {
    std::vector<int64_t> units((size_t)std::max<int64_t>(nr_ranks, 1));
    chk(pygim_init_ranks(nr_ranks, units.data()));
    units.resize((size_t)nr_ranks);
    return units;
}
std::vector<int64_t> dpu_init_dpus(int64_t nr_dpus) {
    std::vector<int64_t> units((size_t)(nr_dpus + 7) / 8 + 1);
    int64_t ranks = 0;
    chk(pygim_init_units(nr_dpus, units.data(), &ranks));
    units.resize((size_t)ranks);
    return units;
}
int64_t spmm_csr_to_device_group(std::vector<at::Tensor> row_indices, std::vector<at::Tensor> col_indices,
                                 std::vector<at::Tensor> values, std::vector<int64_t> nrows,
                                 std::vector<int64_t> ncols, std::vector<at::Tensor> dense_cols, int64_t h_size) {
    std::vector<int64_t> n_dense, flat;
    for (auto &t : dense_cols) {
        auto c = t.to(at::kCPU).contiguous();
        const int32_t *p = c.data_ptr<int32_t>();
        n_dense.push_back(c.numel());
        for (int64_t k = 0; k < c.numel(); k++) flat.push_back(p[k]);
    }
    return to_device_group(PYGIM_CSR, row_indices, col_indices, values, nrows, ncols, n_dense, flat, h_size);
}
at::Tensor spmm_csr_run_group(int64_t handle, std::vector<at::Tensor> B_parts) { return run_common(handle, B_parts, 1); }
#else
// ---- default / spmv --------------------------------------------------------------------------
void dpu_init_ranks(int64_t nr_ranks) { chk(pygim_init_ranks(nr_ranks, nullptr)); }
void dpu_init_dpus(int64_t nr_dpus) { chk(pygim_init_units(nr_dpus, nullptr, nullptr)); }
#if PYGIM_VARIANT == 0
int64_t spmm_csr_to_device_group(std::vector<at::Tensor> row_indices, std::vector<at::Tensor> col_indices,
                                 std::vector<at::Tensor> values, std::vector<int64_t> nrows,
                                 std::vector<int64_t> ncols, std::vector<int64_t> dense_cols, int64_t h_size) {
    const size_t n = ncols.size();
    return to_device_group(PYGIM_CSR, row_indices, col_indices, values, nrows, ncols,
                           std::vector<int64_t>(n, (int64_t)dense_cols.size()), repeat(dense_cols, n), h_size);
}
int64_t spmm_coo_to_device_group(std::vector<at::Tensor> row_indices, std::vector<at::Tensor> col_indices,
                                 std::vector<at::Tensor> values, std::vector<int64_t> nrows,
                                 std::vector<int64_t> ncols, std::vector<int64_t> dense_cols, int64_t h_size) {
    const size_t n = ncols.size();
    return to_device_group(PYGIM_COO, row_indices, col_indices, values, nrows, ncols,
                           std::vector<int64_t>(n, (int64_t)dense_cols.size()), repeat(dense_cols, n), h_size);
}
at::Tensor spmm_csr_run_group(int64_t handle, std::vector<at::Tensor> B_parts) { return run_common(handle, B_parts, 0); }
at::Tensor spmm_coo_run_group(int64_t handle, std::vector<at::Tensor> B_parts) { return run_common(handle, B_parts, 0); }
#else
int64_t spmv_coo_to_device_group(std::vector<at::Tensor> row_indices, std::vector<at::Tensor> col_indices,
                                 std::vector<at::Tensor> values, std::vector<int64_t> nrows,
                                 std::vector<int64_t> ncols, std::vector<int64_t> dense_cols, int64_t h_size,
                                 int64_t ranks_per_spmv) {
    TORCH_CHECK(ranks_per_spmv >= 1, "ranks_per_spmv must be >= 1");
    const size_t n = ncols.size();
    return to_device_group(PYGIM_COO, row_indices, col_indices, values, nrows, ncols,
                           std::vector<int64_t>(n, (int64_t)dense_cols.size()), repeat(dense_cols, n), h_size);
}
at::Tensor spmv_coo_run_group(int64_t handle, std::vector<at::Tensor> B_parts) { return run_common(handle, B_parts, 2); }
#endif
#endif

#if PYGIM_VARIANT != 2
// ---- MatrixMarket debug ops (spmm_default/utils.hpp:139-173; the grande library registers the same five from its own copy of
// that header, spmm_grande/pytorch_api.cpp:338-342) -----------------------------------------------
// Same result as the reference reader + coo2csr (utils.hpp:15-127): '%' lines skipped, the size line
// gives rows / columns / entries, both dimensions padded to even, indices 1-based in the file, the
// value column ignored (every stored value is 1), entries of a row kept in file order.
struct MtxCsr {
    int64_t nrows = 0, ncols = 0;
    std::vector<int32_t> rowptr, colind, values;
};
MtxCsr read_mtx(const std::string &path) {
    std::ifstream in(path);
    TORCH_CHECK(in.good(), "cannot open ", path);
    MtxCsr m;
    std::vector<int32_t> r, c;
    std::string line;
    bool sized = false;
    int64_t nnz = 0;
    while (std::getline(in, line)) {
        const size_t b = line.find_first_not_of(" \t\r");
        if (b == std::string::npos || line[b] == '%') continue;
        const char *p = line.c_str() + b;
        char *q = nullptr;
        const long a0 = std::strtol(p, &q, 10);
        const long a1 = std::strtol(q, &q, 10);
        if (!sized) {
            nnz = std::strtol(q, &q, 10);
            m.nrows = a0 + (a0 & 1);
            m.ncols = a1 + (a1 & 1);
            r.reserve((size_t)nnz);
            c.reserve((size_t)nnz);
            sized = true;
        } else if ((int64_t)r.size() < nnz) {
            r.push_back((int32_t)(a0 - 1));
            c.push_back((int32_t)(a1 - 1));
        }
    }
    TORCH_CHECK(sized, "no size line in ", path);
    TORCH_CHECK((int64_t)r.size() == nnz, path, ": size line promises ", nnz, " entries, file holds ", r.size());
    m.rowptr.assign((size_t)m.nrows + 1, 0);
    for (size_t k = 0; k < r.size(); k++) {
        TORCH_CHECK(r[k] >= 0 && r[k] < m.nrows && c[k] >= 0 && c[k] < m.ncols, path, ": entry ", k, " out of range");
        m.rowptr[(size_t)r[k] + 1]++;
    }
    for (int64_t i = 0; i < m.nrows; i++) m.rowptr[(size_t)i + 1] += m.rowptr[(size_t)i];
    std::vector<int32_t> next(m.rowptr.begin(), m.rowptr.end() - 1);
    m.colind.resize(r.size());
    for (size_t k = 0; k < r.size(); k++) m.colind[(size_t)next[(size_t)r[k]]++] = c[k];
    m.values.assign(r.size(), 1);
    return m;
}
at::Tensor as_i32(const std::vector<int32_t> &v) {
    at::Tensor t = at::empty({(int64_t)v.size()}, at::kInt);
    if (!v.empty()) std::memcpy(t.data_ptr<int32_t>(), v.data(), v.size() * sizeof(int32_t));
    return t;
}
at::Tensor read_matrix_rowptr(std::string f) { return as_i32(read_mtx(f).rowptr); }
at::Tensor read_matrix_colind(std::string f) { return as_i32(read_mtx(f).colind); }
at::Tensor read_matrix_values(std::string f) { return as_i32(read_mtx(f).values); }
int64_t read_matrix_nrows(std::string f) { return read_mtx(f).nrows; }
int64_t read_matrix_ncols(std::string f) { return read_mtx(f).ncols; }
#endif

}  // namespace

TORCH_LIBRARY(pim_ops, m) {
    m.def("dpu_init_ranks", &dpu_init_ranks);
    m.def("dpu_init_dpus", &dpu_init_dpus);
    m.def("dpu_release", &dpu_release);
    m.def("spmm_free_group", &spmm_free_group);
#if PYGIM_VARIANT == 2
    m.def("spmv_coo_to_device_group(Tensor[] row_indices, Tensor[] col_indices, Tensor[] values, int[] nrows, "
          "int[] ncols, int[] dense_cols, int h_size, int ranks_per_spmv=1) -> int", &spmv_coo_to_device_group);
    m.def("spmv_coo_run_group", &spmv_coo_run_group);
#else
    m.def("spmm_csr_to_device_group", &spmm_csr_to_device_group);
    m.def("spmm_csr_run_group", &spmm_csr_run_group);
#if PYGIM_VARIANT == 0
    m.def("spmm_coo_to_device_group", &spmm_coo_to_device_group);
    m.def("spmm_coo_run_group", &spmm_coo_run_group);
#endif
    m.def("read_matrix_rowptr", &read_matrix_rowptr);
    m.def("read_matrix_colind", &read_matrix_colind);
    m.def("read_matrix_values", &read_matrix_values);
    m.def("read_matrix_nrows", &read_matrix_nrows);
    m.def("read_matrix_ncols", &read_matrix_ncols);
#endif
}
