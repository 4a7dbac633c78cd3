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

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                           \
            return fail(PYGIM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));     \
        }                                                                                      \
    } while (0)

size_t dtype_size(int dt) {
    switch (dt) {
        case PYGIM_INT8: return 1;
        case PYGIM_INT16: return 2;
        case PYGIM_INT32: return 4;
        case PYGIM_INT64: return 8;
        case PYGIM_FLT32: return 4;
        case PYGIM_DBL64: return 8;
    }
    return 0;
}

struct Tunables {
    int64_t long_row_threshold = 4096;  // rows with more entries are cut into segments
    int64_t long_segment = 512;         // entries per segment (one wave each) of such a row
    int64_t force_vec_bytes = 0;        // 0 = pick by alignment
    int64_t csr_kernel = 0;             // 0 = auto, 1 = force wide, 2 = force sub-wave
    int64_t coo_chunk = 512;            // entries per wave in the nnz-split COO kernel
    int64_t coo_via_rowptr = 0;         // 1 = run COO groups through the CSR kernels (derived rowptr)
    int64_t panel_mode = 0;             // 0 = auto (cost rule), 1 = force the L2-blocked panel kernel, 2 = never
    int64_t panel_bytes = 4 << 20;      // L2 budget of one (column panel x 128-byte feature slice)
    int64_t panel_min_seg = 8;          // auto: least average entries per (row, panel) worth a panel sweep
    int64_t panel_block = 256;          // threads per block of the sweep kernel (64, 128 or 256)
    int64_t panel_lds_pad = 0;          // (experiment) unused dynamic LDS per sweep block: fewer resident blocks per CU
    int64_t panel_coop = 512;           // items longer than this are walked by a whole wave (8 lane groups)
    int64_t panel_pack = 1;             // 1 = gather from a slice-major copy of X (made per product)
    int64_t slice_group_bytes = 640ll << 20;  // sweep at most this many bytes of slice-major X per launch (0 = all slices at once)
    int64_t vec_kernel = 1;             // 1 = rows of X of at most 4 elements (SpMV) take the CSR-vector kernel
    int64_t vec_lds = 1;                // 1 = ... and, when a column panel of X fits, the LDS-staged form of it (k_spmv_lds)
    int64_t vec_lds_min_seg = 14;       // ... from this many entries per (row, panel) on average
    int64_t merge_parts = 1;            // 1 = groups of several sparse parts also get the merged matrix (used by the run entry points)
    int64_t narrow_vals = 1;            // 1 = INT64 / DBL64 values that are all exactly 4-byte representable are streamed as int32 / float by the sweep
    int64_t split_unit_pattern = 1;     // 1 = integer weights that are 1 almost everywhere: unit pattern + a small correction part
    int64_t panel_locality = 1;         // (2 = always, whatever the size or the labels' quality: tests) 1 = the sweep's work items in LOCALITY order (blocks of 2048 rows of the similarity / id order, longest first inside a block) when the graph has structure; 0 = by length alone
    int64_t panel_col16 = 1;            // 1 = the sweep reads 16-bit panel-local column ids (built with the plan)
    int64_t fuse_windows = 1;           // 1 = the dense windows of a sparse part become ONE block product of the full width
    int64_t kernel_events = 0;          // 1 = bracket the dominant kernel of every block product with HIP events
    int64_t lds_mode = 0;               // LDS-staged product (k_lds_spmm): 0 = auto (reuse rule), 1 = whenever a part has the plan, 2 = never
    int64_t lds_min_reuse_x100 = 75;    // auto: least stored entries per staged column of X (x 100) for the LDS-staged product (measured crossing: between 0.48 and 0.96, profiles/r04_exp_share.txt; round 3's kernel: 1.1)
    int64_t lds_min_width = 33;         // narrower products keep the sweep (a 64-feature slice would be mostly padding)
    int64_t lds_threads = 0;            // host threads of the schedule builder (0 = all)
    int64_t lds_waves = 16;             // waves per workgroup of the LDS-staged kernel the plan is made for (8 or 16)
    int64_t lds_long_slots = 128;       // tokens per (wave, chunk) from which the 16-token-batch geometry is planned (0 = never)
    int64_t lds_code = 1;               // 1 = FLT32 / INT32 unit-weight plans are also compiled into machine code (lds_plan.hpp lds_code_from_plan) and run by k_lds_code_*; 0 = the token kernels
    int64_t lds_col_split = 0;          // column-split workgroup tiles for short row shares: 0 = automatic (integers; FLT32 with lds_col_split_f32), 1 = never, S > 1 = S ranges
    int64_t lds_col_split_f32 = 0;      // 1 = FLT32 shares may be split too (a row's sum is then the sum of its column ranges' sums: the norm-wise contract, not the bit-identical one)
    int64_t lds_code_nbuf = 0;          // chunk buffers of a code-stream plan: 0 = by the product's width (3 x 192 columns up to two slices, else 2 x 320), 2, 3
    int64_t lds_code_waves = 0;         // waves per workgroup of a code-stream plan: 16 (x 96 accumulators), 8 (x 228: taller tiles, fewer rounds of workgroups), 0 = automatic
    int64_t lds_code_kc = 0;            // columns per chunk of a code-stream plan (0 = by the ring: 320 / 192 / 160 / 128 for 2 / 3 / 4 / 5 buffers)
    int64_t lds_code_gsize = 0;         // staged columns per group of reads of a code stream (0 = lds_plan.hpp lds_code_regs)
    int64_t lds_code_nsets = 0;         // x-register sets of a code stream: the reads run nsets - 1 groups ahead of the adds (0 = default)
    int64_t lds_round_tiles = 1;        // 1 = tile height chosen so that tiles x slices fill whole rounds of workgroups
    int64_t lds_tile_order = 2;         // code-stream plans: which rows share a tile -- 0 = consecutive rows, 1 = similarity order (label propagation, lds_reorder_dev.hpp), 2 = automatic (similarity for square parts of >= 1 M entries)
    int64_t lds_lp_rounds = 6;          // rounds of the label propagation
    int64_t lds_codegen = 1;            // code streams: 1 = generated on the device from the resident CSR (lds_codegen_dev.hpp), 0 = by the host encoder, 2 = on the device AND checked word for word against the host encoder (tests)
    int64_t lds_xcd_slices = 0;         // code-stream kernels: slices of X per XCD (0 = automatic; 1 = an XCD streams one slice; 2 / 4: a tile's slices side by side on one XCD share its code in L2)
    int64_t lds_code_boundary = 0;      // rings of >= 3 buffers: 0 / 1 = the workgroup meets at the slot boundary (one more chunk in flight, the last group's adds cross the barrier), 2 = in the middle of a slot
    int64_t lds_code_exp = 0;           // (timing experiments, WRONG results) code streams without barriers (1) / without the chunk DMA (2)
    int64_t lds_fail = 0;               // (tests) force a step of the code-stream set-up to fail: 1 = code generation, 2 = executable memory, 4 = schedule build
    int64_t lds_ablate = 0;             // (timing experiments) 1..4: kernel variants with parts of the loop removed -- WRONG results
} g_tune;

struct LongPlan {
    uint32_t n_long = 0, n_tasks = 0, thresh = 0;
    uint32_t *d_tasks = nullptr;  // (row, s, e) x n_tasks
    uint32_t *d_desc = nullptr;   // (row, first_task, n_tasks) x n_long
};

struct Part {
    int64_t nrows = 0, ncols = 0, nnz = 0;
    uint32_t *rowptr = nullptr;  // CSR rowptr, or derived from the COO row index
    uint32_t *rowind = nullptr;  // COO only
    uint32_t *colind = nullptr;
    void *vals = nullptr;        // nullptr when all ones
    void *vals_narrow = nullptr; // 8-byte element types whose values all fit 4 bytes exactly: the sweep streams these (narrow_values)
    bool own_rowptr = false, own_rowind = false, own_colind = false, own_vals = false;
    // long-row plans (rows cut into segments over many waves).  lp_base: used by the row-per-wave kernels; lp_panel: used with the panel sweep, where a row is
    // only "long" when its share of ONE panel would be (threshold x npanels): segment kernels gather
    // whole rows of X past the L2 blocking, so the panel sweep keeps as many rows as it can
    LongPlan lp_base, lp_panel;
    // L2-blocked plan: per column panel, the list of (row, first entry, length|FIRST) work items
    // sorted by length (rows without entries in a panel do not appear in its list)
    uint32_t *d_items = nullptr;            // [3][n_items]: rows | begins | lens
    size_t n_items = 0;
    std::vector<size_t> panel_off;          // npanels + 1 offsets into the item arrays
    std::vector<uint32_t> panel_coop;       // per panel: leading items long enough for the wave-cooperative mode
    uint32_t npanels = 0, panel_cols = 0;
    unsigned short *col16 = nullptr;        // panel-local 16-bit column ids (panels of <= 65536 columns), same order as colind
    std::vector<uint32_t> panel_long128;    // per panel: leading items with more than 256 / 128 / 64 / 32 entries, 4 counts each (the
                                            // LDS-staged SpMV kernel's length classes)
    std::vector<uint64_t> panel_nnz;        // per panel: entries of its items
    SpmvUnit *d_spmv_units = nullptr;       // the LDS-staged SpMV kernel's (panel, slot) table, built at its first launch
    uint32_t n_spmv_units = 0;
    std::vector<int64_t> dense_cols;
    // integer weights that are 1 almost everywhere (a coalesced multigraph, backend_pim/spmm.py:40-42): this part keeps
    // the PATTERN only (unit weights: no value array, no multiplies, packed 8/16-bit sums) and `extra` holds the few
    // entries with weight v != 1 as (v - 1); A.X = pattern.X + extra.X exactly in modular arithmetic
    std::unique_ptr<Part> extra;
    // LDS-staged product (lds_plan.hpp): token streams and tile table on the device; lds_tiles == nullptr: no such plan
    uint32_t *lds_tok = nullptr, *lds_nb = nullptr, *lds_chunks = nullptr, *lds_rowmap = nullptr;
    LdsTile *lds_tiles = nullptr;
    int cols_sorted = -1;   // stored order inside every row is column order: -1 = not checked yet, 0 / 1
    uint32_t lds_ntiles = 0, lds_nw = 8, lds_batch = 8, lds_wdelta = 0;   // lds_wdelta != 0: the plan carries the entries' values
    char *lds_code = nullptr;              // the schedule as gfx950 machine code (EXECUTABLE device memory from the HSA pool), or nullptr
    uint64_t *lds_code_start = nullptr;    // [ntiles][16]: byte offset of a (tile, wave) stream
    uint64_t lds_code_bytes = 0, lds_code_pairs = 0;
    uint32_t lds_col_splits = 1;           // > 1: the plan's tiles are (row tile, column range) pairs writing partial sums (launch_lds reduces them)
    uint32_t lds_code_piece = 0;           // bytes of a chunk one wave DMAs (the code plan's ring geometry)
    uint32_t lds_kc = 0, lds_nbuf = 0;     // the plan's ring: columns per chunk, buffers
    uint32_t lds_row_bytes = 256, lds_ka = 0;   // bytes of a staged row (512: the 8-byte element form), accumulators (rows) per wave
    uint32_t lds_code_gsize = 0, lds_code_nsets = 0;
    uint64_t lds_code_shared = 0;          // entries of the code stream served by another entry's read
    std::string lds_note;                  // which form of the product this part got, and why not a faster one
    bool lds_codegen_device = false;       // its code stream was generated on the device
    // which rows are alike (round 5, lds_reorder_dev.hpp): found once per part, used by the LDS plan's tiles and by the sweep's item order
    bool panel_locality_used = false;      // the sweep's items are in locality order
    int sim_kind = -1;                     // -1 = not looked at, 0 = no structure found, 1 = the stored ids are local already (consecutive order), 2 = label propagation
    std::vector<uint32_t> sim_order;       // kind 2: the rows ordered by (label, id)
    uint32_t sim_labels = 0, sim_largest = 0;
    double sim_agree = 0.0;                // kind 2: share of the stored entries whose column carries its row's label
    std::string sim_why;
    uint32_t lds_tile_labels = 0, lds_tile_largest = 0;   // similarity tiles: labels the propagation ended with, rows of the largest (0 = consecutive rows)
    std::string lds_codegen_why;           // ... or why not
    bool lds_is_code = false;              // the LDS plan of this part is in the code-stream geometry (three 192-column buffers): k_lds_code_* only
    uint64_t lds_slots = 0, lds_tokens = 0;   // 80 KiB chunk fills per slice and product; tokens incl. padding
    bool is_extra = false;  // widths of the dense parts paired with this part
};

struct Group {
    int format = 0, dtype = 0;
    int64_t h = 0, total_rows = 0, total_cols = 0;
    std::vector<Part> parts;
    // the sparse parts are column blocks of ONE matrix (spmm.py:127-136) whose partial products are summed: merged here
    // (global column ids, rows concatenated in block order = sorted) so that the group product is one sweep with the
    // plan that suits the whole matrix, whatever sp_parts the caller chose (32 parts of Reddit: 10.3 -> 6.8 ms)
    std::unique_ptr<Part> merged;
    bool all_ones = true;
    // scratch (device), grown on demand
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    void *stage_in = nullptr;
    size_t stage_in_bytes = 0;
    void *stage_out = nullptr;
    size_t stage_out_bytes = 0;
    void *xq = nullptr;       // quantised features / integer result of pygim_quant_spmm_run
    size_t xq_bytes = 0;
    void *oq = nullptr;
    size_t oq_bytes = 0;
    // slice-major copy made (or reused) by the block product in flight: its correction part (Part::extra) gathers from
    // the same copy instead of repeating it
    const void *packed_src = nullptr;
    void *packed_buf = nullptr;
    int64_t packed_ld = 0, packed_w = 0;
    // per-call options of the entry point in flight (a group serves one call at a time)
    bool x_unchanged = false;        // the caller vouches: same X, same contents as the product that last packed it
    const void *pre_xs = nullptr;    // slice-major copy already made by the caller of launch_block_any (fused quantiser)
    float *deq_out = nullptr;        // fused dequantisation: rows' LAST items store float(sum) * scale here
    int64_t deq_ld = 0;
    const uint32_t *deq_amax = nullptr;
    int deq_log2 = 0;
    const float *post_mul = nullptr, *post_add = nullptr;  // per-column epilogue of the fused store (nullptr = none)
    int post_relu = 0;
    // pinned pointer tables of the SpMV pack (two slots, each guarded by an event)
    void **h_ptrs = nullptr;
    size_t h_ptrs_n = 0;
    hipEvent_t ev_ptrs[2] = {nullptr, nullptr};
    int ptr_slot = 0;
    void *xcat = nullptr;     // dense windows of one sparse part laid side by side (fused block product)
    size_t xcat_bytes = 0;
    void **d_ptrs = nullptr;  // device array of pointers (spmv pack)
    size_t d_ptrs_n = 0;
    int *d_flags = nullptr;
    // long rows run beside the main sweep on a forked stream (fork/join with events)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    double timers[5] = {0, 0, 0, 0, 0};
    // HIP-event pairs around the dominant kernel (tunable kernel_events), resolved on query
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
    bool kernel_events = false;  // pygim_group_kernel_events
    double ev_ms = 0;
    int64_t ev_count = 0;
};

struct Context {
    bool inited = false;
    int device = 0;
    int cu_count = 0;
    int64_t nr_ranks = 0;
    std::set<Group *> groups;
    std::mutex mu;
    // slice-major copies of X for the panel sweep: one buffer per (device, launch stream) -- products on different
    // streams never share a buffer -- each with a record of what it holds, for callers that pass x_unchanged.
    // At most XS_MAX buffers are kept (least recently used goes first); all are freed with the last group.
    struct XsBuf {
        void *ptr = nullptr;
        size_t bytes = 0;
        const void *src = nullptr;  // what the buffer holds: X pointer, stride, rows, width, element size
        int64_t ld = 0, rows = 0, w = 0;
        uint64_t rows_pad = 0;  // kind 1 / 3: rows of a slice in THIS copy (whole chunks of the part that packed it) -> its slice stride
        size_t es = 0;
        int kind = 0;  // 0 = 128-byte slices (sweep), 1 = 256-byte slices, rows padded to whole chunks (LDS-staged product)
        uint64_t stamp = 0;
        int in_use = 0;  // handed to a call that has not enqueued its kernels yet: not an eviction victim
    };
    static constexpr size_t XS_MAX = 4;
    std::map<std::pair<int, hipStream_t>, XsBuf> xs_bufs;
    uint64_t xs_clock = 0;
    int64_t generation = 0;  // bumped by pygim_release: handles made before it are dead even if an address comes back
} g_ctx;

void free_xs_buffers_locked() {
    for (auto &kv : g_ctx.xs_bufs)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    g_ctx.xs_bufs.clear();
}

// the slice-major buffer of (current device, stream), at least `need` bytes; evicts the least recently used buffer
// when more than XS_MAX exist.  Caller holds g_ctx.mu.
int xs_buffer_locked(hipStream_t st, size_t need, Context::XsBuf **out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_pair(dev, st);
    if (!g_ctx.xs_bufs.count(key) && g_ctx.xs_bufs.size() >= Context::XS_MAX) {
        auto victim = g_ctx.xs_bufs.end();
        for (auto it = g_ctx.xs_bufs.begin(); it != g_ctx.xs_bufs.end(); ++it)
            if (it->second.in_use == 0 && (victim == g_ctx.xs_bufs.end() || it->second.stamp < victim->second.stamp)) victim = it;
        if (victim != g_ctx.xs_bufs.end()) {  // (all busy: grow past the cap rather than pull a buffer from under a caller)
            if (victim->second.ptr) (void)hipFree(victim->second.ptr);  // hipFree waits for the work that uses it
            g_ctx.xs_bufs.erase(victim);
        }
    }
    Context::XsBuf &b = g_ctx.xs_bufs[key];
    if (b.bytes < need) {
        if (b.ptr) (void)hipFree(b.ptr);
        b = Context::XsBuf();
        if (hipMalloc(&b.ptr, need) != hipSuccess) {
            (void)hipGetLastError();
            g_ctx.xs_bufs.erase(key);
            return fail(PYGIM_ERR_HIP, "slice-major buffer alloc");
        }
        b.bytes = need;
    }
    b.stamp = ++g_ctx.xs_clock;
    b.in_use++;
    *out = &b;
    return 0;
}

// a slice-major buffer stays pinned (not evictable by other threads' calls) until the kernels that read it are enqueued;
// after that hipFree's implicit wait protects it
struct XsPin {
    void *ptr = nullptr;
    void hold(void *p) { ptr = p; }
    ~XsPin() {
        if (!ptr) return;
        std::lock_guard<std::mutex> lk(g_ctx.mu);
        for (auto &kv : g_ctx.xs_bufs)
            if (kv.second.ptr == ptr && kv.second.in_use > 0) kv.second.in_use--;
    }
};

bool is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int ensure(void **buf, size_t *have, size_t need) {
    if (*have >= need) return 0;
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(buf, need));
    *have = need;
    return 0;
}

// bring an array to the device (copy if it is host memory, alias if already there)
template <typename P> int to_device(const void *src, size_t bytes, P **dst, bool *owned, hipStream_t st) {
    if (bytes == 0) {
        // keep a valid non-null pointer so kernels can take it
        HIP_TRY(hipMalloc((void **)dst, 16));
        *owned = true;
        return 0;
    }
    if (is_device_ptr(src)) {
        *dst = (P *)const_cast<void *>(src);
        *owned = false;
        return 0;
    }
    HIP_TRY(hipMalloc((void **)dst, bytes));
    *owned = true;
    HIP_TRY(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st));
    return 0;
}

// Executable device memory for the code-stream kernels: hipMalloc memory is not executable (an instruction fetch from it faults),
// the HSA runtime's coarse-grained GPU pool with HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG is (it is what the loader puts kernels in).
struct ExecPool {
    bool tried = false, ok = false;
    hsa_agent_t agent{};
    hsa_amd_memory_pool_t pool{};
};
static std::map<int, ExecPool> g_exec_pools;   // per HIP device ordinal (guarded by g_ctx.mu)
static ExecPool *exec_pool_locked(int dev) {
    ExecPool &ep = g_exec_pools[dev];
    if (ep.tried) return ep.ok ? &ep : nullptr;
    ep.tried = true;
    if (hsa_init() != HSA_STATUS_SUCCESS) return nullptr;   // (reference-counted: HIP holds the runtime open already)
    // the HSA agent of this HIP device: by PCI address (HIP_VISIBLE_DEVICES renumbers HIP's devices, not the runtime's agents);
    // by ordinal only when the address cannot be had
    struct Find { int want, seen; int bdf, domain; hsa_agent_t agent, nth; bool found, have_nth; } f{dev, 0, -1, -1, {}, {}, false, false};
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
        f.bdf = (prop.pciBusID << 8) | (prop.pciDeviceID << 3);
        f.domain = prop.pciDomainID;
    }
    hsa_iterate_agents([](hsa_agent_t a, void *d) -> hsa_status_t {
        Find *f = (Find *)d;
        hsa_device_type_t t;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
        if (f->seen++ == f->want) { f->nth = a; f->have_nth = true; }
        uint32_t bdf = 0, domain = 0;
        if (f->bdf >= 0 && hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
            hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain) == HSA_STATUS_SUCCESS &&
            (int)(bdf & ~7u) == f->bdf && (int)domain == f->domain) {
            f->agent = a;
            f->found = true;
            return HSA_STATUS_INFO_BREAK;
        }
        return HSA_STATUS_SUCCESS;
    }, &f);
    if (!f.found) {
        if (!f.have_nth) return nullptr;
        f.agent = f.nth;
    }
    struct FindPool { hsa_amd_memory_pool_t pool; bool found; } fp{{}, false};
    hsa_amd_agent_iterate_memory_pools(f.agent, [](hsa_amd_memory_pool_t p, void *d) -> hsa_status_t {
        FindPool *fp = (FindPool *)d;
        hsa_amd_segment_t seg;
        uint32_t flags = 0;
        bool alloc = false;
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
        if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) {
            fp->pool = p;
            fp->found = true;
            return HSA_STATUS_INFO_BREAK;
        }
        return HSA_STATUS_SUCCESS;
    }, &fp);
    if (!fp.found) return nullptr;
    ep.agent = f.agent;
    ep.pool = fp.pool;
    ep.ok = true;
    return &ep;
}
// executable device memory from the HSA pool (hipMalloc memory faults on instruction fetch); nullptr + *why when there is none
static void *exec_alloc(size_t bytes, std::string *why = nullptr) {
    auto say = [&](const char *m) { if (why) *why = m; };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { say("no current HIP device"); return nullptr; }
    ExecPool *ep;
    {
        std::lock_guard<std::mutex> lk(g_ctx.mu);
        ep = exec_pool_locked(dev);
    }
    if (!ep) { say("the HSA runtime offers no executable device memory pool"); return nullptr; }
    void *ptr = nullptr;
    if (hsa_amd_memory_pool_allocate(ep->pool, bytes, HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG, &ptr) != HSA_STATUS_SUCCESS) {
        say("the executable pool could not allocate the code");
        return nullptr;
    }
    return ptr;
}
static void *exec_alloc_upload(const void *host, size_t bytes, std::string *why = nullptr) {
    auto say = [&](const char *m) { if (why) *why = m; };
    // the code goes up through HIP (a bounded staging buffer and a copy kernel: the executable allocation is a device address like any
    // other inside a kernel); the HSA runtime is asked for the memory only
    const size_t piece = std::min<size_t>(bytes, (size_t)64 << 20);   // 64 MiB at a time: no second full-size allocation beside the code
    void *ptr = exec_alloc(bytes, why), *stage = nullptr;
    if (!ptr) return nullptr;
    if (hipMalloc(&stage, std::max<size_t>(piece, 256)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hsa_amd_memory_pool_free(ptr);
        say("out of device memory for the staging buffer of the code upload");
        return nullptr;
    }
    bool ok = true;
    for (size_t off = 0; off < bytes && ok; off += piece) {
        const size_t nb = std::min(piece, bytes - off);
        ok = hipMemcpy(stage, (const char *)host + off, nb, hipMemcpyHostToDevice) == hipSuccess;
        const uint64_t n16 = nb / 16;   // (the code blob is a multiple of 256 bytes)
        if (ok && n16) {
            hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const u32x4_t *)stage, (u32x4_t *)((char *)ptr + off), n16);
            ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;   // (the staging buffer is re-used)
        }
    }
    (void)hipFree(stage);
    if (!ok) {
        (void)hsa_amd_memory_pool_free(ptr);
        say("copying the code to the device failed");
        return nullptr;
    }
    return ptr;
}

void free_part(Part &p) {
    if (p.own_rowptr && p.rowptr) (void)hipFree(p.rowptr);
    if (p.own_rowind && p.rowind) (void)hipFree(p.rowind);
    if (p.own_colind && p.colind) (void)hipFree(p.colind);
    if (p.own_vals && p.vals) (void)hipFree(p.vals);
    if (p.vals_narrow) (void)hipFree(p.vals_narrow);
    p.vals_narrow = nullptr;
    for (LongPlan *lp : {&p.lp_base, &p.lp_panel}) {
        if (lp->d_tasks) (void)hipFree(lp->d_tasks);
        if (lp->d_desc) (void)hipFree(lp->d_desc);
    }
    if (p.d_items) (void)hipFree(p.d_items);
    if (p.col16) (void)hipFree(p.col16);
    if (p.d_spmv_units) (void)hipFree(p.d_spmv_units);
    for (void *q : {(void *)p.lds_tok, (void *)p.lds_nb, (void *)p.lds_chunks, (void *)p.lds_rowmap, (void *)p.lds_tiles})
        if (q) (void)hipFree(q);
    p.lds_tok = p.lds_nb = p.lds_chunks = p.lds_rowmap = nullptr;
    p.lds_tiles = nullptr;
    if (p.lds_code) (void)hsa_amd_memory_pool_free(p.lds_code);
    if (p.lds_code_start) (void)hipFree(p.lds_code_start);
    p.lds_code = nullptr;
    p.lds_code_start = nullptr;
    p.lds_is_code = false;
    p.lds_row_bytes = 256;
    p.lds_col_splits = 1;
    if (p.extra) free_part(*p.extra);
}

void free_group(Group *g) {
    for (auto &p : g->parts) free_part(p);
    if (g->merged) free_part(*g->merged);
    if (g->scratch) (void)hipFree(g->scratch);
    if (g->stage_in) (void)hipFree(g->stage_in);
    if (g->stage_out) (void)hipFree(g->stage_out);
    if (g->xcat) (void)hipFree(g->xcat);
    if (g->xq) (void)hipFree(g->xq);
    if (g->oq) (void)hipFree(g->oq);
    if (g->d_ptrs) (void)hipFree(g->d_ptrs);
    if (g->h_ptrs) (void)hipHostFree(g->h_ptrs);
    for (hipEvent_t e : g->ev_ptrs)
        if (e) (void)hipEventDestroy(e);
    if (g->d_flags) (void)hipFree(g->d_flags);
    if (g->side) (void)hipStreamDestroy(g->side);
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    if (g->ev_join) (void)hipEventDestroy(g->ev_join);
    for (auto &e : g->ev_pending) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete g;
}

Group *lookup(int64_t handle) {
    Group *g = reinterpret_cast<Group *>(static_cast<uintptr_t>(handle));
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    return g_ctx.groups.count(g) ? g : nullptr;
}

// ---------------------------------------------------------------------------
// launch plan for one block product
// ---------------------------------------------------------------------------
int pick_vec_bytes(size_t es, const void *X, int64_t ldx, const void *C, int64_t ldc) {
    // widest of {16, 8, element} bytes that every row start of X and C is aligned to
    auto ok = [&](size_t vb) {
        return ((uintptr_t)X % vb == 0) && ((uintptr_t)C % vb == 0) && ((size_t)ldx * es % vb == 0) &&
               ((size_t)ldc * es % vb == 0);
    };
    size_t cap = g_tune.force_vec_bytes > 0 ? (size_t)g_tune.force_vec_bytes : 16;
    for (size_t vb : {(size_t)16, (size_t)8})
        if (vb <= cap && vb > es && ok(vb)) return (int)vb;
    return (int)es;
}

struct KernelTimer {
    Group *g;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    KernelTimer(Group *g_, hipStream_t st_, bool on = true) : g(g_), st(st_) {
        if (on && (g_tune.kernel_events || g->kernel_events) && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
            (void)hipEventRecord(a, st);
        else
            a = b = nullptr;
    }
    void stop() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        g->ev_pending.emplace_back(a, b);
        a = b = nullptr;
    }
};

// Does this block product take the panel sweep?  (16-byte pieces at any alignment and any width; narrow or ragged rows
// whose tail piece would reach past the row stride gather from the padded slice-major copy, which is skipped when
// that copy would be enormous.)
template <typename T>
bool want_panel(const Part &p, uint32_t w, int64_t ldx) {
    if (p.d_items == nullptr || g_tune.panel_mode == 2 || p.nrows == 0 || w == 0) return false;
    if (g_tune.panel_mode == 0 && !(p.npanels == 1 || (double)p.nnz / ((double)p.nrows * p.npanels) >= (double)g_tune.panel_min_seg))
        return false;
    constexpr uint32_t V = 16 / sizeof(T), F = V * 8;
    const uint32_t nslices = (w + F - 1) / F;
    const bool tail_inside = (w % V) == 0;  // no partial 16-byte piece: nothing is read past a row's width
    (void)ldx;
    if (!tail_inside && (uint64_t)p.ncols * nslices * 128ull > (8ull << 30)) return false;
    return true;
}

// element types of the conv layers' quantiser (models/quantize.py:22-30): the fused store exists for these
template <typename T> struct DeqType { static constexpr bool ok = false; };
template <> struct DeqType<int8_t> { static constexpr bool ok = true; };
template <> struct DeqType<int16_t> { static constexpr bool ok = true; };
template <> struct DeqType<int32_t> { static constexpr bool ok = true; };
template <> struct DeqType<float> { static constexpr bool ok = true; };

template <typename T, int VEC>
int launch_block_t(Group *g, Part &p, const T *X, int64_t ldx, T *C, int64_t ldc, uint32_t w, bool accumulate,
                   hipStream_t st) {
    const T *vals = (const T *)p.vals;
    const uint32_t nrows = (uint32_t)p.nrows;
    const uint32_t lanes_needed = (w + VEC - 1) / VEC;
    // L2-blocked panel sweep or row-per-wave kernels?
    bool use_panel = false;
    if constexpr (VEC * sizeof(T) == 16) use_panel = want_panel<T>(p, w, ldx);
    // COO groups: the panel sweep (through the row pointers derived at create time) when it pays,
    // else the native equal-nnz kernel
    const bool coo_native = (g->format == PYGIM_COO) && p.rowind != nullptr && !use_panel && !g_tune.coo_via_rowptr && lanes_needed > 16;
    if (coo_native) {
        // nnz-split kernel + carry fix-up
        if (!accumulate && (uint64_t)nrows * w > 0)
            hipLaunchKernelGGL((k_zero_rows<T>), dim3((unsigned)(((uint64_t)nrows * w + 255) / 256)), dim3(256), 0, st, C, ldc,
                               (uint64_t)nrows, w);
        if (p.nnz == 0) return 0;
        const uint32_t chunk = (uint32_t)g_tune.coo_chunk;
        const uint32_t nchunks = (uint32_t)((p.nnz + chunk - 1) / chunk);
        const size_t need = (size_t)nchunks * 2 * w * sizeof(T);
        if (int rc = ensure(&g->scratch, &g->scratch_bytes, need)) return rc;
        dim3 grid((nchunks + 3) / 4, (lanes_needed + 63) / 64);
        KernelTimer kt(g, st, !p.is_extra);
        hipLaunchKernelGGL((k_coo_wide<T, VEC>), grid, dim3(256), 0, st, p.rowind, p.colind, vals,
                           (uint32_t)p.nnz, chunk, X, ldx, C, ldc, w, (T *)g->scratch, accumulate ? 1 : 0);
        kt.stop();
        hipLaunchKernelGGL((k_coo_fixup<T>), dim3((nchunks + 3) / 4), dim3(256), 0, st, p.rowind,
                           (uint32_t)p.nnz, chunk, nchunks, (const T *)g->scratch, C, ldc, w, accumulate ? 1 : 0);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const LongPlan &lp = use_panel ? p.lp_panel : p.lp_base;
    // long rows: fixed-size segments on a forked stream, so that their few, long-running waves
    // overlap the main sweep instead of trailing it
    bool forked = false;
    if (lp.n_tasks > 0) {
        const size_t need = (size_t)lp.n_tasks * w * sizeof(T);
        if (int rc = ensure(&g->scratch, &g->scratch_bytes, need)) return rc;
        HIP_TRY(hipEventRecord(g->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(g->side, g->ev_fork, 0));
        dim3 grid((lp.n_tasks + 3) / 4, (lanes_needed + 63) / 64);
        hipLaunchKernelGGL((k_long_segments<T, VEC>), grid, dim3(256), 0, g->side, lp.d_tasks, lp.n_tasks, p.colind,
                           vals, X, ldx, (T *)g->scratch, w);
        hipLaunchKernelGGL((k_long_reduce<T>), dim3((w + 255) / 256, lp.n_long), dim3(256), 0, g->side, lp.d_desc,
                           lp.n_long, (const T *)g->scratch, C, ldc, w, accumulate ? 1 : 0);
        HIP_TRY(hipEventRecord(g->ev_join, g->side));
        forked = true;
    }
    auto join = [&]() -> int {
        if (forked) HIP_TRY(hipStreamWaitEvent(st, g->ev_join, 0));
        HIP_TRY(hipGetLastError());
        return 0;
    };
    // L2-blocked panel sweep (fast path for wide feature rows; see k_csr_panel)
    if constexpr (VEC * sizeof(T) == 16) {
        if (use_panel) {
            constexpr int LOG_LPR = 3;
            constexpr uint32_t F = VEC << LOG_LPR;            // elements per 128-byte slice
            const uint32_t nslices = (w + F - 1) / F;
            const uint32_t bthreads = (g_tune.panel_block == 64 || g_tune.panel_block == 128) ? (uint32_t)g_tune.panel_block : 256u;
            const uint32_t bwaves = bthreads >> 6;
            const uint32_t rows_per_block = bwaves * (64 >> LOG_LPR);
            // gather source: slice-major copy (default) or the caller's row-major X
            const T *Xg = X;
            int64_t ldg = ldx, slice_stride = F;
            KernelTimer kt(g, st, !p.is_extra);
            XsPin pin;
            // (rows of one slice that are already contiguous lines need no copy -- as long as no 16-byte piece is
            // partial: X may be a window that ends at the end of an allocation (pygim_block_run on x + f0), so a
            // piece over a ragged tail must never be read from the caller's matrix; those go to the padded copy)
            const bool tail_inside = (w % VEC) == 0;
            if (g->pre_xs) {
                // the caller packed (and quantised) the features slice-major on this stream already
                Xg = (const T *)g->pre_xs;
                ldg = F;
                slice_stride = (int64_t)p.ncols * F;
            } else if ((g_tune.panel_pack || !tail_inside) && !(nslices == 1 && (size_t)ldx * sizeof(T) <= 128 && tail_inside)) {
                const size_t need = (size_t)p.ncols * nslices * F * sizeof(T);
                void *xs_use = nullptr;
                if (p.is_extra && g->packed_buf && g->packed_src == (const void *)X && g->packed_ld == ldx &&
                    g->packed_w == (int64_t)w) {
                    xs_use = g->packed_buf;  // the pattern product of this very call packed it on this stream
                } else {
                    std::lock_guard<std::mutex> lk(g_ctx.mu);
                    if (g->x_unchanged) {
                        // the most recent copy of exactly this operand on this device, whichever stream made it
                        int dev = 0;
                        (void)hipGetDevice(&dev);
                        Context::XsBuf *hit = nullptr;
                        for (auto &kv : g_ctx.xs_bufs) {
                            Context::XsBuf &b = kv.second;
                            if (kv.first.first == dev && b.ptr && b.src == (const void *)X && b.ld == ldx && b.rows == p.ncols &&
                                b.w == (int64_t)w && b.es == sizeof(T) && b.kind == 0 && (!hit || b.stamp > hit->stamp))
                                hit = &b;
                        }
                        if (hit) {
                            hit->stamp = ++g_ctx.xs_clock;
                            hit->in_use++;
                            xs_use = hit->ptr;
                            pin.hold(xs_use);
                        }
                    }
                    if (!xs_use) {
                        Context::XsBuf *b = nullptr;
                        if (int rc = xs_buffer_locked(st, std::max<size_t>(need, 256), &b)) return rc;
                        xs_use = b->ptr;
                        pin.hold(xs_use);
                        b->src = X;
                        b->ld = ldx;
                        b->rows = p.ncols;
                        b->w = (int64_t)w;
                        b->es = sizeof(T);
                        b->kind = 0;
                        const uint64_t threads = (uint64_t)p.ncols * nslices * (1u << LOG_LPR);
                        if (threads > 0)
                            hipLaunchKernelGGL((k_slice_pack<T, VEC, LOG_LPR>), dim3((unsigned)((threads + 255) / 256)),
                                               dim3(256), 0, st, X, ldx, (uint32_t)p.ncols, w, nslices, (T *)xs_use, (uint32_t)p.ncols);
                    }
                }
                g->packed_src = X;
                g->packed_buf = xs_use;
                g->packed_ld = ldx;
                g->packed_w = (int64_t)w;
                Xg = (const T *)xs_use;
                ldg = F;
                slice_stride = (int64_t)p.ncols * F;
            }
            // 32-bit gather offsets when every gathered byte of a slice sits below 4 GiB of its base
            const bool off32 = ((uint64_t)p.ncols * (uint64_t)ldg + F) * sizeof(T) < (1ull << 32);
            // X far beyond the Infinity Cache (256 MiB): sweep a few slices per launch, every XCD on the same
            // slice(s), so that the gather footprint of a launch is ncols * 128 B * group instead of all of X
            uint32_t sgroup = nslices;
            if (g_tune.slice_group_bytes > 0) {
                const uint64_t per_slice = (uint64_t)p.ncols * F * sizeof(T);
                if (per_slice * nslices > (uint64_t)g_tune.slice_group_bytes)
                    sgroup = (uint32_t)std::max<uint64_t>(1, (uint64_t)g_tune.slice_group_bytes / std::max<uint64_t>(per_slice, 1));
                sgroup = std::min(sgroup, nslices);
            }
            for (uint32_t s0 = 0; s0 < nslices; s0 += sgroup) {
            const uint32_t ns = std::min(sgroup, nslices - s0);
            const T *Xs0 = Xg + (int64_t)s0 * slice_stride;
            T *Cs0 = C + (size_t)s0 * F;
            const uint32_t ws = std::min<uint32_t>(w - s0 * F, ns * F);
            for (uint32_t q = 0; q < p.npanels; q++) {
                const size_t o = p.panel_off[q];
                const uint32_t cnt = (uint32_t)(p.panel_off[q + 1] - o);
                if (cnt == 0) continue;
                const uint32_t ncoop = p.panel_coop[q], nnorm = cnt - ncoop;
                const uint32_t row_blocks = (nnorm + rows_per_block - 1) / rows_per_block;
                const uint32_t coop_blocks = (ncoop + bwaves - 1) / bwaves;  // one wave per long item
                const uint32_t coop_grid = ncoop ? 8u * ns * ((coop_blocks + 7) / 8) : 0u;
                const uint32_t norm_grid = nnorm ? 8u * ns * ((row_blocks + 7) / 8) : 0u;
                const uint32_t *ir = p.d_items + o, *ib = p.d_items + p.n_items + o, *il = p.d_items + 2 * p.n_items + o;
#define PYGIM_LAUNCH_PANEL_D(AM, HV, DQ)                                                                    \
    hipLaunchKernelGGL((k_csr_panel<T, VEC, LOG_LPR, AM, HV, DQ>), dim3(coop_grid + norm_grid), dim3(bthreads), (size_t)g_tune.panel_lds_pad, st,   \
                       ir + ncoop, ib + ncoop, il + ncoop, nnorm, ir, ib, il, ncoop, coop_grid,                           \
                       (AM == 3 ? (const uint32_t *)p.col16 : p.colind), vals, Xs0, ldg,                                  \
                       slice_stride, Cs0, ldc, ws, ns, accumulate ? 1 : 0, q * p.panel_cols,                              \
                       g->deq_out ? g->deq_out + (size_t)s0 * F : nullptr, g->deq_ld, g->deq_amax, g->deq_log2,                  \
                       g->post_mul ? g->post_mul + (size_t)s0 * F : nullptr, g->post_add ? g->post_add + (size_t)s0 * F : nullptr, \
                       g->post_relu)
#define PYGIM_LAUNCH_PANEL(AM, HV) PYGIM_LAUNCH_PANEL_D(AM, HV, false)
                // addressing mode of the gathers (kernels.hpp gather_raw): 128-byte rows of the slice-major copy
                // (with 16-bit panel-local column ids when the plan has them), any stride below 4 GiB, or 64-bit
                int amode = !off32 ? 0 : ((size_t)ldg * sizeof(T) == 128 ? 2 : 1);
                if (amode == 2 && p.col16 && g_tune.panel_col16) amode = 3;
                if constexpr (DeqType<T>::ok) {
                    if (g->deq_out && amode >= 2 && !vals) {
                        if (amode == 3) PYGIM_LAUNCH_PANEL_D(3, 0, true);
                        else PYGIM_LAUNCH_PANEL_D(2, 0, true);
                        continue;
                    }
                }
                if (g->deq_out) return fail(PYGIM_ERR_INVALID, "internal: fused dequantisation on an unsupported sweep");
                if constexpr (sizeof(T) == 8) {
                    if (vals && p.vals_narrow && g_tune.narrow_vals) {  // the 4-byte copy of the values (narrow_values)
                        const T *vals = (const T *)p.vals_narrow;       // (the kernel reads it as NarrowOf<T>)
                        if (amode == 3) PYGIM_LAUNCH_PANEL(3, 2);
                        else if (amode == 2) PYGIM_LAUNCH_PANEL(2, 2);
                        else if (amode == 1) PYGIM_LAUNCH_PANEL(1, 2);
                        else PYGIM_LAUNCH_PANEL(0, 2);
                        continue;
                    }
                }
                if (amode == 3 && vals) PYGIM_LAUNCH_PANEL(3, 1);
                else if (amode == 3) PYGIM_LAUNCH_PANEL(3, 0);
                else if (amode == 2 && vals) PYGIM_LAUNCH_PANEL(2, 1);
                else if (amode == 2) PYGIM_LAUNCH_PANEL(2, 0);
                else if (amode == 1 && vals) PYGIM_LAUNCH_PANEL(1, 1);
                else if (amode == 1) PYGIM_LAUNCH_PANEL(1, 0);
                else if (vals) PYGIM_LAUNCH_PANEL(0, 1);
                else PYGIM_LAUNCH_PANEL(0, 0);
#undef PYGIM_LAUNCH_PANEL
#undef PYGIM_LAUNCH_PANEL_D
            }
            }
            kt.stop();
            return join();
        }
    }
    // CSR kernels (native CSR, or COO through its derived rowptr)
    bool wide = lanes_needed > 32;
    if (g_tune.csr_kernel == 1) wide = true;
    if (g_tune.csr_kernel == 2 && lanes_needed <= 32) wide = false;
    if (nrows > 0) {
        KernelTimer kt(g, st, !p.is_extra);
        if (wide) {
            dim3 grid((nrows + 3) / 4, (lanes_needed + 63) / 64);
            hipLaunchKernelGGL((k_csr_wide<T, VEC>), grid, dim3(256), 0, st, p.rowptr, p.colind, vals, X, ldx, C,
                               ldc, nrows, w, lp.thresh, accumulate ? 1 : 0);
        } else {
            int log_lpr = 0;
            while ((1u << log_lpr) < lanes_needed) log_lpr++;
            const uint32_t rows_per_wave = 64u >> log_lpr;
            const uint32_t waves = (nrows + rows_per_wave - 1) / rows_per_wave;
            hipLaunchKernelGGL((k_csr_sub<T, VEC>), dim3((waves + 3) / 4), dim3(256), 0, st, p.rowptr, p.colind,
                               vals, X, ldx, C, ldc, nrows, w, lp.thresh, accumulate ? 1 : 0, log_lpr);
        }
        kt.stop();
    }
    return join();
}

// LDS-staged product (lds_kernel_gen.hpp / lds_plan.hpp): X is copied slice-major in 256-byte slices (64 features, rows padded to
// whole 256-column chunks), then ONE launch: a 512-thread workgroup per (tile of rows, slice) streams the tile's chunks of X
// through a double-buffered 160 KiB LDS ring (two chunks of 320 columns) and keeps the tile's running sums in registers; C is written once.
// lds_xs: a slice-major copy the caller already made on this stream (the fused quantiser); deq_amax != nullptr: the store
// dequantises, C is then the FLOAT result (row stride ldc elements of 4 bytes)
template <typename T>
int launch_lds(Group *g, Part &p, const T *X, int64_t ldx, T *C, int64_t ldc, uint32_t w, bool accumulate, hipStream_t st,
               const void *lds_xs = nullptr, const uint32_t *deq_amax = nullptr, int deq_log2 = 0) {
    constexpr bool WIDE8 = sizeof(T) == 1;                 // INT8: staged as INT16 (k_slice_pack_widen8), summed by the INT16 stream
    constexpr bool EL8 = sizeof(T) == 8;                   // INT64 / DBL64: slices of 64 features = 512 bytes, a register pair per value
    constexpr uint32_t ROWB = EL8 ? 512 : 256;             // bytes of a staged row
    constexpr uint32_t EPS = WIDE8 ? 128 : ROWB / sizeof(T);   // elements of a slice
    constexpr int PVEC = WIDE8 ? 8 : 16 / (int)sizeof(T);
    // (INT8, and the dequantising INT16 store: the kernel masks features one by one; 8-byte: a lane = a feature)
    const uint32_t w_lanes = (WIDE8 || EL8 || (sizeof(T) == 2 && deq_amax)) ? w : (uint32_t)(((size_t)w * sizeof(T) + 3) / 4);
    const uint32_t nslices = (w + EPS - 1) / EPS;
    const uint64_t rows_pad = lds_rows_pad((uint64_t)p.ncols, p.lds_kc);
    const size_t need = (size_t)rows_pad * nslices * ROWB;
    KernelTimer kt(g, st, !p.is_extra);
    XsPin pin;
    void *xs_use = const_cast<void *>(lds_xs);
    if (!xs_use) {
        std::lock_guard<std::mutex> lk(g_ctx.mu);
        if (g->x_unchanged) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            Context::XsBuf *hit = nullptr;
            for (auto &kv : g_ctx.xs_bufs) {
                Context::XsBuf &b = kv.second;
                // (the padded row count depends on the part's chunk size: a copy made for another ring geometry has another slice stride)
                if (kv.first.first == dev && b.ptr && b.src == (const void *)X && b.ld == ldx && b.rows == p.ncols && b.w == (int64_t)w &&
                    b.es == sizeof(T) && b.kind == (WIDE8 ? 3 : 1) && b.rows_pad == rows_pad && b.bytes >= need &&
                    (!hit || b.stamp > hit->stamp))
                    hit = &b;
            }
            if (hit) {
                hit->stamp = ++g_ctx.xs_clock;
                hit->in_use++;
                xs_use = hit->ptr;
                pin.hold(xs_use);
            }
        }
        if (!xs_use) {
            Context::XsBuf *b = nullptr;
            if (int rc = xs_buffer_locked(st, std::max<size_t>(need, 256), &b)) return rc;
            xs_use = b->ptr;
            pin.hold(xs_use);
            b->src = X;
            b->ld = ldx;
            b->rows = p.ncols;
            b->w = (int64_t)w;
            b->es = sizeof(T);
            b->kind = WIDE8 ? 3 : 1;
            b->rows_pad = rows_pad;
            const uint64_t threads = (uint64_t)p.ncols * nslices * 16;
            if constexpr (WIDE8) {
                if (threads > 0)
                    hipLaunchKernelGGL(k_slice_pack_widen8, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const int8_t *)X, ldx,
                                       (uint32_t)p.ncols, w, nslices, (int16_t *)xs_use, (uint32_t)rows_pad);
            } else if constexpr (EL8) {
                const uint64_t th8 = (uint64_t)p.ncols * nslices * 32;   // 32 pieces of 16 bytes per 512-byte row
                if (th8 > 0)
                    hipLaunchKernelGGL((k_slice_pack<T, 2, 5>), dim3((unsigned)((th8 + 255) / 256)), dim3(256), 0, st, X, ldx,
                                       (uint32_t)p.ncols, w, nslices, (T *)xs_use, (uint32_t)rows_pad);
            } else if (threads > 0) {
                hipLaunchKernelGGL((k_slice_pack<T, PVEC, 4>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, X, ldx,
                                   (uint32_t)p.ncols, w, nslices, (T *)xs_use, (uint32_t)rows_pad);
            }
        }
    }
    LdsArgs a{};
    a.tok = p.lds_tok;
    a.tiles = p.lds_tiles;
    a.rowmap = p.lds_rowmap;
    a.xs = (const char *)xs_use;
    a.c = (char *)C;
    a.slice_stride = rows_pad * ROWB;
    a.ldc_bytes = (uint32_t)((size_t)ldc * (((WIDE8 || sizeof(T) == 2) && deq_amax) ? 4 : sizeof(T)));   // (the dequantising store writes floats)
    a.w = w_lanes;
    a.nslices = nslices;
    a.ntiles = p.lds_ntiles;
    a.accumulate = accumulate ? 1 : 0;
    // column-split plans: the tiles write the partial sums of column range c to row r + c * nrows of a scratch block (compact rows of
    // w_lanes dwords); k_lds_reduce adds the S blocks in range order into C afterwards
    const uint32_t S = p.lds_col_splits;
    const uint64_t ldp = (uint64_t)w_lanes * 4 / sizeof(T);   // elements of a partial row
    if (S > 1) {
        if (deq_amax) return fail(PYGIM_ERR_INVALID, "internal: dequantising store on a column-split LDS plan");
        const size_t need = (size_t)S * (size_t)p.nrows * ldp * sizeof(T);
        if (int rc = ensure(&g->scratch, &g->scratch_bytes, std::max<size_t>(need, 256))) return rc;
        a.c = (char *)g->scratch;
        a.ldc_bytes = (uint32_t)(ldp * sizeof(T));
        a.accumulate = 0;
    }
    a.wdelta = p.lds_wdelta;
    a.deq_amax = deq_amax;
    a.deq_log2 = deq_log2;
    a.post_mul = deq_amax ? g->post_mul : nullptr;   // (the per-column epilogue rides the dequantising store only)
    a.post_add = deq_amax ? g->post_add : nullptr;
    a.post_relu = deq_amax ? g->post_relu : 0;
    // (slice counts that do not divide 8 interleave the slices over the XCDs; a slice-major order measured the same: 2.86 vs 2.80 ms at h = 192)
    a.xcd_group = (nslices == 1 || nslices == 2 || nslices == 4 || nslices == 8) ? 8 / nslices : 0;
    a.xcd_sx = 1;
    // round 5: sx slices per XCD -- the workgroups an XCD runs side by side are then the sx slices of ONE tile and
    // share its code stream in that XCD's L2 (fetched from the fabric nslices / sx times instead of nslices), at the price of sx
    // slices of X streamed through every L2 instead of one
    if (a.xcd_group && p.lds_is_code && nslices > 1) {
        // 0 = automatic (profiles/r05_exp_xcd.txt): two slices per XCD -- fabric traffic 5.28 -> 4.19 GB on the bench workload --
        // and all (up to four) of them for plans whose tiles skip most chunks (little of X to un-share): 5.57 -> 3.72 GB, 1.05 -> 0.98 ms
        uint32_t sx = (uint32_t)std::max<int64_t>(0, g_tune.lds_xcd_slices);
        if (sx == 0) {
            const uint64_t nchunks = ((uint64_t)p.ncols + p.lds_kc - 1) / std::max(1u, p.lds_kc);
            const double fill = p.lds_ntiles && nchunks ? (double)p.lds_slots / ((double)p.lds_ntiles * (double)nchunks) : 1.0;
            sx = fill < 0.3 ? std::min(nslices, 4u) : 2u;
        }
        if (sx > 1 && sx <= nslices && nslices % sx == 0) {
            a.xcd_sx = sx;
            a.xcd_group *= sx;
        }
    }
    const uint32_t grid = a.xcd_group ? 8 * ((p.lds_ntiles + a.xcd_group - 1) / a.xcd_group) * a.xcd_sx : p.lds_ntiles * nslices;
    using KernelFn = void (*)(LdsArgs);
    KernelFn fn = nullptr;
    const bool long16 = p.lds_nw == 16 && p.lds_batch == LDS_L16_BATCH;   // the 16-token-batch geometry (no values)
    if (p.lds_is_code) {   // the schedule compiled into machine code: 16 waves x 96 accumulators, or 8 x 228 (k_lds_code8_*)
        if (!g_tune.lds_code || g_tune.lds_ablate) return fail(PYGIM_ERR_INVALID, "this group's LDS plan is a code stream: lds_code was switched off (or lds_ablate on) after it was created");
        const bool w8 = p.lds_nw == 8;
        if constexpr (EL8) {
            if (!w8 || deq_amax || S > 1 || p.lds_row_bytes != 512) return fail(PYGIM_ERR_INVALID, "internal: 8-byte LDS-staged product on an unsupported plan");
            if constexpr (std::is_same<T, double>::value) fn = k_lds_code8_f64;
            else fn = k_lds_code8_i64;
        } else if constexpr (sizeof(T) == 1) {
            if (!w8 || accumulate || S > 1) return fail(PYGIM_ERR_INVALID, "internal: INT8 LDS-staged product on an unsupported plan");
            fn = deq_amax ? k_lds_code8_i8_deq : k_lds_code8_i8;   // (widened to 16 bits in the staged copy: the INT16 stream)
        } else if constexpr (sizeof(T) == 2) {
            if (deq_amax && (!w8 || S > 1)) return fail(PYGIM_ERR_INVALID, "internal: dequantising LDS-staged product on an unsupported plan");
            fn = deq_amax ? k_lds_code8_i16_deq : (w8 ? k_lds_code8_i16 : k_lds_code_i16);   // (two features to a lane: v_pk_add_u16)
        } else if constexpr (std::is_same<T, float>::value) {
            fn = deq_amax ? (w8 ? k_lds_code8_f32_deq : k_lds_code_f32_deq) : (w8 ? k_lds_code8_f32 : k_lds_code_f32);
        } else {
            fn = deq_amax ? (w8 ? k_lds_code8_i32_deq : k_lds_code_i32_deq) : (w8 ? k_lds_code8_i32 : k_lds_code_i32);
        }
        a.code = p.lds_code;
        a.code_start = p.lds_code_start;
        a.piece_bytes = p.lds_code_piece;
    } else if constexpr (sizeof(T) == 1 || sizeof(T) == 8) {
        return fail(PYGIM_ERR_INVALID, "internal: the INT8 / INT64 / DBL64 LDS-staged product exists in the code-stream form only");
    } else if (deq_amax) {
        if (p.lds_nw != 16 || p.lds_wdelta || sizeof(T) != 4) return fail(PYGIM_ERR_INVALID, "internal: dequantising LDS-staged product on an unsupported plan");
        if constexpr (std::is_same<T, float>::value) fn = long16 ? k_lds_spmm_f32_w16b_deq : k_lds_spmm_f32_w16_deq;
        else if constexpr (std::is_same<T, int32_t>::value) fn = long16 ? k_lds_spmm_i32_w16b_deq : k_lds_spmm_i32_w16_deq;
    } else if constexpr (sizeof(T) == 2) {
        if (p.lds_nw != 16) return fail(PYGIM_ERR_INVALID, "internal: INT16 LDS-staged product needs the 16-wave plan");
        fn = long16 ? k_lds_spmm_i16_w16b : (p.lds_wdelta ? k_lds_spmm_i16_w16_val : k_lds_spmm_i16_w16);
    } else if (long16) {
        if constexpr (std::is_same<T, float>::value) fn = k_lds_spmm_f32_w16b;
        else fn = k_lds_spmm_i32_w16b;
    } else if constexpr (std::is_same<T, float>::value) {
        fn = p.lds_nw == 16 ? (p.lds_wdelta ? k_lds_spmm_f32_w16_val : k_lds_spmm_f32_w16) : k_lds_spmm_f32_w8;
#ifdef PYGIM_LDS_ABLATE
        if (p.lds_nw == 16 && !p.lds_wdelta && !long16) {  // timing experiments (wrong results, scripts/gen_lds_kernel.py; make ablate)
            switch (g_tune.lds_ablate) {
                case 6: fn = k_lds_spmm_f32_w16_ab6; break;
                case 7: fn = k_lds_spmm_f32_w16_ab7; break;
                case 10: fn = k_lds_spmm_f32_w16_ab10; break;
                case 11: fn = k_lds_spmm_f32_w16_ab11; break;
                case 12: fn = k_lds_spmm_f32_w16_ab12; break;
                case 15: fn = k_lds_spmm_f32_w16_ab15; break;
                case 16: fn = k_lds_spmm_f32_w16_ab16; break;
                case 17: fn = k_lds_spmm_f32_w16_ab17; break;
                case 18: fn = k_lds_spmm_f32_w16_ab18; break;
                case 19: fn = k_lds_spmm_f32_w16_ab19; break;
                default: break;
            }
        }
#else
        if (g_tune.lds_ablate) return fail(PYGIM_ERR_INVALID, "lds_ablate needs the ablation build (make -C pygim_amd/csrc ablate)");
#endif
    } else {
        fn = p.lds_nw == 16 ? (p.lds_wdelta ? k_lds_spmm_i32_w16_val : k_lds_spmm_i32_w16) : k_lds_spmm_i32_w8;
    }
    if (deq_amax && g_tune.lds_ablate) return fail(PYGIM_ERR_INVALID, "lds_ablate is a timing experiment of the plain kernel");
    if (!fn) return fail(PYGIM_ERR_INVALID, "internal: no LDS-staged kernel for this plan");
    {
        static std::set<std::pair<int, KernelFn>> attr_done;   // (the attribute is per device)
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(g_ctx.mu);
        if (!attr_done.count({dev, fn})) {
            HIP_TRY(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
            attr_done.insert({dev, fn});
        }
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(p.lds_nw * 64), LDS_BYTES, st, a);
    if (S > 1 && p.nrows > 0 && w > 0) {
        const uint64_t total = (uint64_t)p.nrows * w;
        hipLaunchKernelGGL((k_lds_reduce<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const T *)g->scratch, S, (uint64_t)p.nrows, w, ldp, C,
                           ldc, accumulate ? 1 : 0);
    }
    kt.stop();
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T> bool want_lds(const Group *g, const Part &p, uint32_t w, int64_t ldc, const void *C, bool accumulate = false) {
    if constexpr (sizeof(T) == 8) {  // INT64 / DBL64: the 8-wave code stream on 512-byte rows (unit weights)
        if (!p.lds_is_code || p.lds_nw != 8 || p.lds_row_bytes != 512 || p.lds_col_splits > 1 || p.vals) return false;
    } else if (p.lds_row_bytes == 512) return false;
    if constexpr (sizeof(T) == 1) {  // INT8 rides the 8-wave INT16 code stream (features widened to 16 bits in the staged copy); it neither accumulates
                                     // into C nor splits tiles into column ranges
        if (!p.lds_is_code || p.lds_nw != 8 || accumulate || p.lds_col_splits > 1 || p.vals) return false;
        if ((int64_t)(w / 2) < g_tune.lds_min_width) return false;
    }
    if constexpr (sizeof(T) == 2) {  // two features to a lane: whole lanes, dword-aligned rows of C, the 16-wave plan
        if ((w & 1) || (ldc & 1) || ((uintptr_t)C & 3) || (p.lds_nw != 16 && !p.lds_is_code)) return false;
    }
    if (!p.lds_tiles || g_tune.lds_mode == 2 || (!p.lds_is_code && (p.vals != nullptr) != (p.lds_wdelta != 0)) || g->deq_out || g->pre_xs) return false;
    if (p.lds_is_code && (!g_tune.lds_code || g_tune.lds_ablate)) return false;   // (a code-stream plan serves no token kernel: the sweep instead)
    if (g_tune.lds_mode == 0 && (g_tune.panel_mode == 1 || g_tune.csr_kernel != 0 || g_tune.force_vec_bytes != 0)) return false;  // another kernel was asked for by name
    if (sizeof(T) > 1 && (int64_t)(sizeof(T) == 8 ? (size_t)w : (size_t)w * sizeof(T) / 4) < g_tune.lds_min_width) return false;   // (lanes: 4-byte units of a row; 8-byte: features)
    if ((uint64_t)ldc * sizeof(T) >= (1ull << 32)) return false;
    return true;
}

template <typename T>
int launch_block(Group *g, Part &p, const void *X, int64_t ldx, void *C, int64_t ldc, int64_t w, bool accumulate,
                 hipStream_t st) {
    const T *x = (const T *)X;
    T *c = (T *)C;
    const uint32_t ww = (uint32_t)w;
    if (want_lds<T>(g, p, ww, ldc, C, accumulate)) return launch_lds<T>(g, p, x, ldx, c, ldc, ww, accumulate, st);
    // SpMV end of the path: rows of X of at most 4 elements -> lanes over the ENTRIES of a row (k_csr_vec)
    if (ww <= 4 && g_tune.vec_kernel && g_tune.force_vec_bytes == 0 && g_tune.csr_kernel == 0 && p.rowptr && p.nrows > 0) {
        // LDS-staged form: the plan's column panels with 16-bit local ids, a panel of X (panel_cols x w elements) plus the
        // staging of results inside a workgroup's LDS; rows cut into segments (segment kernels) keep the plain form
        using A = typename AccOf<T>::type;
        constexpr size_t LDS_TOTAL = 160 * 1024 - 64;
        const size_t panel_lds = (((((size_t)p.panel_cols + 1) * ww * sizeof(T)) + 15) & ~(size_t)15) + 16;  // + zero row + alignment lead-in
        const size_t per_entry = 4 + (size_t)ww * sizeof(A);                                                 // a parked (row, sum)
        const bool lds_rule = g_tune.vec_lds && g_tune.panel_mode != 2 && p.d_items && p.col16 && p.lp_panel.n_tasks == 0 &&
                              p.npanels >= 1 && p.panel_long128.size() == 4 * (size_t)p.npanels && (uint64_t)p.nnz < (1ull << 31) &&
                              // every (row, panel) item costs one pipeline step whatever its length: against k_csr_vec it pays from about
                              // 14 entries per item (Reddit-shaped: w = 4, 17 entries per item, 0.46 against 0.73 ms)
                              (double)p.nnz >= (double)g_tune.vec_lds_min_seg * (double)p.nrows * (double)p.npanels;
        if (lds_rule && panel_lds + 64 * 12 * per_entry <= LDS_TOTAL) {
            // ONE launch over all panels (kernels.hpp, k_spmv_lds): a workgroup per (panel, slot) unit, the panels' sums parked
            // in part[panel][row] and added in panel order by k_spmv_reduce
            const size_t stage_budget = std::min<size_t>(32 * 1024, LDS_TOTAL - panel_lds);
            // parked pairs per 16-lane group: a multiple of 24 so that 4-lane groups get F16 / 4 (a multiple of the pipeline's 6), or
            // -- wide sums, little room -- a multiple of 12 and no 4-lane class
            uint32_t F16 = (uint32_t)(stage_budget / (64 * per_entry)) / 24 * 24;
            const int merge4 = F16 == 0 ? 1 : 0;
            if (merge4) F16 = (uint32_t)(stage_budget / (64 * per_entry)) / 12 * 12;
            const size_t shmem = panel_lds + 64 * (size_t)F16 * per_entry;
            const uint32_t cu = (uint32_t)std::max(g_ctx.cu_count, 1);
            if (!p.d_spmv_units) {
                std::vector<uint32_t> act;
                double wsum = 0;
                for (uint32_t q = 0; q < p.npanels; q++)
                    if (p.panel_off[q + 1] > p.panel_off[q]) {
                        act.push_back(q);
                        wsum += (double)std::max<uint64_t>(p.panel_nnz[q], p.panel_off[q + 1] - p.panel_off[q]);
                    }
                if (!act.empty()) {
                    // workgroups per panel in proportion to its entries (at least one); more panels than CUs: one each, taken in turn
                    std::vector<uint32_t> nb(act.size(), 1);
                    if (act.size() < cu) {
                        uint32_t left = cu - (uint32_t)act.size();
                        std::vector<std::pair<double, size_t>> rem;
                        uint32_t given = 0;
                        for (size_t k = 0; k < act.size(); k++) {
                            const uint32_t q = act[k];
                            const double share = (double)left * (double)std::max<uint64_t>(p.panel_nnz[q], p.panel_off[q + 1] - p.panel_off[q]) / wsum;
                            const uint32_t fl = (uint32_t)share;
                            nb[k] += fl;
                            given += fl;
                            rem.emplace_back(share - fl, k);
                        }
                        std::sort(rem.begin(), rem.end(), [](const std::pair<double, size_t> &x, const std::pair<double, size_t> &y) { return x.first > y.first; });
                        for (size_t k = 0; given < left && k < rem.size(); k++, given++) nb[rem[k].second]++;
                    }
                    // slot-major order: consecutive workgroups (dispatched round-robin over the XCDs) take different panels, so with
                    // 8 panels an XCD's L2 serves one panel of X to all its CUs
                    std::vector<SpmvUnit> units;
                    const uint32_t most = *std::max_element(nb.begin(), nb.end());
                    for (uint32_t slot = 0; slot < most; slot++)
                        for (size_t k = 0; k < act.size(); k++) {
                            if (slot >= nb[k]) continue;
                            const uint32_t q = act[k];
                            SpmvUnit u;
                            u.item_off = (uint32_t)p.panel_off[q];
                            u.n_items = (uint32_t)(p.panel_off[q + 1] - p.panel_off[q]);
                            u.n64 = p.panel_long128[4 * q];
                            u.n32 = p.panel_long128[4 * q + 1] - u.n64;
                            u.n16 = p.panel_long128[4 * q + 2] - u.n64 - u.n32;
                            u.n8 = p.panel_long128[4 * q + 3] - u.n64 - u.n32 - u.n16;
                            u.col_base = q * p.panel_cols;
                            u.pcols = (uint32_t)std::min<int64_t>(p.panel_cols, p.ncols - (int64_t)u.col_base);
                            u.slot = slot;
                            u.nslots = nb[k];
                            u.panel = q;
                            units.push_back(u);
                        }
                    HIP_TRY(hipMalloc((void **)&p.d_spmv_units, units.size() * sizeof(SpmvUnit)));
                    HIP_TRY(hipMemcpyAsync(p.d_spmv_units, units.data(), units.size() * sizeof(SpmvUnit), hipMemcpyHostToDevice, st));
                    HIP_TRY(hipStreamSynchronize(st));  // (units is a local; one-time)
                    p.n_spmv_units = (uint32_t)units.size();
                }
            }
            const size_t part_bytes = ((size_t)p.npanels * (size_t)p.nrows * ww * sizeof(A) + 15) & ~(size_t)15;
            if (int rc = ensure(&g->scratch, &g->scratch_bytes, part_bytes)) return rc;
            A *part = (A *)g->scratch;
            KernelTimer kt(g, st, !p.is_extra);
            hipLaunchKernelGGL(k_zero16, dim3((unsigned)((part_bytes / 16 + 255) / 256)), dim3(256), 0, st, (u32x4_t *)part,
                               (uint64_t)(part_bytes / 16));
            if (p.n_spmv_units > 0) {
                const uint32_t *ir = p.d_items, *ib = p.d_items + p.n_items, *il = p.d_items + 2 * p.n_items;
                const uint32_t blocks = std::min(p.n_spmv_units, cu);
#define PYGIM_SPMV_LDS(W, HV)                                                                                                  \
    {                                                                                                                          \
        static std::atomic<uint64_t> attr_devs{0};   /* the attribute is per device: one bit per device ordinal */            \
        int dev_ = 0;                                                                                                          \
        HIP_TRY(hipGetDevice(&dev_));                                                                                          \
        const uint64_t bit_ = 1ull << (dev_ & 63);                                                                             \
        if (!(attr_devs.load(std::memory_order_acquire) & bit_)) {                                                             \
            HIP_TRY(hipFuncSetAttribute((const void *)k_spmv_lds<T, W, HV>, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                        (int)(LDS_TOTAL + 64)));                                                               \
            attr_devs.fetch_or(bit_, std::memory_order_release);                                                               \
        }                                                                                                                      \
        hipLaunchKernelGGL((k_spmv_lds<T, W, HV>), dim3(blocks), dim3(1024), shmem, st, p.d_spmv_units, p.n_spmv_units, ir, ib, \
                           il, p.col16, (const T *)p.vals, x, ldx, part, (uint32_t)p.nrows, (uint32_t)panel_lds, F16, merge4); \
    }
                if (p.vals) {
                    if (ww == 1) PYGIM_SPMV_LDS(1, true) else if (ww == 2) PYGIM_SPMV_LDS(2, true)
                    else if (ww == 3) PYGIM_SPMV_LDS(3, true) else PYGIM_SPMV_LDS(4, true)
                } else {
                    if (ww == 1) PYGIM_SPMV_LDS(1, false) else if (ww == 2) PYGIM_SPMV_LDS(2, false)
                    else if (ww == 3) PYGIM_SPMV_LDS(3, false) else PYGIM_SPMV_LDS(4, false)
                }
#undef PYGIM_SPMV_LDS
            }
            {
                const uint64_t total = (uint64_t)p.nrows * ww;
                const dim3 rgrid((unsigned)((total + 255) / 256));
                if (ww == 1) hipLaunchKernelGGL((k_spmv_reduce<T, 1>), rgrid, dim3(256), 0, st, part, p.npanels, (uint32_t)p.nrows, c, ldc, accumulate ? 1 : 0);
                else if (ww == 2) hipLaunchKernelGGL((k_spmv_reduce<T, 2>), rgrid, dim3(256), 0, st, part, p.npanels, (uint32_t)p.nrows, c, ldc, accumulate ? 1 : 0);
                else if (ww == 3) hipLaunchKernelGGL((k_spmv_reduce<T, 3>), rgrid, dim3(256), 0, st, part, p.npanels, (uint32_t)p.nrows, c, ldc, accumulate ? 1 : 0);
                else hipLaunchKernelGGL((k_spmv_reduce<T, 4>), rgrid, dim3(256), 0, st, part, p.npanels, (uint32_t)p.nrows, c, ldc, accumulate ? 1 : 0);
            }
            kt.stop();
            HIP_TRY(hipGetLastError());
            return 0;
        }
        const double avg = (double)p.nnz / (double)p.nrows;
        const int log_g = avg >= 96 ? 6 : avg >= 48 ? 5 : avg >= 24 ? 4 : avg >= 10 ? 3 : 2;
        const uint64_t waves = ((uint64_t)p.nrows + (64u >> log_g) - 1) / (64u >> log_g);
        KernelTimer kt(g, st, !p.is_extra);
        const dim3 grid((unsigned)((waves + 3) / 4));
#define PYGIM_VEC(W)                                                                                                     \
    if (ww == W) {                                                                                                       \
        if (p.vals)                                                                                                      \
            hipLaunchKernelGGL((k_csr_vec<T, W, true>), grid, dim3(256), 0, st, p.rowptr, p.colind, (const T *)p.vals, x, ldx, c, \
                               ldc, (uint32_t)p.nrows, accumulate ? 1 : 0, log_g);                                      \
        else                                                                                                             \
            hipLaunchKernelGGL((k_csr_vec<T, W, false>), grid, dim3(256), 0, st, p.rowptr, p.colind, (const T *)nullptr, x, ldx, \
                               c, ldc, (uint32_t)p.nrows, accumulate ? 1 : 0, log_g);                                   \
    }
        PYGIM_VEC(1) PYGIM_VEC(2) PYGIM_VEC(3) PYGIM_VEC(4)
#undef PYGIM_VEC
        kt.stop();
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const int vb = pick_vec_bytes(sizeof(T), X, ldx, C, ldc);
    int vec = vb / (int)sizeof(T);
    // The panel sweep moves 16-byte pieces with byte-aligned accesses (kernels.hpp u32x4_b), so it takes rows of
    // ANY alignment (h = 41 floats, 100 int8 ...); only its long-row side kernels want aligned rows, so parts that
    // have such rows keep the alignment-matched kernels.
    if (vb < 16 && g_tune.force_vec_bytes == 0 && p.lp_panel.n_tasks == 0 && want_panel<T>(p, ww, ldx))
        vec = 16 / (int)sizeof(T);
#define CASE(V)                                                               \
    if constexpr (V >= 1 && (size_t)V * sizeof(T) <= 16) {                    \
        if (vec == V) return launch_block_t<T, V>(g, p, x, ldx, c, ldc, ww, accumulate, st); \
    }
    CASE(16) CASE(8) CASE(4) CASE(2)
#undef CASE
    return launch_block_t<T, 1>(g, p, x, ldx, c, ldc, ww, accumulate, st);
}

int launch_block_main(Group *g, Part &p, const void *X, int64_t ldx, void *C, int64_t ldc, int64_t w, bool accumulate,
                      hipStream_t st);

int launch_block_any(Group *g, Part &p, const void *X, int64_t ldx, void *C, int64_t ldc, int64_t w,
                     bool accumulate, hipStream_t st) {
    if (w <= 0 || p.nrows == 0) return 0;  // nothing to write
    g->packed_buf = nullptr;
    if (int rc = launch_block_main(g, p, X, ldx, C, ldc, w, accumulate, st)) return rc;
    // the few entries with a weight other than 1, as (weight - 1), added on top of the pattern product
    if (p.extra) return launch_block_main(g, *p.extra, X, ldx, C, ldc, w, true, st);
    return 0;
}

int launch_block_main(Group *g, Part &p, const void *X, int64_t ldx, void *C, int64_t ldc, int64_t w, bool accumulate,
                      hipStream_t st) {
    switch (g->dtype) {
        case PYGIM_INT8: return launch_block<int8_t>(g, p, X, ldx, C, ldc, w, accumulate, st);
        case PYGIM_INT16: return launch_block<int16_t>(g, p, X, ldx, C, ldc, w, accumulate, st);
        case PYGIM_INT32: return launch_block<int32_t>(g, p, X, ldx, C, ldc, w, accumulate, st);
        case PYGIM_INT64: return launch_block<int64_t>(g, p, X, ldx, C, ldc, w, accumulate, st);
        case PYGIM_FLT32: return launch_block<float>(g, p, X, ldx, C, ldc, w, accumulate, st);
        case PYGIM_DBL64: return launch_block<double>(g, p, X, ldx, C, ldc, w, accumulate, st);
    }
    return fail(PYGIM_ERR_INVALID, "unknown dtype");
}

template <typename T> void launch_check_ones(const void *v, uint32_t n, int *flag, hipStream_t st) {
    hipLaunchKernelGGL((k_check_ones<T>), dim3((n + 255) / 256), dim3(256), 0, st, (const T *)v, n, flag);
}

// Long-row plan from a host copy of rowptr: rows above the threshold are cut
// into segments of `thresh` entries.
void plan_long_rows(const uint32_t *rowptr, int64_t nrows, uint32_t thresh, uint32_t seg, const std::vector<char> *flags,
                    std::vector<uint32_t> &tasks, std::vector<uint32_t> &desc) {
    // a row is long when it has more than `thresh` entries, or -- when flags are given -- when flagged
    for (int64_t r = 0; r < nrows; r++) {
        const uint32_t s = rowptr[r], e = rowptr[r + 1];
        const bool is_long = flags ? (*flags)[(size_t)r] != 0 : (e - s > thresh);
        if (!is_long || e == s) continue;
        const uint32_t first = (uint32_t)(tasks.size() / 3);
        uint32_t n = 0;
        for (uint32_t a = s; a < e; a += seg, n++) {
            tasks.push_back((uint32_t)r);
            tasks.push_back(a);
            tasks.push_back(std::min(e, a + seg));
        }
        desc.push_back((uint32_t)r);
        desc.push_back(first);
        desc.push_back(n);
    }
}

// One-time plans of a part (needs its row pointers on the host): the long-row segment plans and the
// L2-blocked panel plan.  Replaces the reference's prepare_pim_csr/prepare_pim_coo balancing
// (spmm_mul_csr.c:118-259) -- same purpose, different machine.
int build_lds_plan(Part &p, size_t es, int *d_flag_sorted, hipStream_t st, const std::vector<uint32_t> &h_rowptr, int64_t h_hint);
double now_ms();
static thread_local int t_plan_dtype = -1;   // element type of the group being created (the plan builders see the element SIZE only)

// Which rows are alike: once per part (square parts of a million entries and more).  sim_kind 1: a row's columns span a small part of
// the id range -- the ids carry the locality themselves; 2: label propagation found communities (more than one label, none holding half
// of the rows); 0: neither.
static void find_similarity(Part &p, hipStream_t st, bool force_lp) {
    if (p.sim_kind >= 0 && !(force_lp && p.sim_kind != 2)) return;
    p.sim_kind = 0;
    p.sim_order.clear();
    p.sim_why.clear();
    if (p.nrows != p.ncols || p.nrows == 0 || (!force_lp && p.nnz < (1 << 20))) { p.sim_why = "not a square part of a million entries and more"; return; }
    if (p.cols_sorted == 0) { p.sim_why = "rows not stored in column order"; return; }
    const double t0 = now_ms();
    const double span = force_lp ? 1.0 : lds_mean_row_span(p.rowptr, p.colind, (uint32_t)p.nrows, (uint32_t)p.ncols, st);
    if (span < 0.5) {
        p.sim_kind = 1;
        p.sim_why = "the stored ids are local already (mean row span " + std::to_string(span) + " of the id range)";
    } else {
        uint64_t agree = 0;
        std::string why = lds_similarity_order(p.rowptr, p.colind, (uint32_t)p.nrows, (int)std::min<int64_t>(std::max<int64_t>(g_tune.lds_lp_rounds, 0), 64), st, p.sim_order,
                                               &p.sim_labels, &p.sim_largest, &agree);
        p.sim_agree = p.nnz ? (double)agree / (double)p.nnz : 0.0;
        // communities: more than one label, none holding half of the rows, and the labels MEAN something -- at least 30 % of the stored
        // entries join two rows of one label (a planted partition: its inside fraction; labels of a graph without structure: ~0, and
        // ordering by them only costs the sweep its balance: products-shaped uniform 19.8 -> 22.6 ms)
        if (why.empty() && !force_lp && (p.sim_labels < 2 || (uint64_t)p.sim_largest * 2 > (uint64_t)p.nrows || p.sim_agree < 0.3))
            why = "the propagation found no communities (" + std::to_string(p.sim_labels) + " labels, " + std::to_string(p.sim_agree) + " of the entries inside one)";
        if (!why.empty()) {
            p.sim_order.clear();
            p.sim_why = why;
            (void)hipGetLastError();
        } else {
            p.sim_kind = 2;
        }
    }
    if (getenv("PYGIM_PLAN_TIMING"))
        fprintf(stderr, "[pygim plan] similarity: kind %d, %u labels, the largest holds %u of %lld rows%s%s  (%.1f ms)\n", p.sim_kind, p.sim_labels, p.sim_largest,
                (long long)p.nrows, p.sim_why.empty() ? "" : " -- ", p.sim_why.c_str(), now_ms() - t0);
}

int build_plans(Part &p, size_t es, int *d_flag_sorted, hipStream_t st, int64_t h_hint = 0, bool allow_lds = true) {
    std::vector<uint32_t> h_rowptr((size_t)p.nrows + 1);
    if (hipMemcpy(h_rowptr.data(), p.rowptr, h_rowptr.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(PYGIM_ERR_HIP, "rowptr D2H");
    auto build_long = [&](LongPlan &lp, uint32_t thresh, const std::vector<char> *flags) -> bool {
        lp.thresh = thresh;
        std::vector<uint32_t> tasks, desc;
        const uint32_t seg = (uint32_t)std::min<int64_t>(thresh, std::max<int64_t>(64, g_tune.long_segment));
        plan_long_rows(h_rowptr.data(), p.nrows, thresh, seg, flags, tasks, desc);
        lp.n_tasks = (uint32_t)(tasks.size() / 3);
        lp.n_long = (uint32_t)(desc.size() / 3);
        if (!lp.n_tasks) return true;
        return hipMalloc((void **)&lp.d_tasks, tasks.size() * 4) == hipSuccess &&
               hipMalloc((void **)&lp.d_desc, desc.size() * 4) == hipSuccess &&
               hipMemcpy(lp.d_tasks, tasks.data(), tasks.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
               hipMemcpy(lp.d_desc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    };
    const uint32_t base_thresh = (uint32_t)std::min<int64_t>(0x7FFFFFFF, std::max<int64_t>(64, g_tune.long_row_threshold));
    if (!build_long(p.lp_base, base_thresh, nullptr)) return fail(PYGIM_ERR_HIP, "long-row plan upload");
    // L2-blocked plan: columns cut into panels whose 128-byte feature slice fits the L2 budget,
    // per panel the length-sorted list of work items
    if (g_tune.panel_mode != 2 && p.nrows > 0 && p.nnz > 0) {
        int64_t budget_rows = std::max<int64_t>(1, g_tune.panel_bytes / 128);
        // groups whose rows of X hold at most 4 elements never take the wide sweep: their panels are sized for the LDS-staged
        // SpMV kernel instead (a panel of X, h elements per column, inside 128 KiB of a workgroup's LDS; the rest parks results)
        if (h_hint >= 1 && h_hint <= 4 && g_tune.vec_lds && g_tune.vec_kernel)
            budget_rows = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(budget_rows, 65536),  // (16-bit panel-local ids)
                                                                 (int64_t)((128 * 1024 - 64) / ((size_t)h_hint * es))));
        uint32_t npan = (uint32_t)std::max<int64_t>(1, (p.ncols + budget_rows - 1) / budget_rows);
        bool worth = g_tune.panel_mode == 1 || npan == 1 ||
                     (double)p.nnz / ((double)p.nrows * npan) >= (double)g_tune.panel_min_seg;
        if (npan > 1 && worth) {
            // column panels are cut by binary search inside each row: the rows must hold sorted column ids
            // (torch_sparse / coalesce() deliver them sorted); otherwise fall back to one panel
            int unsorted = 0;
            if (hipMemsetAsync(d_flag_sorted, 0, sizeof(int), st) != hipSuccess) return fail(PYGIM_ERR_HIP, "flag reset");
            hipLaunchKernelGGL(k_check_sorted_cols, dim3((unsigned)((p.nrows + 255) / 256)), dim3(256), 0, st, p.rowptr,
                               p.colind, (uint32_t)p.nrows, d_flag_sorted);
            if (hipMemcpy(&unsorted, d_flag_sorted, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
                return fail(PYGIM_ERR_HIP, "sortedness check");
            p.cols_sorted = unsorted ? 0 : 1;
            if (unsorted) worth = false;
        }
        if (!worth) {
            // too few entries per (row, panel) for L2 blocking: ONE panel.  The sweep's other half --
            // length-sorted items, 32-id column chunks, line-sized gathers from the slice-major copy, wave-
            // cooperative long rows -- still beats whole-row gathers (products-shaped, X = 2.5 GB:
            // 23.8 ms against 28.6 ms for k_csr_wide)
            npan = 1;
            worth = true;
        }
        if (worth) {
            p.npanels = npan;
            p.panel_cols = (uint32_t)((p.ncols + npan - 1) / npan);
            // panel pointers of every row (device binary searches), then the lists on the host
            const size_t pp_elems = (size_t)(npan + 1) * (size_t)p.nrows;
            std::vector<uint32_t> pp(pp_elems);
            if (npan == 1) {
                std::copy(h_rowptr.begin(), h_rowptr.end() - 1, pp.begin());
                std::copy(h_rowptr.begin() + 1, h_rowptr.end(), pp.begin() + p.nrows);
            } else {
                uint32_t *d_pp = nullptr;
                if (hipMalloc((void **)&d_pp, pp_elems * 4) != hipSuccess) return fail(PYGIM_ERR_HIP, "panel pointers");
                hipLaunchKernelGGL(k_build_panel_ptr, dim3((unsigned)((pp_elems + 255) / 256)), dim3(256), 0, st,
                                   (const uint32_t *)nullptr, p.rowptr, p.colind, (uint32_t)p.nrows, npan, p.panel_cols, d_pp);
                const hipError_t ce = hipMemcpy(pp.data(), d_pp, pp_elems * 4, hipMemcpyDeviceToHost);
                (void)hipFree(d_pp);
                if (ce != hipSuccess) return fail(PYGIM_ERR_HIP, "panel pointers D2H");
            }
            const uint32_t *rp = h_rowptr.data();
            const size_t nr = (size_t)p.nrows;
            // rows whose share of ONE panel is enormous (16x the long-row threshold) leave the sweep for
            // the segment kernels; merely long items stay and are walked by a whole wave (panel_coop)
            std::vector<char> heavy(nr, 0);
            for (uint32_t q = 0; q < npan; q++) {
                const uint32_t *lo = pp.data() + (size_t)q * nr, *hi = lo + nr;
                for (size_t r = 0; r < nr; r++)
                    if ((uint64_t)(hi[r] - lo[r]) > (uint64_t)base_thresh * 16) heavy[r] = 1;
            }
            if (!build_long(p.lp_panel, base_thresh, &heavy)) return fail(PYGIM_ERR_HIP, "long-row plan upload");
            std::vector<uint32_t> rows_v, beg_v, len_v, order;
            // rank of every row in the locality order (empty = items by length alone): the SpMV end keeps its length classes
            std::vector<uint32_t> loc_rank;
            if (g_tune.panel_locality && !(h_hint >= 1 && h_hint <= 4) && !p.is_extra) {
                find_similarity(p, st, g_tune.panel_locality == 2);   // (2: propagation whatever the size -- tests)
                if (p.sim_kind == 1) {
                    loc_rank.resize(nr);
                    for (size_t r = 0; r < nr; r++) loc_rank[r] = (uint32_t)r;
                } else if (p.sim_kind == 2) {
                    loc_rank.resize(nr);
                    for (size_t k = 0; k < nr; k++) loc_rank[p.sim_order[k]] = (uint32_t)k;
                }
                p.panel_locality_used = !loc_rank.empty();
            }
            p.panel_off.assign(1, 0);
            p.panel_coop.clear();
            p.panel_long128.clear();
            p.panel_nnz.clear();
            const uint32_t coop_cap = (uint32_t)std::max<int64_t>(64, g_tune.panel_coop);
            for (uint32_t q = 0; q < npan; q++) {
                const uint32_t *lo = pp.data() + (size_t)q * nr, *hi = lo + nr;
                order.clear();
                for (size_t r = 0; r < nr; r++) {
                    const uint32_t deg = rp[r + 1] - rp[r];
                    if (heavy[r]) continue;                          // segment kernels
                    // (empty rows are items of panel 0: they zero their row of C; a correction part only ever adds, so it has none)
                    if (hi[r] > lo[r] || (q == 0 && deg == 0 && !p.is_extra)) order.push_back((uint32_t)r);
                }
                {   // stable counting sort by item length, longest first (O(items + longest))
                    uint32_t longest = 0;
                    for (uint32_t r : order) longest = std::max(longest, hi[r] - lo[r]);
                    std::vector<size_t> start((size_t)longest + 2, 0);
                    for (uint32_t r : order) start[(size_t)(longest - (hi[r] - lo[r])) + 1]++;
                    for (size_t k = 1; k < start.size(); k++) start[k] += start[k - 1];
                    std::vector<uint32_t> sorted(order.size());
                    for (uint32_t r : order) sorted[start[(size_t)(longest - (hi[r] - lo[r]))]++] = r;
                    order.swap(sorted);
                }
                uint32_t nco = 0;
                for (uint32_t r : order) nco += (hi[r] - lo[r] > coop_cap) ? 1u : 0u;  // sorted: a prefix
                if (!loc_rank.empty() && order.size() > nco) {
                    // LOCALITY order (round 5) behind the wave-cooperative prefix: blocks of 2048 rows of the similarity (or id) order,
                    // longest first inside a block (a stable counting sort by block of the length-sorted list).  Workgroups that run side
                    // by side on an XCD then gather the same community's rows of X out of its L2 instead of 8 x 32 unrelated ones
                    const uint32_t nblk = (uint32_t)((nr + 2047) / 2048);
                    std::vector<size_t> bstart((size_t)nblk + 1, 0);
                    for (size_t k = nco; k < order.size(); k++) bstart[(size_t)(loc_rank[order[k]] >> 11) + 1]++;
                    for (size_t k = 1; k < bstart.size(); k++) bstart[k] += bstart[k - 1];
                    std::vector<uint32_t> byblk(order.size() - nco);
                    for (size_t k = nco; k < order.size(); k++) byblk[bstart[loc_rank[order[k]] >> 11]++] = order[k];
                    std::copy(byblk.begin(), byblk.end(), order.begin() + nco);
                }
                p.panel_coop.push_back(nco);
                uint32_t n256 = 0, n128 = 0, n64 = 0, n32 = 0;
                uint64_t pn = 0;
                for (uint32_t r : order) {
                    const uint32_t l = hi[r] - lo[r];
                    n256 += l > 256u ? 1u : 0u;  // sorted: prefixes
                    n128 += l > 128u ? 1u : 0u;
                    n64 += l > 64u ? 1u : 0u;
                    n32 += l > 32u ? 1u : 0u;
                    pn += l;
                }
                p.panel_long128.push_back(n256);
                p.panel_long128.push_back(n128);
                p.panel_long128.push_back(n64);
                p.panel_long128.push_back(n32);
                p.panel_nnz.push_back(pn);
                for (uint32_t r : order) {
                    rows_v.push_back(r);
                    beg_v.push_back(lo[r]);
                    const uint32_t first = (lo[r] == rp[r]) ? 0x80000000u : 0u;     // no entries in earlier panels
                    const uint32_t last = (hi[r] == rp[r + 1]) ? 0x40000000u : 0u;  // none in later panels
                    len_v.push_back((hi[r] - lo[r]) | first | last);
                }
                p.panel_off.push_back(rows_v.size());
            }
            p.n_items = rows_v.size();
            if (p.n_items > 0) {
                if (hipMalloc((void **)&p.d_items, 3 * p.n_items * 4) != hipSuccess ||
                    hipMemcpy(p.d_items, rows_v.data(), p.n_items * 4, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(p.d_items + p.n_items, beg_v.data(), p.n_items * 4, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(p.d_items + 2 * p.n_items, len_v.data(), p.n_items * 4, hipMemcpyHostToDevice) != hipSuccess)
                    return fail(PYGIM_ERR_HIP, "panel plan upload");
                // 16-bit panel-local column ids (2 bytes per entry more, half the index bytes per sweep)
                if (g_tune.panel_col16 && p.panel_cols <= 65536 && p.nnz > 0) {
                    // (+ 64 bytes: the LDS-staged SpMV kernel fetches ids four at a time and may read past the last entry)
                    if (hipMalloc((void **)&p.col16, (size_t)p.nnz * 2 + 64) != hipSuccess) return fail(PYGIM_ERR_HIP, "col16 alloc");
                    hipLaunchKernelGGL(k_make_col16, dim3((unsigned)((p.nnz + 255) / 256)), dim3(256), 0, st, p.colind,
                                       (uint64_t)p.nnz, p.panel_cols, p.col16);
                    if (hipStreamSynchronize(st) != hipSuccess) return fail(PYGIM_ERR_HIP, "col16 build");
                }
            }
        }
    }
    // (the column blocks of a group that also has its merged matrix run through the merged plan: no second token stream for them)
    return allow_lds ? build_lds_plan(p, es, d_flag_sorted, st, h_rowptr, h_hint) : 0;
}

// One-time: the schedule of the LDS-staged product (lds_plan.hpp) for parts it pays for.  Built on the host from
// the row pointers and column ids (the reference balances its DPU row ranges on the host too, spmm_mul_csr.c:118-259).
// The ladder: the code-stream form; if its generation or its executable memory fails, the SAME product as a token
// plan (k_lds_spmm_*: the schedule as data, ~20 % slower) -- not the L2 sweep (2-3 x slower); if the schedule itself cannot be built
// (host memory), the sweep.  Whatever happened is kept as text with the part (pygim_group_lds_note).
static int build_lds_plan_form(Part &p, size_t es, int *d_flag_sorted, hipStream_t st, const std::vector<uint32_t> &h_rowptr, int64_t h_hint,
                               bool allow_code);
int build_lds_plan(Part &p, size_t es, int *d_flag_sorted, hipStream_t st, const std::vector<uint32_t> &h_rowptr, int64_t h_hint) {
    p.lds_note.clear();
    int rc = build_lds_plan_form(p, es, d_flag_sorted, st, h_rowptr, h_hint, true);
    if (rc == -1) {
        const std::string why = p.lds_note;
        rc = build_lds_plan_form(p, es, d_flag_sorted, st, h_rowptr, h_hint, false);
        p.lds_note = why + (p.lds_tiles ? "; products take the token form of the LDS-staged kernel (k_lds_spmm_*)" : "; no token plan either: the L2 sweep serves this group");
    }
    if (p.lds_note.empty())
        p.lds_note = p.lds_tiles ? (p.lds_is_code ? (p.lds_codegen_device || p.lds_codegen_why.empty() ? std::string("code-stream form")
                                                                                                         : "code-stream form (written by the host encoder: " + p.lds_codegen_why + ")")
                                                  : std::string("token form of the LDS-staged kernel")) : "no LDS-staged plan (rule, type or width): the L2 sweep serves this group";
    return rc;
}

// lds_codegen = 2: the device-generated code stream against the host encoder's, word for word (and the stream offsets, the row map, the
// chunk counts of the tiles and the statistics).  "" = identical.
static std::string codegen_verify(const Part &p, const LdsGeometry &geo, uint32_t code_op, const std::vector<uint32_t> &h_rowptr, const uint32_t *rorder,
                                  const CgDeviceResult &dr) {
    std::vector<uint32_t> h_col((size_t)p.nnz), h_val;
    if (hipMemcpy(h_col.data(), p.colind, h_col.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return "column ids D2H failed";
    if (p.vals) {
        h_val.resize((size_t)p.nnz);
        if (hipMemcpy(h_val.data(), p.vals, h_val.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return "values D2H failed";
    }
    LdsPlanHost plan;
    LdsCodeHost ch;
    try {
        lds_plan_build(h_rowptr.data(), h_col.data(), (uint32_t)p.nrows, (uint32_t)p.ncols, geo, plan, (unsigned)std::max<int64_t>(0, g_tune.lds_threads),
                       p.vals ? h_val.data() : nullptr, rorder);
        lds_code_from_plan(plan, code_op, ch, (unsigned)std::max<int64_t>(0, g_tune.lds_threads), (uint32_t)std::max<int64_t>(0, g_tune.lds_code_gsize),
                           (uint32_t)std::max<int64_t>(0, g_tune.lds_code_nsets), 0);
    } catch (const std::exception &e) {
        return std::string("the host encoder failed: ") + e.what();
    }
    if (ch.code.size() * 4 != dr.code_bytes) return "code size " + std::to_string(dr.code_bytes) + " against " + std::to_string(ch.code.size() * 4) + " bytes";
    if (plan.ntiles != dr.ntiles || plan.slots != dr.slots) return "tile / slot counts";
    std::vector<uint32_t> d_code(ch.code.size());
    void *stage = nullptr;   // (the executable allocation is read through a plain device buffer)
    if (hipMalloc(&stage, std::max<size_t>(dr.code_bytes, 256)) != hipSuccess) return "out of device memory for the comparison";
    const uint64_t n16 = dr.code_bytes / 16;
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const u32x4_t *)dr.code, (u32x4_t *)stage, n16);
    const bool got = hipMemcpy(d_code.data(), stage, dr.code_bytes, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(stage);
    if (!got) return "code D2H failed";
    if (memcmp(d_code.data(), ch.code.data(), dr.code_bytes) != 0) {
        size_t at = 0;
        while (d_code[at] == ch.code[at]) at++;
        size_t s = 0;
        while (s + 1 < ch.start.size() && ch.start[s + 1] / 4 <= at) s++;
        char buf[160];
        snprintf(buf, sizeof buf, "first difference at dword %zu (stream %zu + %zu): %08x against the host's %08x", at, s, at - (size_t)(ch.start[s] / 4), d_code[at], ch.code[at]);
        return buf;
    }
    std::vector<uint64_t> d_start(ch.start.size());
    std::vector<uint32_t> d_rowmap(plan.rowmap.size());
    std::vector<LdsTile> d_tiles(plan.tiles.size());
    if (hipMemcpy(d_start.data(), dr.d_start, d_start.size() * 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(d_rowmap.data(), dr.d_rowmap, d_rowmap.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(d_tiles.data(), dr.d_tiles, d_tiles.size() * sizeof(LdsTile), hipMemcpyDeviceToHost) != hipSuccess)
        return "table D2H failed";
    if (d_start != ch.start) return "stream offsets";
    if (d_rowmap != plan.rowmap) return "row map";
    for (size_t t = 0; t < d_tiles.size(); t++)
        if (d_tiles[t].nch != plan.tiles[t].nch || d_tiles[t].row0 != plan.tiles[t].row0 || d_tiles[t].nnz != plan.tiles[t].nnz || d_tiles[t].chunk0 != plan.tiles[t].chunk0)
            return "tile table";
    if (dr.entries != ch.entries || dr.pairs != ch.pairs || dr.shared != ch.shared) return "statistics";
    return std::string();
}

static int build_lds_plan_form(Part &p, size_t es, int *d_flag_sorted, hipStream_t st, const std::vector<uint32_t> &h_rowptr, int64_t h_hint,
                               bool allow_code) {   // (es by value: INT8 plans are made as INT16 plans)
    // PYGIM_PLAN_TIMING=1: where the creation time of an LDS-staged plan goes (stderr; the reference prints its own prepare / load times)
    static const bool plan_timing = getenv("PYGIM_PLAN_TIMING") != nullptr;
    double t_mark = now_ms();
    auto lap = [&](const char *what) {
        if (!plan_timing) return;
        const double t = now_ms();
        fprintf(stderr, "[pygim plan] %-34s %8.1f ms\n", what, t - t_mark);
        t_mark = t;
    };
    const bool int8_code = es == 1 && t_plan_dtype == PYGIM_INT8 && !p.vals && allow_code && g_tune.lds_code && g_tune.lds_waves == 16 && g_tune.lds_code_waves != 16;
    if (es == 1) {   // INT8: the 8-wave INT16 code stream on features widened to 16 bits (no token form, no 16-wave form)
        if (!int8_code) return 0;
        es = 2;
    }
    // INT64 / DBL64, unit weights: the 8-wave code stream on rows of 512 bytes (a register pair per value; no token form)
    const bool el8_code = es == 8 && (t_plan_dtype == PYGIM_INT64 || t_plan_dtype == PYGIM_DBL64) && !p.vals && allow_code && g_tune.lds_code &&
                          g_tune.lds_waves == 16 && g_tune.lds_code_waves != 16;
    if (es == 8 && !el8_code) return 0;
    if (g_tune.lds_mode == 2 || (es != 4 && es != 2 && es != 8) || p.is_extra || p.nnz == 0 || p.nrows == 0 || p.ncols == 0) return 0;
    if ((p.vals || es == 2) && g_tune.lds_waves != 16) return 0;  // the valued and the INT16 kernels exist for the 16-wave geometry
    if (h_hint > 0 && (es == 8 ? h_hint : (h_hint * (int64_t)es) / 4) < g_tune.lds_min_width) return 0;   // no product of this group is wide enough (want_lds): no plan, no code
    if ((uint64_t)p.ncols * (es == 8 ? 512ull : 256ull) >= (1ull << 32) - (1ull << 20) || (uint64_t)p.nnz >= (1ull << 31)) return 0;
    LdsGeometry geo;
    geo.NW = g_tune.lds_waves == 16 ? 16 : 8;
    geo.KA = lds_ka(geo.NW);
    geo.KC = LDS_KC;
    geo.BATCH = lds_batch(geo.NW);
    // FLT32 / INT32 with unit weights: the code-stream form (the schedule compiled into machine code); its plan is built in the
    // geometry of its kernels and serves no token kernel
    // (valued matrices: FLT32 only -- the value is the literal of a v_mul_f32 in the stream; the integer multiplies have no literal form)
    const bool want_code = allow_code && g_tune.lds_code && geo.NW == 16 &&
                           ((es == 4 && t_plan_dtype == PYGIM_FLT32) || (!p.vals && es == 4 && t_plan_dtype == PYGIM_INT32) ||
                            (!p.vals && es == 2 && t_plan_dtype == PYGIM_INT16) || int8_code || el8_code);
    const int64_t slice_b = es == 8 ? 512 : 256;   // bytes of a row of a slice
    const int64_t nsl_hint = h_hint > 0 ? (h_hint * (int64_t)es + slice_b - 1) / slice_b : 4;
    if (want_code) {
        // 8 waves x 228 accumulators (2 waves per SIMD): tiles of 1 824 rows -- Reddit h = 256 is two rounds of workgroups on 256 CUs
        // instead of three, a third less of X staged -- or round 3's 16 waves x 96
        const int64_t cw = g_tune.lds_code_waves;
        const bool eight = cw == 8 || (cw != 16 && LDS_CODE_AUTO_WAVES == 8) || int8_code || el8_code;
        if (eight) {
            geo.NW = 8;
            geo.KA = el8_code ? LDS_CODE8_KA64 : LDS_CODE8_KA;
            geo.BATCH = 8;
        }
        if (el8_code) geo.row_bytes = 512;
    }
    // tiles sized so that one product of the group's width runs as whole rounds of workgroups (lds_plan.hpp)
    if (g_tune.lds_round_tiles && h_hint > 0)
        geo.rows_per_tile = lds_rows_per_tile((uint32_t)p.nrows, geo.NW * geo.KA, (uint32_t)nsl_hint, (uint32_t)std::max(g_ctx.cu_count, 1));
    // a row share too short to fill the chip with workgroups that each stream a whole slice of X (a rank's share on N GPUs: 24 tiles x
    // 4 slices on 256 CUs): full-height row tiles, each split into S column ranges -- S x as many workgroups, each landing 1/S of X;
    // partial sums per range, reduced in range order (launch_lds).  Integers stay exact; FLT32 only when asked (lds_col_split_f32)
    if (g_tune.lds_col_split != 1 && (geo.NW == 16 || want_code) && !int8_code && !el8_code && (t_plan_dtype != PYGIM_FLT32 || g_tune.lds_col_split_f32) && !p.vals) {
        const uint32_t cus = (uint32_t)std::max(g_ctx.cu_count, 1);
        const uint64_t nsl = h_hint > 0 ? (uint64_t)nsl_hint : 1;
        const uint64_t tall = ((uint64_t)p.nrows + geo.NW * geo.KA - 1) / (geo.NW * geo.KA), wgs = tall * nsl;
        uint32_t S = 1;
        if (g_tune.lds_col_split > 1) S = (uint32_t)std::min<int64_t>(g_tune.lds_col_split, 16);
        else if (wgs * 2 <= cus) S = (uint32_t)std::min<uint64_t>(8, cus / wgs);
        if (S > 1 && (uint64_t)p.nrows * S < (1ull << 31)) {
            geo.col_splits = S;
            geo.rows_per_tile = 0;   // full-height tiles: the column ranges fill the chip
        }
    }
    if (want_code) {
        // the ring.  Two buffers of 320 columns: the workgroup meets at every slot boundary (round 3).  Three or more (round 4): one
        // barrier in the middle of a slot, NBUF - 2 chunks in flight beside the one being read, no drain at the boundary -- the DMA
        // requests of a CU never dry up (scripts/micro/fillrate.hip: 81 GB/s per CU with 2 x 80 KiB, 113 GB/s with 3 x 48 or 4 x 40 KiB)
        int64_t nbuf = g_tune.lds_code_nbuf;
        if (nbuf == 0) nbuf = geo.NW == 8 ? LDS_CODE8_AUTO_NBUF : 2;
        nbuf = std::min<int64_t>(std::max<int64_t>(nbuf, 2), 10);
        static const uint32_t kc_of[11] = {0, 0, 320, 192, 160, 128, 96, 64, 64, 64, 64};
        uint32_t kc = g_tune.lds_code_kc > 0 ? (uint32_t)g_tune.lds_code_kc : kc_of[nbuf] * 256 / geo.row_bytes;
        const uint32_t kq = 1024 * geo.NW / geo.row_bytes;                  // columns of one 1 KiB piece per wave
        kc = std::max(kq, kc / kq * kq);                                    // whole pieces per wave
        while ((uint64_t)kc * geo.row_bytes * (uint64_t)nbuf > LDS_BYTES || kc * (uint32_t)nbuf > 640) kc -= kq;   // (the LDS; 10-bit LDS rows in a token)
        geo.KC = kc;
        geo.NBUF = (uint32_t)nbuf;
        // measured (profiles/r04_lds_kernel.md): with five buffers the boundary form -- four chunks in flight -- is 1-2 % ahead of the mid-slot form on
        // every shape tried (2.02-2.04 against 2.06 ms on the bench workload, DBL64 7.03 against 7.12): the default
        geo.boundary = g_tune.lds_code_boundary == 2 ? 0u : 1u;
    }
    if (g_tune.lds_mode == 0 &&
        lds_plan_uniform_reuse((uint64_t)p.nnz, (uint32_t)p.nrows, (uint32_t)p.ncols, geo) * 100.0 < (double)g_tune.lds_min_reuse_x100)
        return 0;
    if (p.cols_sorted < 0) {  // (the panel plan may have asked already)
        int unsorted = 0;
        if (hipMemsetAsync(d_flag_sorted, 0, sizeof(int), st) != hipSuccess) return fail(PYGIM_ERR_HIP, "flag reset");
        hipLaunchKernelGGL(k_check_sorted_cols, dim3((unsigned)((p.nrows + 255) / 256)), dim3(256), 0, st, p.rowptr, p.colind,
                           (uint32_t)p.nrows, d_flag_sorted);
        if (hipMemcpy(&unsorted, d_flag_sorted, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(PYGIM_ERR_HIP, "sortedness check");
        p.cols_sorted = unsorted ? 0 : 1;
    }
    if (!p.cols_sorted) return 0;  // stored order inside a row must be column order for the chunk walk
    const uint32_t code_op = t_plan_dtype == PYGIM_FLT32 ? 0x02000000u : t_plan_dtype == PYGIM_INT32 ? 0x68000000u : t_plan_dtype == PYGIM_DBL64 ? LDS_CODE_ADD_F64 :
                             t_plan_dtype == PYGIM_INT64 ? LDS_CODE_ADD_U64 : LDS_CODE_PK_ADD_U16;
    // Round 5: similarity tiles (lds_reorder_dev.hpp) -- rows with similar neighbourhoods share a tile: shared LDS reads, skipped chunks.
    // The result does not depend on the order (each row is summed by one wave in stored order whichever tile holds it).
    std::vector<uint32_t> rorder;
    if (want_code && p.nrows == p.ncols &&
        (g_tune.lds_tile_order == 1 || (g_tune.lds_tile_order == 2 && p.nnz >= (1 << 20)))) {
        // automatic: ids that are local already (a row's columns span a small part of the id range: tiles of consecutive rows skip most
        // chunks as they are) keep consecutive tiles; otherwise the propagation decides -- it must have found communities (more than
        // one label, none holding half of the rows), else its order is the consecutive one anyway (profiles/r05_structured.txt)
        find_similarity(p, st, g_tune.lds_tile_order == 1);
        if (p.sim_kind == 2) {
            rorder = p.sim_order;
            p.lds_tile_labels = p.sim_labels;
            p.lds_tile_largest = p.sim_largest;
        }
        lap("similarity order (label propagation)");
    }
    const uint32_t *ro = rorder.empty() ? nullptr : rorder.data();
    // Round 5: the code stream GENERATED ON THE DEVICE (lds_codegen_dev.hpp) from the resident CSR -- the same bytes the host encoder
    // below would write (lds_codegen = 2 checks that, word for word) without the graph ever visiting the host.  Plans it does not
    // cover (column-split tiles, the mid-slot hand-off, timing experiments), or a failure on the way, take the host encoder.
    std::string dev_why;
    if (want_code && g_tune.lds_codegen && !(g_tune.lds_fail & 7) && !g_tune.lds_code_exp) {
        CgDeviceResult dr;
        std::string exec_why;
        auto alloc_exec = [&](size_t bytes) -> void * { return exec_alloc(bytes, &exec_why); };
        auto free_exec = [&](void *q) { (void)hsa_amd_memory_pool_free(q); };
        dev_why = (es != 4 && p.vals) ? std::string("valued entries of this width") :
                  cg_run_on_device(p.rowptr, p.colind, (const uint32_t *)p.vals, h_rowptr.data(), ro, (uint32_t)p.nrows, (uint32_t)p.ncols, geo, code_op,
                                   (uint32_t)std::max<int64_t>(0, g_tune.lds_code_gsize), (uint32_t)std::max<int64_t>(0, g_tune.lds_code_nsets), st, alloc_exec,
                                   free_exec, dr, [] { return now_ms(); });
        if (dev_why.empty()) {
            if (plan_timing)
                fprintf(stderr, "[pygim plan] device code generation: rows %.1f, chunk lists %.1f, keys + sort %.1f, columns %.1f, slots + groups %.1f, sizes %.1f, emission %.1f, "
                                "whole %.1f ms (%.1f MB of code)\n", dr.ms[0], dr.ms[1], dr.ms[2], dr.ms[3], dr.ms[4], dr.ms[5], dr.ms[6], dr.ms[7], dr.code_bytes / 1e6);
            if (g_tune.lds_codegen == 2) {   // the checker: the host encoder's blob against what the device wrote
                std::string diff = codegen_verify(p, geo, code_op, h_rowptr, ro, dr);
                if (!diff.empty()) {
                    (void)hsa_amd_memory_pool_free(dr.code);
                    (void)hipFree(dr.d_start);
                    (void)hipFree(dr.d_rowmap);
                    (void)hipFree(dr.d_tiles);
                    return fail(PYGIM_ERR_INVALID, ("lds_codegen = 2: the device-generated code stream differs from the host encoder's: " + diff).c_str());
                }
            }
            p.lds_code = (char *)dr.code;
            p.lds_code_bytes = dr.code_bytes;
            p.lds_code_start = dr.d_start;
            p.lds_rowmap = dr.d_rowmap;
            p.lds_tiles = dr.d_tiles;
            p.lds_code_pairs = dr.pairs;
            p.lds_code_shared = dr.shared;
            p.lds_is_code = true;
            p.lds_code_piece = geo.KC * geo.row_bytes / geo.NW;
            p.lds_code_gsize = dr.regs.gsize;
            p.lds_code_nsets = dr.regs.nsets;
            p.lds_col_splits = geo.col_splits;
            p.lds_kc = geo.KC;
            p.lds_nbuf = geo.NBUF;
            p.lds_row_bytes = geo.row_bytes;
            p.lds_ka = geo.KA;
            p.lds_ntiles = dr.ntiles;
            p.lds_nw = geo.NW;
            p.lds_batch = geo.BATCH;
            p.lds_slots = dr.slots;
            p.lds_tokens = dr.entries;   // (a code stream has no padding)
            p.lds_codegen_device = true;
            lap("code stream generated on the device");
            return 0;
        }
        (void)hipGetLastError();
        if (plan_timing) fprintf(stderr, "[pygim plan] device code generation not used: %s\n", dev_why.c_str());
        p.lds_codegen_why = dev_why;
    } else if (want_code) {
        p.lds_codegen_why = !g_tune.lds_codegen ? "lds_codegen = 0" : (g_tune.lds_fail & 7) ? "lds_fail set" : "timing experiment";
    }
    std::vector<uint32_t> h_col((size_t)p.nnz);
    if (hipMemcpy(h_col.data(), p.colind, h_col.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(PYGIM_ERR_HIP, "colind D2H");
    lap("sortedness check + column ids D2H");
    std::vector<uint32_t> h_val;
    if (p.vals) {
        h_val.resize((size_t)p.nnz);
        if (es == 4) {
            if (hipMemcpy(h_val.data(), p.vals, h_val.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(PYGIM_ERR_HIP, "values D2H");
        } else {  // INT16: the value in both halves of its dword (packed multiply)
            std::vector<uint16_t> v16((size_t)p.nnz);
            if (hipMemcpy(v16.data(), p.vals, v16.size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return fail(PYGIM_ERR_HIP, "values D2H");
            for (size_t i = 0; i < v16.size(); i++) h_val[i] = (uint32_t)v16[i] * 0x10001u;
        }
    }
    LdsPlanHost plan;
    try {
        if (g_tune.lds_fail & 4) throw std::runtime_error("lds_fail: forced failure of the schedule build");
        lds_plan_build(h_rowptr.data(), h_col.data(), (uint32_t)p.nrows, (uint32_t)p.ncols, geo, plan, (unsigned)std::max<int64_t>(0, g_tune.lds_threads),
                       p.vals ? h_val.data() : nullptr, ro);
    } catch (const std::exception &e) {   // host memory, threads: no LDS-staged plan, the sweep serves the group
        p.lds_note = std::string("the schedule of the LDS-staged product could not be built (") + e.what() + "): the L2 sweep serves this group";
        return 0;
    }
    lap("schedule (tiles, tokens)");
    // long slots (a community-structured graph: a tile streams few chunks, a wave gets hundreds of tokens per chunk): the per-batch
    // bookkeeping is what is left to save -- the 16-token-batch geometry, when the tiles fit its 80 accumulators per wave
    if (geo.NW == 16 && !p.vals && !want_code && g_tune.lds_long_slots && plan.slots > 0 &&
        (double)plan.ntokens / ((double)plan.slots * geo.NW) >= (double)g_tune.lds_long_slots) {
        LdsGeometry gl = geo;
        gl.KA = LDS_L16_KA;
        gl.BATCH = LDS_L16_BATCH;
        gl.rows_per_tile = geo.col_splits > 1 ? 0 : g_tune.lds_round_tiles && h_hint > 0
                               ? lds_rows_per_tile((uint32_t)p.nrows, gl.NW * gl.KA, (uint32_t)nsl_hint, (uint32_t)std::max(g_ctx.cu_count, 1))
                               : 0;
        const uint32_t r_now = geo.rows_per_tile ? geo.rows_per_tile : geo.NW * geo.KA, r_new = gl.rows_per_tile ? gl.rows_per_tile : gl.NW * gl.KA;
        if (r_new >= r_now) {  // no more tiles than before
            try {
                lds_plan_build(h_rowptr.data(), h_col.data(), (uint32_t)p.nrows, (uint32_t)p.ncols, gl, plan,
                               (unsigned)std::max<int64_t>(0, g_tune.lds_threads), nullptr);
            } catch (const std::exception &e) {
                p.lds_note = std::string("the schedule of the LDS-staged product could not be built (") + e.what() + "): the L2 sweep serves this group";
                return 0;
            }
            geo = gl;
        }
    }
    std::vector<uint32_t>().swap(h_col);
    std::vector<uint32_t>().swap(h_val);
    if (p.vals && !want_code) {
        // one buffer: the token stream, then the value stream at the same positions (the kernel adds a 32-bit byte offset)
        if ((uint64_t)plan.tok.size() * 8 >= (1ull << 32)) return 0;
        p.lds_wdelta = (uint32_t)(plan.tok.size() * 4);
        plan.tok.insert(plan.tok.end(), plan.wts.begin(), plan.wts.end());
        std::vector<uint32_t>().swap(plan.wts);
    }
    auto up = [&](auto **dst, const auto &v) {
        using E = typename std::remove_reference<decltype(v[0])>::type;
        const size_t bytes = std::max<size_t>(v.size() * sizeof(E), 64);
        if (hipMalloc((void **)dst, bytes) != hipSuccess) return false;
        return v.empty() || hipMemcpy(*dst, v.data(), v.size() * sizeof(E), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (plan.header_overflow) {  // a slot header field would not fit its 14 bits (a wave with > 16 383 batches in one chunk, > 5 M columns)
        p.lds_note = "the LDS-staged schedule does not fit its slot headers (more than 16 383 batches of one wave in one chunk, or more than 5 M columns): the L2 sweep serves this group";
        return 0;
    }
    if (want_code) {
        // the schedule as machine code (1.5 instructions per stored entry instead of 4 + bookkeeping) in EXECUTABLE memory; the token
        // stream itself stays on the host (the kernel needs the tile table and the row map only)
        LdsCodeHost ch;
        try {
            if (g_tune.lds_fail & 1) throw std::runtime_error("lds_fail: forced failure of the code generation");
            lds_code_from_plan(plan, t_plan_dtype == PYGIM_FLT32 ? 0x02000000u : t_plan_dtype == PYGIM_INT32 ? 0x68000000u : t_plan_dtype == PYGIM_DBL64 ? LDS_CODE_ADD_F64 :
                                     t_plan_dtype == PYGIM_INT64 ? LDS_CODE_ADD_U64 : LDS_CODE_PK_ADD_U16, ch,
                               (unsigned)std::max<int64_t>(0, g_tune.lds_threads), (uint32_t)std::max<int64_t>(0, g_tune.lds_code_gsize),
                               (uint32_t)std::max<int64_t>(0, g_tune.lds_code_nsets), (uint32_t)std::max<int64_t>(0, g_tune.lds_code_exp));
        } catch (const std::exception &e) {   // out of host memory or threads: the same schedule as a token plan (build_lds_plan)
            p.lds_note = std::string("code-stream form not available: generating the instruction streams failed (") + e.what() + ")";
            return -1;
        }
        lap("code generation");
        std::vector<uint32_t>().swap(plan.tok);
        std::string why_exec;
        void *code = (g_tune.lds_fail & 2) ? nullptr : exec_alloc_upload(ch.code.data(), ch.code.size() * 4, &why_exec);
        if (!code) {                     // no executable pool on this runtime, or out of device memory: the token plan
            p.lds_note = "code-stream form not available: " + (why_exec.empty() ? std::string("lds_fail: forced failure of the executable allocation") : why_exec) +
                         " (" + std::to_string(ch.code.size() * 4) + " bytes of code)";
            return -1;
        }
        if (!up(&p.lds_code_start, ch.start) || !up(&p.lds_rowmap, plan.rowmap) || !up(&p.lds_tiles, plan.tiles)) {
            (void)hsa_amd_memory_pool_free(code);
            return fail(PYGIM_ERR_HIP, "code-stream plan upload");
        }
        lap("code + tables upload");
        p.lds_code = (char *)code;
        p.lds_code_bytes = ch.code.size() * 4;
        p.lds_code_pairs = ch.pairs;
        p.lds_is_code = true;
        p.lds_code_piece = geo.KC * geo.row_bytes / geo.NW;
        p.lds_code_gsize = ch.regs.gsize;
        p.lds_code_nsets = ch.regs.nsets;
        p.lds_code_shared = ch.shared;
    } else if (!up(&p.lds_tok, plan.tok) || !up(&p.lds_rowmap, plan.rowmap) || !up(&p.lds_tiles, plan.tiles)) {
        return fail(PYGIM_ERR_HIP, "LDS plan upload");
    }
    p.lds_col_splits = geo.col_splits;
    p.lds_kc = geo.KC;
    p.lds_nbuf = geo.NBUF;
    p.lds_row_bytes = geo.row_bytes;
    p.lds_ka = geo.KA;
    p.lds_ntiles = plan.ntiles;
    p.lds_nw = geo.NW;
    p.lds_batch = geo.BATCH;
    p.lds_slots = plan.slots;
    p.lds_tokens = plan.ntokens;
    return 0;
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int need_init() {
    if (!g_ctx.inited) return fail(PYGIM_ERR_NO_DEVICE, "backend not initialised: call dpu_init_ranks / pygim_init_ranks first");
    return 0;
}

}  // namespace

template <typename T>
static void launch_pack(const void *const *d_ptrs, uint32_t gcount, uint64_t n, void *out, hipStream_t st) {
    const uint64_t total = n * gcount;
    if (total == 0) return;
    hipLaunchKernelGGL((k_pack_vectors<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const T *const *)d_ptrs, gcount, n, (T *)out);
}


// shared body of the three run_group flavours.
//   windows[k]   : dense operand k (flattened over parts when per_part)
//   ld[k]        : its row stride in elements
//   per_part     : grande layout (each window holds only the rows of its own sparse part)
// Integer weights that are 1 except for a few entries (at most nnz / 64): keep the unit-weight pattern in `p` and move
// the exceptions, as (weight - 1), into p.extra (CSR by rows, entries in stored order).  One-time, at group creation.
template <typename T>
int split_unit_pattern_t(Part &p, size_t es, int *d_flag_sorted, uint32_t *d_counter, hipStream_t st) {
    const uint32_t n = (uint32_t)p.nnz;
    if (hipMemsetAsync(d_counter, 0, sizeof(uint32_t), st) != hipSuccess) return fail(PYGIM_ERR_HIP, "counter reset");
    hipLaunchKernelGGL((k_count_non_ones<T>), dim3((n + 255) / 256), dim3(256), 0, st, (const T *)p.vals, n, d_counter);
    uint32_t cnt = 0;
    if (hipMemcpy(&cnt, d_counter, sizeof(cnt), hipMemcpyDeviceToHost) != hipSuccess) return fail(PYGIM_ERR_HIP, "non-one count");
    if (cnt == 0 || (uint64_t)cnt * 64 > (uint64_t)n) return 0;  // all ones is handled by the caller; many weights: keep them
    uint32_t *d_e = nullptr, *d_c = nullptr;
    T *d_v = nullptr;
    auto drop = [&]() {
        if (d_e) (void)hipFree(d_e);
        if (d_c) (void)hipFree(d_c);
        if (d_v) (void)hipFree(d_v);
    };
    if (hipMalloc((void **)&d_e, (size_t)cnt * 4) != hipSuccess || hipMalloc((void **)&d_c, (size_t)cnt * 4) != hipSuccess ||
        hipMalloc((void **)&d_v, (size_t)cnt * sizeof(T)) != hipSuccess) {
        drop();
        return fail(PYGIM_ERR_HIP, "extra part scratch");
    }
    (void)hipMemsetAsync(d_counter, 0, sizeof(uint32_t), st);
    hipLaunchKernelGGL((k_extract_non_ones<T>), dim3((n + 255) / 256), dim3(256), 0, st, (const T *)p.vals, p.colind, n, cnt,
                       d_counter, d_e, d_c, d_v);
    std::vector<uint32_t> e(cnt), c(cnt), rp((size_t)p.nrows + 1);
    std::vector<T> v(cnt);
    const bool ok = hipMemcpy(e.data(), d_e, (size_t)cnt * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(c.data(), d_c, (size_t)cnt * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(v.data(), d_v, (size_t)cnt * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess &&
                    hipMemcpy(rp.data(), p.rowptr, rp.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
    drop();
    if (!ok) return fail(PYGIM_ERR_HIP, "extra part D2H");
    // stored order (the append order on the device is arbitrary)
    std::vector<uint32_t> order(cnt);
    for (uint32_t k = 0; k < cnt; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return e[a] < e[b]; });
    std::vector<uint32_t> xrp((size_t)p.nrows + 1, 0), xc(cnt);
    std::vector<T> xv(cnt);
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t src = order[k];
        const uint32_t row = (uint32_t)(std::upper_bound(rp.begin(), rp.end(), e[src]) - rp.begin()) - 1;
        xrp[(size_t)row + 1]++;
        xc[k] = c[src];
        xv[k] = v[src];
    }
    for (int64_t r = 0; r < p.nrows; r++) xrp[(size_t)r + 1] += xrp[(size_t)r];
    std::unique_ptr<Part> x(new Part);
    x->is_extra = true;
    x->nrows = p.nrows;
    x->ncols = p.ncols;
    x->nnz = cnt;
    x->dense_cols = p.dense_cols;
    x->own_rowptr = x->own_colind = x->own_vals = true;
    if (hipMalloc((void **)&x->rowptr, xrp.size() * 4) != hipSuccess || hipMalloc((void **)&x->colind, (size_t)cnt * 4) != hipSuccess ||
        hipMalloc(&x->vals, (size_t)cnt * sizeof(T)) != hipSuccess ||
        hipMemcpy(x->rowptr, xrp.data(), xrp.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(x->colind, xc.data(), (size_t)cnt * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(x->vals, xv.data(), (size_t)cnt * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        free_part(*x);
        return fail(PYGIM_ERR_HIP, "extra part upload");
    }
    if (int rc = build_plans(*x, es, d_flag_sorted, st)) {
        free_part(*x);
        return rc;
    }
    p.extra = std::move(x);
    if (p.own_vals) (void)hipFree(p.vals);
    p.vals = nullptr;
    p.own_vals = false;
    return 0;
}

// One-time: a 4-byte copy of the values of an 8-byte element type for the sweep (kernels.hpp SweepVal), when every value
// survives the round trip bit for bit.  The weights of a normalised adjacency are float32 numbers widened by the caller's
// DBL64 choice, integer weights are small counts: both halve the value bytes every 128-byte slice re-reads.
template <typename T> int narrow_values_t(Part &p, int *d_flag, hipStream_t st) {
    using NT = typename NarrowOf<T>::type;
    void *buf = nullptr;
    if (hipMalloc(&buf, (size_t)p.nnz * sizeof(NT)) != hipSuccess) {
        (void)hipGetLastError();
        return 0;  // (an optimisation only)
    }
    int bad = 0;
    if (hipMemsetAsync(d_flag, 0, sizeof(int), st) != hipSuccess) { (void)hipFree(buf); return fail(PYGIM_ERR_HIP, "flag reset"); }
    hipLaunchKernelGGL((k_narrow_vals<T>), dim3((unsigned)(((uint64_t)p.nnz + 255) / 256)), dim3(256), 0, st, (const T *)p.vals, (uint64_t)p.nnz,
                       (NT *)buf, d_flag);
    if (hipMemcpyAsync(&bad, d_flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        (void)hipFree(buf);
        return fail(PYGIM_ERR_HIP, "narrow values");
    }
    if (bad) (void)hipFree(buf);
    else p.vals_narrow = buf;
    return 0;
}
int narrow_values(Part &p, int dtype, int *d_flag, hipStream_t st) {
    if (!g_tune.narrow_vals || !p.vals || p.nnz == 0 || p.is_extra || !p.d_items) return 0;  // (d_items: the part has a sweep plan)
    if (dtype == PYGIM_INT64) return narrow_values_t<int64_t>(p, d_flag, st);
    if (dtype == PYGIM_DBL64) return narrow_values_t<double>(p, d_flag, st);
    return 0;
}

int split_unit_pattern(Part &p, int dtype, size_t es, int *d_flag_sorted, uint32_t *d_counter, hipStream_t st) {
    if (!p.vals || p.nnz == 0 || !g_tune.split_unit_pattern) return 0;
    switch (dtype) {
        case PYGIM_INT8: return split_unit_pattern_t<int8_t>(p, es, d_flag_sorted, d_counter, st);
        case PYGIM_INT16: return split_unit_pattern_t<int16_t>(p, es, d_flag_sorted, d_counter, st);
        case PYGIM_INT32: return split_unit_pattern_t<int32_t>(p, es, d_flag_sorted, d_counter, st);
        case PYGIM_INT64: return split_unit_pattern_t<int64_t>(p, es, d_flag_sorted, d_counter, st);
        default: return 0;  // floats keep their weights in the loop (stored-order sums)
    }
}

// One-time: the group's sparse parts as one matrix (see Group::merged).
template <typename T>
int build_merged_t(Group *g, size_t es, hipStream_t st) {
    const int64_t nrows = g->total_rows;
    uint64_t nnz = 0, ncols = 0;
    bool any_vals = false;
    for (auto &p : g->parts) {
        nnz += (uint64_t)p.nnz;
        ncols += (uint64_t)p.ncols;
        any_vals = any_vals || p.vals != nullptr;
    }
    any_vals = any_vals && !g->all_ones;
    if (nnz == 0 || nnz > 0xFFFFFFFFull || ncols > 0xFFFFFFFFull) return 0;  // nothing to gain / would not fit 32-bit ids
    std::unique_ptr<Part> m(new Part);
    m->nrows = nrows;
    m->ncols = (int64_t)ncols;
    m->nnz = (int64_t)nnz;
    m->dense_cols = g->parts[0].dense_cols;
    m->own_rowptr = m->own_colind = true;
    uint32_t *cursor = nullptr;
    auto fail_free = [&](const char *what) {
        if (cursor) (void)hipFree(cursor);
        free_part(*m);
        return fail(PYGIM_ERR_HIP, what);
    };
    const size_t rp_bytes = (size_t)(nrows + 1) * 4;
    if (hipMalloc((void **)&m->rowptr, rp_bytes) != hipSuccess || hipMalloc((void **)&m->colind, (size_t)nnz * 4) != hipSuccess ||
        hipMalloc((void **)&cursor, rp_bytes) != hipSuccess)
        return fail_free("merged matrix alloc");
    if (any_vals) {
        if (hipMalloc(&m->vals, (size_t)nnz * es) != hipSuccess) return fail_free("merged values alloc");
        m->own_vals = true;
    }
    if (hipMemsetAsync(m->rowptr, 0, rp_bytes, st) != hipSuccess) return fail_free("merged rowptr reset");
    const unsigned rgrid = (unsigned)((nrows + 1 + 255) / 256);
    for (auto &p : g->parts) hipLaunchKernelGGL(k_merge_count, dim3(rgrid), dim3(256), 0, st, p.rowptr, (uint32_t)nrows, m->rowptr);
    if (hipMemcpyAsync(cursor, m->rowptr, rp_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail_free("cursor copy");
    uint32_t coff = 0;
    for (auto &p : g->parts) {
        if (p.nnz > 0)
            hipLaunchKernelGGL((k_merge_scatter<T>), dim3((unsigned)((p.nnz + 255) / 256)), dim3(256), 0, st, p.rowptr, p.rowind,
                               p.colind, (const T *)p.vals, (uint32_t)nrows, (uint32_t)p.nnz, coff, cursor, m->colind, (T *)m->vals);
        hipLaunchKernelGGL(k_merge_advance, dim3(rgrid), dim3(256), 0, st, p.rowptr, (uint32_t)nrows, cursor);
        coff += (uint32_t)p.ncols;
    }
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return fail_free("merged matrix build");
    (void)hipFree(cursor);
    cursor = nullptr;
    int rc = 0;
    if (!g->all_ones) rc = split_unit_pattern(*m, g->dtype, es, g->d_flags + 4, (uint32_t *)(g->d_flags + 5), st);
    if (!rc) rc = build_plans(*m, es, g->d_flags + 4, st, g->h);
    if (!rc) rc = narrow_values(*m, g->dtype, g->d_flags + 6, st);
    if (rc) {
        free_part(*m);
        return rc;
    }
    g->merged = std::move(m);
    return 0;
}

int build_merged(Group *g, size_t es, hipStream_t st) {
    switch (g->dtype) {
        case PYGIM_INT8: return build_merged_t<int8_t>(g, es, st);
        case PYGIM_INT16: return build_merged_t<int16_t>(g, es, st);
        case PYGIM_INT32: return build_merged_t<int32_t>(g, es, st);
        case PYGIM_INT64: return build_merged_t<int64_t>(g, es, st);
        case PYGIM_FLT32: return build_merged_t<float>(g, es, st);
        case PYGIM_DBL64: return build_merged_t<double>(g, es, st);
    }
    return 0;
}

static int run_group_common(Group *g, const void *const *windows, const int64_t *ld, bool per_part, void *out,
                            hipStream_t st) {
    const size_t es = dtype_size(g->dtype);
    size_t nwin = 0;
    if (per_part)
        for (auto &p : g->parts) nwin += p.dense_cols.size();
    else
        nwin = g->parts[0].dense_cols.size();
    if (!windows || !out) return fail(PYGIM_ERR_INVALID, "null dense operand or output");
    for (size_t k = 0; k < nwin; k++)
        if (!windows[k]) return fail(PYGIM_ERR_INVALID, "null dense part");
    const bool dev_out = is_device_ptr(out);
    bool dev_in = is_device_ptr(windows[0]);
    for (size_t k = 1; k < nwin; k++)
        if (is_device_ptr(windows[k]) != dev_in) return fail(PYGIM_ERR_INVALID, "dense parts mix host and device memory");
    if (dev_in != dev_out) return fail(PYGIM_ERR_INVALID, "dense parts and output must both be host or both be device memory");

    // window geometry
    std::vector<int64_t> rows(nwin), lds(nwin), widths(nwin);
    {
        size_t k = 0;
        if (per_part) {
            for (auto &p : g->parts)
                for (size_t j = 0; j < p.dense_cols.size(); j++, k++) {
                    rows[k] = p.ncols;
                    widths[k] = p.dense_cols[j];
                    lds[k] = ld ? ld[k] : p.dense_cols[j];
                    if (lds[k] < widths[k]) return fail(PYGIM_ERR_INVALID, "window stride smaller than its width");
                }
        } else {
            for (size_t j = 0; j < nwin; j++) {
                rows[j] = g->total_cols;
                widths[j] = g->parts[0].dense_cols[j];
                lds[j] = ld ? ld[j] : widths[j];
            }
        }
    }
    std::vector<const void *> dwin(nwin);
    void *dout = out;
    double t_in = 0, t_k = 0, t_out = 0;
    if (!dev_in) {
        const double t0 = now_ms();
        size_t total = 0;
        std::vector<size_t> off(nwin);
        for (size_t k = 0; k < nwin; k++) {
            off[k] = total;
            total += (((size_t)rows[k] * lds[k] * es) + 255) & ~(size_t)255;
        }
        if (int rc = ensure(&g->stage_in, &g->stage_in_bytes, std::max<size_t>(total, 256))) return rc;
        if (int rc = ensure(&g->stage_out, &g->stage_out_bytes, std::max<size_t>((size_t)g->total_rows * g->h * es, 256))) return rc;
        for (size_t k = 0; k < nwin; k++) {
            dwin[k] = (char *)g->stage_in + off[k];
            const size_t bytes = (size_t)rows[k] * lds[k] * es;
            if (bytes) HIP_TRY(hipMemcpyAsync((void *)dwin[k], windows[k], bytes, hipMemcpyHostToDevice, st));
        }
        dout = g->stage_out;
        HIP_TRY(hipStreamSynchronize(st));
        t_in = now_ms() - t0;
    } else {
        for (size_t k = 0; k < nwin; k++) dwin[k] = windows[k];
    }
    const double t1 = now_ms();
    // block products: sum over sparse parts, concatenate over dense parts
    if (g->merged && g_tune.merge_parts && g->parts.size() > 1) {
        // ... as ONE product with the merged matrix (Group::merged)
        Part &m = *g->merged;
        const size_t nd0 = g->parts[0].dense_cols.size();
        bool in_place = !per_part;  // one window, or windows that sit side by side in one row-major matrix
        for (size_t j = 1; in_place && j < nd0; j++)
            in_place = lds[j] == lds[0] && (const char *)dwin[j] == (const char *)dwin[j - 1] + (size_t)widths[j - 1] * es;
        const char *x = nullptr;
        int64_t ldx = 0;
        if (in_place) {
            x = (const char *)dwin[0];
            ldx = lds[0];
        } else {
            // lay the windows side by side (and, for per-part windows, the parts' row ranges one below the other)
            const int64_t ldc_el = (int64_t)((((size_t)g->h * es + 15) & ~(size_t)15) / es);
            const size_t need = (size_t)g->total_cols * ldc_el * es;
            if (int rc = ensure(&g->xcat, &g->xcat_bytes, std::max<size_t>(need, 256))) return rc;
            int64_t brow = 0;
            size_t kbase = 0;
            for (size_t i = 0; i < g->parts.size(); i++) {
                Part &p = g->parts[i];
                int64_t a = 0;
                for (size_t j = 0; j < p.dense_cols.size(); j++) {
                    const size_t k = per_part ? kbase + j : j;
                    const char *src = (const char *)dwin[k] + (per_part ? 0 : (size_t)brow * lds[k] * es);
                    if (widths[k] > 0 && p.ncols > 0)
                        HIP_TRY(hipMemcpy2DAsync((char *)g->xcat + ((size_t)brow * ldc_el + (size_t)a) * es, (size_t)ldc_el * es, src,
                                                 (size_t)lds[k] * es, (size_t)widths[k] * es, (size_t)p.ncols,
                                                 hipMemcpyDeviceToDevice, st));
                    a += widths[k];
                }
                brow += p.ncols;
                kbase += p.dense_cols.size();
            }
            x = (const char *)g->xcat;
            ldx = ldc_el;
        }
        if (int rc = launch_block_any(g, m, x, ldx, dout, g->h, g->h, /*accumulate=*/false, st)) return rc;
    } else {
    int64_t brow = 0;
    size_t kbase = 0;
    for (size_t i = 0; i < g->parts.size(); i++) {
        Part &p = g->parts[i];
        int64_t acol = 0;
        const size_t nd = p.dense_cols.size();
        // Several dense windows (ds_parts chunks, grande's per-unit feature windows) are one product of the
        // full width here: a window is an artefact of the reference's DPU layout, and narrow windows
        // (32 int8 = 32 bytes) would waste the 128-byte gather.  Windows that already sit side by side in
        // one row-major matrix are used in place; others are laid side by side first (one 2-D copy each).
        if (nd > 1 && g_tune.fuse_windows) {
            const size_t k0 = per_part ? kbase : 0;
            int64_t wsum = 0;
            bool adjacent = true;
            for (size_t j = 0; j < nd; j++) {
                if (j > 0 && (lds[k0 + j] != lds[k0] ||
                              (const char *)dwin[k0 + j] != (const char *)dwin[k0 + j - 1] + (size_t)widths[k0 + j - 1] * es))
                    adjacent = false;
                wsum += widths[k0 + j];
            }
            const char *x = nullptr;
            int64_t ldx = 0;
            if (adjacent) {
                x = (const char *)dwin[k0] + (per_part ? 0 : (size_t)brow * lds[k0] * es);
                ldx = lds[k0];
            } else {
                // row stride rounded up to 16 bytes so that the sweep's 16-byte pieces stay available
                const int64_t ldc_el = (int64_t)((((size_t)wsum * es + 15) & ~(size_t)15) / es);
                const size_t need = (size_t)p.ncols * ldc_el * es;
                if (int rc = ensure(&g->xcat, &g->xcat_bytes, std::max<size_t>(need, 256))) return rc;
                int64_t a = 0;
                for (size_t j = 0; j < nd; j++) {
                    const size_t k = k0 + j;
                    const char *src = (const char *)dwin[k] + (per_part ? 0 : (size_t)brow * lds[k] * es);
                    if (widths[k] > 0 && p.ncols > 0)
                        HIP_TRY(hipMemcpy2DAsync((char *)g->xcat + (size_t)a * es, (size_t)ldc_el * es, src, (size_t)lds[k] * es,
                                                 (size_t)widths[k] * es, (size_t)p.ncols, hipMemcpyDeviceToDevice, st));
                    a += widths[k];
                }
                x = (const char *)g->xcat;
                ldx = ldc_el;
            }
            if (int rc = launch_block_any(g, p, x, ldx, dout, g->h, wsum, /*accumulate=*/i > 0, st)) return rc;
        } else
        for (size_t j = 0; j < nd; j++) {
            const size_t k = per_part ? kbase + j : j;
            const int64_t w = widths[k];
            const char *x = (const char *)dwin[k] + (per_part ? 0 : (size_t)brow * lds[k] * es);
            char *c = (char *)dout + (size_t)acol * es;
            if (int rc = launch_block_any(g, p, x, lds[k], c, g->h, w, /*accumulate=*/i > 0, st)) return rc;
            acol += w;
        }
        brow += p.ncols;
        kbase += p.dense_cols.size();
    }
    }
    if (!dev_in) {
        HIP_TRY(hipStreamSynchronize(st));
        t_k = now_ms() - t1;
        const double t2 = now_ms();
        const size_t bytes = (size_t)g->total_rows * g->h * es;
        if (bytes) HIP_TRY(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        t_out = now_ms() - t2;
        g->timers[0] = t_in;
        g->timers[1] = t_k;
        g->timers[2] = t_out;
        g->timers[3] = 0;
    }
    return 0;
}



// ---- the conv layers' quantiser (models/quantize.py:20-42) in three launches -------------------------
static bool flat4(const void *a, int64_t ld, uint64_t rows, uint32_t w, const void *b = nullptr) {
    return ld == (int64_t)w && (rows * w) % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0;
}
static int launch_absmax(const float *X, int64_t ldx, uint64_t rows, uint32_t w, uint32_t *amax, hipStream_t st) {
    const uint64_t total = rows * w;
    if (!total) return 0;
    const bool flat = flat4(X, ldx, rows, w);
    const uint64_t work = flat ? total / 4 : total;
    const unsigned grid = (unsigned)std::min<uint64_t>((work + 255) / 256, 2048);  // 8 blocks per CU, grid-stride loop
    if (flat) hipLaunchKernelGGL(k_absmax_bits<true>, dim3(grid), dim3(256), 0, st, X, ldx, rows, w, amax);
    else hipLaunchKernelGGL(k_absmax_bits<false>, dim3(grid), dim3(256), 0, st, X, ldx, rows, w, amax);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <typename T>
static int launch_quantize(const float *X, int64_t ldx, uint64_t rows, uint32_t w, const uint32_t *amax, int log2_range, T *xq,
                           float *scale_out, hipStream_t st) {
    const uint64_t total = rows * w;
    if (!total) return 0;
    if (flat4(X, ldx, rows, w, xq))
        hipLaunchKernelGGL((k_quantize<T, true>), dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, X, ldx, rows, w, amax,
                           log2_range, xq, scale_out);
    else
        hipLaunchKernelGGL((k_quantize<T, false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, ldx, rows, w, amax,
                           log2_range, xq, scale_out);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <typename T>
static int launch_dequantize(const T *q, uint64_t n, const uint32_t *amax, int log2_range, float *out, hipStream_t st) {
    if (!n) return 0;
    if (n % 4 == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)out % 16) == 0)
        hipLaunchKernelGGL((k_dequantize<T, true>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, q, n, amax, log2_range, out);
    else
        hipLaunchKernelGGL((k_dequantize<T, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q, n, amax, log2_range, out);
    HIP_TRY(hipGetLastError());
    return 0;
}
static int quant_log2_range(int dtype) {  // ranges of models/quantize.py:22-30
    switch (dtype) {
        case PYGIM_INT8: return 5;
        case PYGIM_INT16: return 10;
        case PYGIM_INT32: return 20;
        case PYGIM_FLT32: return 20;
        default: return -1;
    }
}

// Can the group's product run as ONE sweep that reads a caller-made slice-major copy and dequantises in its last
// store?  (unit weights, one matrix -- a single part or the merged one --, a panel plan without segment-kernel rows)
template <typename T>
static Part *fusable_part(Group *g) {
    if (!g->all_ones || !g_tune.fuse_windows || g_tune.panel_mode == 2 || g->h <= 4) return nullptr;
    Part *p = nullptr;
    if (g->parts.size() == 1) p = &g->parts[0];
    else if (g->merged && g_tune.merge_parts) p = g->merged.get();
    if (!p || p->vals || p->extra || !p->d_items || p->lp_panel.n_tasks > 0 || p->nrows == 0 || p->ncols == 0) return nullptr;
    if (!want_panel<T>(*p, (uint32_t)g->h, g->h)) return nullptr;
    constexpr uint32_t F = 128 / sizeof(T);
    const uint64_t nslices = ((uint64_t)g->h + F - 1) / F;
    if (((uint64_t)p->ncols * 128ull + 128ull) >= (1ull << 32)) return nullptr;  // the fused store rides the 32-bit gather modes
    if ((uint64_t)p->ncols * nslices * 128ull > (8ull << 30)) return nullptr;
    return p;
}

// the part whose LDS-staged plan can carry the conv layers' quantised aggregation with the dequantisation in its store
// (INT32 / FLT32 adjacency types; the per-column epilogue, if any, is applied in the same store)
template <typename T>
static Part *lds_fusable_part(Group *g) {
    if constexpr (!(std::is_same<T, float>::value || std::is_same<T, int32_t>::value || std::is_same<T, int8_t>::value || std::is_same<T, int16_t>::value)) return nullptr;
    if (!g->all_ones || g_tune.lds_mode == 2 || (int64_t)g->h / (sizeof(T) <= 2 ? 2 : 1) < g_tune.lds_min_width) return nullptr;
    if (g_tune.lds_mode == 0 && (g_tune.panel_mode != 0 || g_tune.csr_kernel != 0 || g_tune.force_vec_bytes != 0 || !g_tune.fuse_windows)) return nullptr;
    Part *p = nullptr;
    if (g->parts.size() == 1) p = &g->parts[0];
    else if (g->merged && g_tune.merge_parts) p = g->merged.get();
    if (!p || p->vals || p->extra || !p->lds_tiles || p->lds_wdelta || (p->lds_nw != 16 && !p->lds_is_code) || p->nrows == 0 || p->ncols == 0) return nullptr;
    if (p->lds_is_code && (!g_tune.lds_code || g_tune.lds_ablate)) return nullptr;
    if (p->lds_col_splits > 1) return nullptr;   // (partial sums per column range cannot be dequantised in the store)
    if (sizeof(T) <= 2 && (!p->lds_is_code || p->lds_nw != 8)) return nullptr;   // (INT8 / INT16: the 8-wave code stream's dequantising stores)
    return p;
}

static int launch_post(Group *g, float *out, hipStream_t st) {
    if (!g->post_mul) return 0;
    const uint64_t total = (uint64_t)g->total_rows * (uint64_t)g->h;
    if (total)
        hipLaunchKernelGGL(k_post_affine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, (uint64_t)g->total_rows,
                           (uint32_t)g->h, g->post_mul, g->post_add, g->post_relu);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int quant_run_t(Group *g, const float *X, int64_t ldx, float *out, float *scale_out, int log2_range, hipStream_t st) {
    const uint64_t rows = (uint64_t)g->total_cols, orows = (uint64_t)g->total_rows;
    const uint32_t h = (uint32_t)g->h;
    uint32_t *amax = (uint32_t *)(g->d_flags + 3);
    hipLaunchKernelGGL(k_zero_word, dim3(1), dim3(1), 0, st, amax);
    if (int rc = launch_absmax(X, ldx, rows, h, amax, st)) return rc;
    if (rows * h == 0 && scale_out) hipLaunchKernelGGL(k_zero_word, dim3(1), dim3(1), 0, st, (uint32_t *)scale_out);
    if constexpr (sizeof(T) == 4 || sizeof(T) <= 2) {
        if (Part *p = lds_fusable_part<T>(g)) {
            // FUSED on the LDS-staged kernel: the 256-byte-slice copy is written quantised from the float features, the kernel's
            // store dequantises (no quantised matrix, no integer result).  INT8 (the conv layers' own type, models/quantize.py:22-23):
            // the quantised values are staged as 16-bit numbers, 128 features to a slice, and summed by the INT16 stream
            using S = typename std::conditional<sizeof(T) == 1, int16_t, T>::type;
            constexpr uint32_t EPS = 256 / sizeof(S);
            const uint32_t nslices = (h + EPS - 1) / EPS;
            const uint64_t rows_pad = lds_rows_pad((uint64_t)p->ncols, p->lds_kc);
            void *xs = nullptr;
            {
                std::lock_guard<std::mutex> lk(g_ctx.mu);
                Context::XsBuf *b = nullptr;
                if (int rc = xs_buffer_locked(st, std::max<size_t>((size_t)rows_pad * nslices * 256, 256), &b)) return rc;
                b->src = nullptr;  // quantised values of a float matrix: never matched by x_unchanged
                xs = b->ptr;
            }
            XsPin pin;
            pin.hold(xs);
            const uint64_t threads = (uint64_t)p->ncols * nslices * 16;
            hipLaunchKernelGGL((k_slice_pack_quant<S, 16 / (int)sizeof(S), 4>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, X, ldx,
                               (uint32_t)p->ncols, h, nslices, amax, log2_range, (S *)xs, scale_out, (uint32_t)rows_pad);
            HIP_TRY(hipGetLastError());
            return launch_lds<T>(g, *p, (const T *)nullptr, (int64_t)h, (T *)out, (int64_t)h, h, false, st, xs, amax, log2_range);
        }
    }
    if (Part *p = fusable_part<T>(g)) {
        // FUSED: |max| reduction, then the slice-major copy is written quantised straight from the float features,
        // and every row's last panel item stores float(sum) * scale (no row-major quantised matrix, no integer result,
        // no separate quantise / dequantise passes)
        constexpr int VEC = 16 / (int)sizeof(T), LOG_LPR = 3;
        constexpr uint32_t F = VEC << LOG_LPR;
        const uint32_t nslices = (h + F - 1) / F;
        const size_t need = (size_t)p->ncols * nslices * F * sizeof(T);
        void *xs = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_ctx.mu);
            Context::XsBuf *b = nullptr;
            if (int rc = xs_buffer_locked(st, std::max<size_t>(need, 256), &b)) return rc;
            b->src = nullptr;  // holds quantised values of a float matrix: never matched by x_unchanged
            xs = b->ptr;
        }
        XsPin pin;
        pin.hold(xs);
        const uint64_t threads = (uint64_t)p->ncols * nslices * (1u << LOG_LPR);
        hipLaunchKernelGGL((k_slice_pack_quant<T, VEC, LOG_LPR>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, X, ldx,
                           (uint32_t)p->ncols, h, nslices, amax, log2_range, (T *)xs, scale_out, (uint32_t)p->ncols);
        HIP_TRY(hipGetLastError());
        // running sums between panels (rows whose entries span several panels) live in the group's integer buffer
        if (int rc = ensure(&g->oq, &g->oq_bytes, std::max<size_t>(orows * h * sizeof(T), 256))) return rc;
        g->pre_xs = xs;
        g->deq_out = out;
        g->deq_ld = (int64_t)h;
        g->deq_amax = amax;
        g->deq_log2 = log2_range;
        g->packed_buf = nullptr;
        const int rc = launch_block_main(g, *p, xs /* unused: pre_xs */, (int64_t)h, g->oq, (int64_t)h, (int64_t)h, false, st);
        g->pre_xs = nullptr;
        g->deq_out = nullptr;
        return rc;
    }
    // unfused: quantise -> product -> dequantise as three steps
    if (int rc = ensure(&g->xq, &g->xq_bytes, std::max<size_t>(rows * h * sizeof(T), 256))) return rc;
    if (int rc = ensure(&g->oq, &g->oq_bytes, std::max<size_t>(orows * h * sizeof(T), 256))) return rc;
    if (int rc = launch_quantize<T>(X, ldx, rows, h, amax, log2_range, (T *)g->xq, scale_out, st)) return rc;
    // the group's own dense split, as windows into the row-major quantised matrix
    const size_t nd = g->parts[0].dense_cols.size();
    std::vector<const void *> win(nd);
    std::vector<int64_t> lds(nd, (int64_t)h);
    int64_t off = 0;
    for (size_t j = 0; j < nd; j++) {
        win[j] = (const char *)g->xq + (size_t)off * sizeof(T);
        off += g->parts[0].dense_cols[j];
    }
    if (int rc = run_group_common(g, win.data(), lds.data(), false, g->oq, st)) return rc;
    if (int rc = launch_dequantize<T>((const T *)g->oq, orows * h, amax, log2_range, out, st)) return rc;
    return launch_post(g, out, st);
}

// A . Xq on ALREADY quantised features with the dequantisation in the sweep's last store (row-sharded multi-GPU
// aggregation: the quantised blocks were exchanged between the ranks, pygim_amd/dist.py RowShardAdj)
template <typename T>
static int dequant_run_t(Group *g, const void *Xq, int64_t ldx, float *out, const uint32_t *amax, int log2_range, hipStream_t st) {
    const uint64_t orows = (uint64_t)g->total_rows;
    const uint32_t h = (uint32_t)g->h;
    if constexpr (sizeof(T) == 4 || sizeof(T) <= 2) {
        if (Part *pl = lds_fusable_part<T>(g))  // the LDS-staged kernel packs the quantised rows itself and dequantises in its store
            return launch_lds<T>(g, *pl, (const T *)Xq, ldx, (T *)out, (int64_t)h, h, false, st, nullptr, amax, log2_range);
    }
    if (int rc = ensure(&g->oq, &g->oq_bytes, std::max<size_t>(orows * h * sizeof(T), 256))) return rc;
    Part *p = fusable_part<T>(g);
    // the dequantising store rides the sweep's slice-major gather modes: whole 16-byte pieces (so that the copy is made from
    // rows that are never over-read) and not the narrow-row shortcut that gathers from the caller's row-major matrix
    constexpr uint32_t V = 16 / sizeof(T), F = V * 8;
    const bool narrow_rowmajor = (h + F - 1) / F == 1 && (size_t)ldx * sizeof(T) < 128;
    if (p && h % V == 0 && g_tune.panel_pack && !narrow_rowmajor) {
        g->deq_out = out;
        g->deq_ld = (int64_t)h;
        g->deq_amax = amax;
        g->deq_log2 = log2_range;
        const int rc = launch_block_any(g, *p, Xq, ldx, g->oq, (int64_t)h, (int64_t)h, false, st);
        g->deq_out = nullptr;
        return rc;
    }
    const size_t nd = g->parts[0].dense_cols.size();
    std::vector<const void *> win(nd);
    std::vector<int64_t> lds(nd, ldx);
    int64_t off = 0;
    for (size_t j = 0; j < nd; j++) {
        win[j] = (const char *)Xq + (size_t)off * sizeof(T);
        off += g->parts[0].dense_cols[j];
    }
    if (int rc = run_group_common(g, win.data(), lds.data(), false, g->oq, st)) return rc;
    return launch_dequantize<T>((const T *)g->oq, orows * h, amax, log2_range, out, st);
}

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
    else if (n == "lds_threads") slot = &g_tune.lds_threads;
    else if (n == "lds_waves") slot = &g_tune.lds_waves;
    else if (n == "lds_ablate") slot = &g_tune.lds_ablate;
    else if (n == "lds_round_tiles") slot = &g_tune.lds_round_tiles;
    else if (n == "lds_code") slot = &g_tune.lds_code;
    else if (n == "lds_code_nbuf") slot = &g_tune.lds_code_nbuf;
    else if (n == "lds_code_waves") slot = &g_tune.lds_code_waves;
    else if (n == "lds_fail") slot = &g_tune.lds_fail;
    else if (n == "lds_code_exp") slot = &g_tune.lds_code_exp;
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
    }
    if (hipDeviceSynchronize() != hipSuccess) return bail(fail(PYGIM_ERR_HIP, "sync after create"));
    g->timers[4] = now_ms() - t0;
    {
        std::lock_guard<std::mutex> lk(g_ctx.mu);
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
