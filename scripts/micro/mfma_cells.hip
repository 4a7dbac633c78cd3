// Prototype (round 6): the DENSE 32 x 32 cells of a 0/1 (multigraph: small counts) adjacency contracted on the matrix cores --
// C[32 rb .. +32, 0 .. h) += A_cell[32 x 32, int8] . X[32 cb .. +32, 0 .. h) [int8] with v_mfma_i32_32x32x32_i8 into int32 (the low byte of the
// sum is the modular int8 sum the path's INT8 contract asks for, models/quantize.py:22-23 / pyg_gcn_conv.py:130-137).  Everything sparser than the
// threshold stays with the library's code stream.  Built stand-alone (scripts/exp_mfma_cells.py loads it with ctypes) so that the decision --
// keep and integrate, or close with the measurement -- is made with numbers before the library grows a fourth product path.
//
// Operand images are laid out in MFMA REGISTER ORDER by the host side, so that every operand load is one coalesced 16-byte load per lane:
//   a_blk[p][hh][r][j]      = A[32 rb + r][32 cb + 16 hh + j]           (lane l = 32 hh + r holds k = 16 hh .. 16 hh + 15 of row r)
//   xt[cb][t][hh][n][j]     = X[32 cb + 16 hh + j][32 t + n]            (lane l = 32 hh + n holds k = 16 hh .. + 15 of feature 32 t + n)
//   C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)   (cdna_hip_programming.md, "Fragment layout")
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/micro/mfma_cells.hip -o scripts/micro/libmfma_cells.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// one wave per row block (32 rows) x TN feature tiles of 32; 4 waves per workgroup; the wave walks the row block's dense cells
template <int TN>
__global__ __launch_bounds__(256) void k_mfma_cells(const int *__restrict__ rb_ptr, const int *__restrict__ cb_idx, const v4i *__restrict__ a_blk,
                                                    const v4i *__restrict__ xt, int *__restrict__ c, int nrb, int h_tiles, int64_t ldc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rb = blockIdx.x * 4 + wave;
    const int t0 = blockIdx.y * TN;
    if (rb >= nrb) return;
    const int p0 = rb_ptr[rb], p1 = rb_ptr[rb + 1];
    if (p0 == p1) return;   // (C is zeroed by the caller)
    v16i acc[TN];
#pragma unroll
    for (int t = 0; t < TN; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0;
    v4i a = a_blk[(size_t)p0 * 64 + lane];
    int cb = cb_idx[p0];
    for (int p = p0; p < p1; p++) {
        const v4i *xb = xt + ((size_t)cb * h_tiles + t0) * 64 + lane;
        v4i b[TN];
#pragma unroll
        for (int t = 0; t < TN; t++) b[t] = xb[t * 64];
        const v4i a_now = a;
        if (p + 1 < p1) {   // the next cell's operands while this one's products run
            a = a_blk[(size_t)(p + 1) * 64 + lane];
            cb = cb_idx[p + 1];
        }
#pragma unroll
        for (int t = 0; t < TN; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_now, b[t], acc[t], 0, 0, 0);
    }
    const int n = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < TN; t++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            c[(int64_t)(32 * rb + row) * ldc + 32 * (t0 + t) + n] = acc[t][reg];
        }
}

// Second form: the B operand shared through LDS.  The first form reads a cell's 8 KB of X (h = 256) out of the L2 for EVERY cell: 10 KB per cell, 8.4 TB/s, 450-750
// CU-cycles per cell against the 64 the matrix cores need (profiles/r06_mfma_cells.txt).  Here a 512-thread workgroup owns 8 consecutive row blocks (256 rows) and
// walks the column blocks any of them has a dense cell in: the column block's tile of X is staged once (double-buffered, 2 x 8 KB of LDS), and each of the 8 waves
// that has a cell there contracts its own 32 x 32 panel against it.
template <int NT>
__global__ __launch_bounds__(512) void k_mfma_cells_lds(const int *__restrict__ srb_ptr, const int *__restrict__ pair_cb, const int *__restrict__ pair_cell,
                                                        const v4i *__restrict__ a_blk, const v4i *__restrict__ xt, int *__restrict__ c, int nrb, int64_t ldc) {
    static_assert(NT * 64 == 512, "one 16-byte piece of the tile per thread");
    __shared__ v4i bt[2][NT * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int srb = blockIdx.x;
    const int q0 = srb_ptr[srb], q1 = srb_ptr[srb + 1];
    if (q0 == q1) return;
    v16i acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0;
    bt[0][tid] = xt[(size_t)pair_cb[q0] * (NT * 64) + tid];
    int cell = pair_cell[(size_t)q0 * 8 + wave];
    v4i a = {0, 0, 0, 0};
    if (cell >= 0) a = a_blk[(size_t)cell * 64 + lane];
    bool any = false;
    __syncthreads();
    for (int q = q0; q < q1; q++) {
        const int cur = (q - q0) & 1;
        v4i b_next = {0, 0, 0, 0}, a_next = {0, 0, 0, 0};
        int cell_next = -1;
        if (q + 1 < q1) {
            b_next = xt[(size_t)pair_cb[q + 1] * (NT * 64) + tid];
            cell_next = pair_cell[(size_t)(q + 1) * 8 + wave];
            if (cell_next >= 0) a_next = a_blk[(size_t)cell_next * 64 + lane];
        }
        if (cell >= 0) {
            any = true;
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bt[cur][t * 64 + lane], acc[t], 0, 0, 0);
        }
        if (q + 1 < q1) bt[cur ^ 1][tid] = b_next;
        __syncthreads();
        cell = cell_next;
        a = a_next;
    }
    const int rb = srb * 8 + wave;
    if (!any || rb >= nrb) return;
    const int n = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            c[(int64_t)(32 * rb + row) * ldc + 32 * t + n] = acc[t][reg];
        }
}

// Third form: the second with the operands of the next D pairs in flight.  One workgroup fits a CU (8 waves x 128 accumulators), so nothing but the wave's own
// prefetch hides the ~1.2 us a load takes under this traffic: with one pair ahead (second form) an iteration is 1.25 us for 0.11 us of matrix-core work.
template <int NT, int D>
__global__ __launch_bounds__(512) void k_mfma_cells_deep(const int *__restrict__ srb_ptr, const int *__restrict__ pair_cb, const int *__restrict__ pair_cell,
                                                         const v4i *__restrict__ a_blk, const v4i *__restrict__ xt, int *__restrict__ c, int nrb, int64_t ldc) {
    static_assert(NT * 64 == 512, "one 16-byte piece of the tile per thread");
    __shared__ v4i bt[2][NT * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int srb = blockIdx.x;
    const int q0 = srb_ptr[srb], q1 = srb_ptr[srb + 1];
    if (q0 == q1) return;
    v16i acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0;
    // (the operand addresses come out of two small index arrays: THEIR loads run a further D pairs ahead -- an operand load that waits for its own index
    // load stalls the wave for a whole memory round trip per pair, which is what the first version of this kernel did: 0.86 -> 0.6 ms only)
    v4i bq[D], aq[D];
    int cq[D], icb[D], icell[D];
#pragma unroll
    for (int s = 0; s < D; s++) {
        bq[s] = v4i{0, 0, 0, 0};
        aq[s] = v4i{0, 0, 0, 0};
        cq[s] = -1;
        icb[s] = 0;
        icell[s] = -1;
        if (q0 + s < q1) {
            bq[s] = xt[(size_t)pair_cb[q0 + s] * (NT * 64) + tid];
            cq[s] = pair_cell[(size_t)(q0 + s) * 8 + wave];
            if (cq[s] >= 0) aq[s] = a_blk[(size_t)cq[s] * 64 + lane];
        }
        if (q0 + s + D < q1) {
            icb[s] = pair_cb[q0 + s + D];
            icell[s] = pair_cell[(size_t)(q0 + s + D) * 8 + wave];
        }
    }
    bool any = false;
    for (int q = q0; q < q1; q += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            if (q + s < q1) {   // (uniform over the workgroup)
                const int cur = s & 1;   // D is even: consecutive pairs alternate buffers across the outer loop too
                bt[cur][tid] = bq[s];
                const v4i a = aq[s];
                const int cell = cq[s];
                __syncthreads();
                const int qn = q + s + D;
                if (qn < q1) {
                    bq[s] = xt[(size_t)icb[s] * (NT * 64) + tid];
                    cq[s] = icell[s];
                    if (cq[s] >= 0) aq[s] = a_blk[(size_t)cq[s] * 64 + lane];
                    if (qn + D < q1) {
                        icb[s] = pair_cb[qn + D];
                        icell[s] = pair_cell[(size_t)(qn + D) * 8 + wave];
                    }
                }
                if (cell >= 0) {
                    any = true;
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bt[cur][t * 64 + lane], acc[t], 0, 0, 0);
                }
            }
        }
    }
    const int rb = srb * 8 + wave;
    if (!any || rb >= nrb) return;
    const int n = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            c[(int64_t)(32 * rb + row) * ldc + 32 * t + n] = acc[t][reg];
        }
}

// X [ncols, h] int8 row-major -> xt[cb][t][hh][n][j] (one thread per 16 output bytes)
__global__ void k_pack_xt(const int8_t *__restrict__ x, int64_t ldx, int ncols, int h_tiles, int8_t *__restrict__ xt, int ncb) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // ((cb * h_tiles + t) * 2 + hh) * 32 + n
    const int64_t total = (int64_t)ncb * h_tiles * 64;
    if (id >= total) return;
    const int n = id & 31, hh = (id >> 5) & 1;
    const int64_t ct = id >> 6;
    const int t = (int)(ct % h_tiles);
    const int64_t cb = ct / h_tiles;
    int8_t v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int64_t col = cb * 32 + 16 * hh + j;
        v[j] = col < ncols ? x[col * ldx + 32 * t + n] : (int8_t)0;
    }
    *(v4i *)(xt + id * 16) = *(const v4i *)v;
}

// out8[i] = (int8)(sparse8[i] + dense32[i])   -- the modular int8 sum of the two halves
__global__ void k_combine(const int8_t *__restrict__ sparse8, const int *__restrict__ dense32, int8_t *__restrict__ out8, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out8[i] = (int8_t)((int)sparse8[i] + dense32[i]);
}

extern "C" {
int mfma_cells_run(const int *rb_ptr, const int *cb_idx, const void *a_blk, const void *xt, int *c, int nrb, int h_tiles, int64_t ldc, int tn, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((nrb + 3) / 4, h_tiles / tn);
    if (tn == 8) hipLaunchKernelGGL((k_mfma_cells<8>), grid, dim3(256), 0, st, rb_ptr, cb_idx, (const v4i *)a_blk, (const v4i *)xt, c, nrb, h_tiles, ldc);
    else if (tn == 4) hipLaunchKernelGGL((k_mfma_cells<4>), grid, dim3(256), 0, st, rb_ptr, cb_idx, (const v4i *)a_blk, (const v4i *)xt, c, nrb, h_tiles, ldc);
    else if (tn == 2) hipLaunchKernelGGL((k_mfma_cells<2>), grid, dim3(256), 0, st, rb_ptr, cb_idx, (const v4i *)a_blk, (const v4i *)xt, c, nrb, h_tiles, ldc);
    else return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int mfma_cells_lds_run(const int *srb_ptr, const int *pair_cb, const int *pair_cell, const void *a_blk, const void *xt, int *c, int nsrb, int nrb, int h_tiles, int64_t ldc,
                       void *stream) {
    if (h_tiles != 8) return 1;
    hipLaunchKernelGGL((k_mfma_cells_lds<8>), dim3(nsrb), dim3(512), 0, (hipStream_t)stream, srb_ptr, pair_cb, pair_cell, (const v4i *)a_blk, (const v4i *)xt, c, nrb, ldc);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int mfma_cells_deep_run(const int *srb_ptr, const int *pair_cb, const int *pair_cell, const void *a_blk, const void *xt, int *c, int nsrb, int nrb, int h_tiles, int64_t ldc,
                        int depth, void *stream) {
    if (h_tiles != 8) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (depth == 4) hipLaunchKernelGGL((k_mfma_cells_deep<8, 4>), dim3(nsrb), dim3(512), 0, st, srb_ptr, pair_cb, pair_cell, (const v4i *)a_blk, (const v4i *)xt, c, nrb, ldc);
    else if (depth == 8) hipLaunchKernelGGL((k_mfma_cells_deep<8, 8>), dim3(nsrb), dim3(512), 0, st, srb_ptr, pair_cb, pair_cell, (const v4i *)a_blk, (const v4i *)xt, c, nrb, ldc);
    else if (depth == 6) hipLaunchKernelGGL((k_mfma_cells_deep<8, 6>), dim3(nsrb), dim3(512), 0, st, srb_ptr, pair_cb, pair_cell, (const v4i *)a_blk, (const v4i *)xt, c, nrb, ldc);
    else return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int mfma_pack_xt(const void *x, int64_t ldx, int ncols, int h_tiles, void *xt, int ncb, void *stream) {
    const int64_t total = (int64_t)ncb * h_tiles * 64;
    hipLaunchKernelGGL(k_pack_xt, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const int8_t *)x, ldx, ncols, h_tiles, (int8_t *)xt, ncb);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int mfma_combine(const void *sparse8, const int *dense32, void *out8, int64_t n, void *stream) {
    hipLaunchKernelGGL(k_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const int8_t *)sparse8, dense32, (int8_t *)out8, n);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
}
