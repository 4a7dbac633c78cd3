// kernels.hpp -- gfx950 (CDNA4) device code of the aggregation backend.
//
// What is computed (the reference's device loop, spmm_default/dpu_kernels/
// spmm_mul_csr_dpu.c:108-126 and spmm_mul_coo_dpu.c:142-390):
//     C[r, k] = sum over stored entries e of row r, in stored order, of
//               val[e] * X[col[e], k]
// in val_dt arithmetic (spmm_default/support/common.h:39-60): integers are
// two's-complement modular at the element width, floats are summed in stored
// order.  How it is computed is CDNA4-specific.  Kernels, most used first:
//   k_csr_panel / panel_sweep   the general path: features cut into 128-byte slices (one
//       cache line per gathered row), columns into L2-sized panels, one launch per panel over
//       length-sorted (row, panel) work items; an 8-lane group per item, a whole wave for long
//       items; XCD-aware block->slice mapping; X read from a slice-major copy (k_slice_pack).
//   k_csr_wide, k_csr_sub       row-per-wave / rows-per-wave gathers for operands that are not
//       16-byte aligned (odd widths, grande's 8-byte-padded windows).
//   k_long_segments + k_long_reduce   rows far too long for one wave: fixed segments on a forked
//       stream, partial sums added in segment order (deterministic, no atomics).
//   k_coo_wide + k_coo_fixup    native COO: equal-nnz chunks per wave, carries fixed up in order.
//   k_absmax_bits / k_quantize / k_dequantize   the conv layers' quantiser around the product.
// Sums stay in registers in stored order wherever one lane group owns a row, which makes float
// results bit-identical to a sequential CPU loop there (the file is built with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace pygim {

// ---------------------------------------------------------------------------
// element traits: accumulator type (modular for integers) and bit-broadcast
// ---------------------------------------------------------------------------
template <typename T> struct AccOf { using type = T; };
template <> struct AccOf<int8_t> { using type = uint32_t; };
template <> struct AccOf<int16_t> { using type = uint32_t; };
template <> struct AccOf<int32_t> { using type = uint32_t; };
template <> struct AccOf<int64_t> { using type = uint64_t; };

template <typename T> __device__ __forceinline__ typename AccOf<T>::type to_acc(T v) {
    using A = typename AccOf<T>::type;
    if constexpr (std::is_floating_point<T>::value) return v;
    else if constexpr (sizeof(T) == 8) return (A)v;
    else return (A)(int32_t)v;  // sign-extend, then wrap
}
template <typename T> __device__ __forceinline__ T from_acc(typename AccOf<T>::type a) {
    if constexpr (std::is_floating_point<T>::value) return a;
    else return (T)a;  // truncation == reduction mod 2^bits
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// broadcast the value held by lane `src` (wave-uniform index) to a scalar
template <typename T> __device__ __forceinline__ T bcast_lane(T v, int src) {
    if constexpr (sizeof(T) == 8) {
        union { T t; uint32_t u[2]; } in, out;
        in.t = v;
        out.u[0] = __builtin_amdgcn_readlane(in.u[0], src);
        out.u[1] = __builtin_amdgcn_readlane(in.u[1], src);
        return out.t;
    } else if constexpr (sizeof(T) == 4) {
        union { T t; uint32_t u; } in, out;
        in.t = v;
        out.u = __builtin_amdgcn_readlane(in.u, src);
        return out.t;
    } else {
        int32_t w = (int32_t)v;
        return (T)__builtin_amdgcn_readlane(w, src);
    }
}

// per-lane shuffle inside a power-of-two lane group
template <typename T> __device__ __forceinline__ T shfl_lane(T v, int src) {
    if constexpr (sizeof(T) == 8) {
        union { T t; int u[2]; } in, out;
        in.t = v;
        out.u[0] = __shfl(in.u[0], src);
        out.u[1] = __shfl(in.u[1], src);
        return out.t;
    } else if constexpr (sizeof(T) == 4) {
        union { T t; int u; } in, out;
        in.t = v;
        out.u = __shfl(in.u, src);
        return out.t;
    } else {
        return (T)__shfl((int)v, src);
    }
}


template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int mask) {
    if constexpr (sizeof(T) == 8) {
        union { T t; int u[2]; } in, out;
        in.t = v;
        out.u[0] = __shfl_xor(in.u[0], mask);
        out.u[1] = __shfl_xor(in.u[1], mask);
        return out.t;
    } else {
        union { T t; int u; } in, out;
        in.t = v;
        out.u = __shfl_xor(in.u, mask);
        return out.t;
    }
}

// 16-byte vector of column ids that is only 4-byte aligned (segment starts are arbitrary)
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));

// four values of T that are only element-aligned
template <typename T> struct Vec4U { typedef T type __attribute__((ext_vector_type(4), aligned(sizeof(T)))); };

template <typename T, int VEC> struct VecOf { typedef T type __attribute__((ext_vector_type(VEC))); };
template <typename T> struct VecOf<T, 1> { typedef T type; };

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T *p, T (&out)[VEC]) {
    if constexpr (VEC == 1) {
        out[0] = *p;
    } else {
        using V = typename VecOf<T, VEC>::type;
        V v = *reinterpret_cast<const V *>(p);
#pragma unroll
        for (int k = 0; k < VEC; k++) out[k] = v[k];
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T *p, const T (&in)[VEC]) {
    if constexpr (VEC == 1) {
        *p = in[0];
    } else {
        using V = typename VecOf<T, VEC>::type;
        V v;
#pragma unroll
        for (int k = 0; k < VEC; k++) v[k] = in[k];
        *reinterpret_cast<V *>(p) = v;
    }
}

// streaming (non-temporal) forms: C rows and index arrays are touched once per sweep and must
// not displace the X panel from the L2
template <typename T, int VEC>
__device__ __forceinline__ void load_vec_nt(const T *p, T (&out)[VEC]) {
    if constexpr (VEC == 1) {
        out[0] = __builtin_nontemporal_load(p);
    } else {
        using V = typename VecOf<T, VEC>::type;
        V v = __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
#pragma unroll
        for (int k = 0; k < VEC; k++) out[k] = v[k];
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec_nt(T *p, const T (&in)[VEC]) {
    if constexpr (VEC == 1) {
        __builtin_nontemporal_store(in[0], p);
    } else {
        using V = typename VecOf<T, VEC>::type;
        V v;
#pragma unroll
        for (int k = 0; k < VEC; k++) v[k] = in[k];
        __builtin_nontemporal_store(v, reinterpret_cast<V *>(p));
    }
}

// acc[k] += a * x[k] in val_dt arithmetic.  Floats: separate multiply and add
// (the file is built with -ffp-contract=off) so that the rounding matches a
// CPU loop compiled without FMA contraction.
template <typename T, int VEC>
__device__ __forceinline__ void axpy(typename AccOf<T>::type (&acc)[VEC], typename AccOf<T>::type a,
                                     const T (&x)[VEC]) {
#pragma unroll
    for (int k = 0; k < VEC; k++) acc[k] += a * to_acc<T>(x[k]);
}
template <typename T, int VEC>
__device__ __forceinline__ void add_only(typename AccOf<T>::type (&acc)[VEC], const T (&x)[VEC]) {
#pragma unroll
    for (int k = 0; k < VEC; k++) acc[k] += to_acc<T>(x[k]);
}

// write VEC results at C[f0 .. f0+VEC) of a row whose valid width is w
template <typename T, int VEC>
__device__ __forceinline__ void emit(T *crow, uint32_t f0, uint32_t w, typename AccOf<T>::type (&acc)[VEC],
                                     bool accumulate) {
    if (f0 >= w) return;
    T *p = crow + f0;
    if (f0 + VEC <= w) {
        T o[VEC];
        if (accumulate) {
            T old[VEC];
            load_vec<T, VEC>(p, old);
#pragma unroll
            for (int k = 0; k < VEC; k++) o[k] = from_acc<T>(to_acc<T>(old[k]) + acc[k]);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; k++) o[k] = from_acc<T>(acc[k]);
        }
        store_vec<T, VEC>(p, o);
    } else {
#pragma unroll
        for (int k = 0; k < VEC; k++)
            if (f0 + k < w) p[k] = accumulate ? from_acc<T>(to_acc<T>(p[k]) + acc[k]) : from_acc<T>(acc[k]);
    }
}

// ---------------------------------------------------------------------------
// One wave accumulates the stored entries [s, e) of one row over the 64*VEC
// features starting at f_base (wave-uniform s, e).  Gathers are issued 8 deep.
// ---------------------------------------------------------------------------
template <typename T, int VEC>
__device__ __forceinline__ void wave_segment(typename AccOf<T>::type (&acc)[VEC], uint32_t s, uint32_t e,
                                             const uint32_t *__restrict__ colind, const T *__restrict__ vals,
                                             const T *__restrict__ xlane /* X + f0 of this lane */,
                                             int64_t ldx, bool lane_on, int lane) {
    constexpr int DEPTH = 8;
    uint32_t mycol_next = (s + lane < e) ? __builtin_nontemporal_load(colind + s + lane) : 0u;
    T myval_next = T(1);
    if (vals) myval_next = (s + lane < e) ? __builtin_nontemporal_load(vals + s + lane) : T(0);
    for (uint32_t e0 = s; e0 < e; e0 += 64) {
        const uint32_t n = min(64u, e - e0);
        const uint32_t mycol = mycol_next;
        const T myval = myval_next;
        // prefetch the next 64 column ids while this batch is gathered
        const uint32_t nx = e0 + 64 + lane;
        if (e0 + 64 < e) {
            mycol_next = (nx < e) ? __builtin_nontemporal_load(colind + nx) : 0u;
            if (vals) myval_next = (nx < e) ? __builtin_nontemporal_load(vals + nx) : T(0);
        }
        uint32_t j = 0;
        for (; j + DEPTH <= n; j += DEPTH) {
            T x[DEPTH][VEC];
#pragma unroll
            for (int u = 0; u < DEPTH; u++) {
                const uint32_t c = __builtin_amdgcn_readlane(mycol, j + u);
                if (lane_on) load_vec<T, VEC>(xlane + (int64_t)c * ldx, x[u]);
            }
            if (lane_on) {
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    if (vals) axpy<T, VEC>(acc, to_acc<T>(bcast_lane<T>(myval, j + u)), x[u]);
                    else add_only<T, VEC>(acc, x[u]);
                }
            }
        }
        for (; j < n; j++) {
            T x[VEC];
            const uint32_t c = __builtin_amdgcn_readlane(mycol, j);
            if (lane_on) {
                load_vec<T, VEC>(xlane + (int64_t)c * ldx, x);
                if (vals) axpy<T, VEC>(acc, to_acc<T>(bcast_lane<T>(myval, j)), x);
                else add_only<T, VEC>(acc, x);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// CSR, one row per wave ("wide": the row of X spans >= 32 lanes).
// grid.x = ceil(nrows / waves_per_block), grid.y = ceil(w / (64*VEC)).
// Rows with more than long_thresh entries are left to the long-row kernels.
// ---------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_csr_wide(const uint32_t *__restrict__ rowptr,
                                                  const uint32_t *__restrict__ colind,
                                                  const T *__restrict__ vals, const T *__restrict__ X,
                                                  int64_t ldx, T *__restrict__ C, int64_t ldc, uint32_t nrows,
                                                  uint32_t w, uint32_t long_thresh, int accumulate) {
    using A = typename AccOf<T>::type;
    const int lane = threadIdx.x & 63;
    const uint32_t row = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (row >= nrows) return;
    const uint32_t s = rfl(rowptr[row]), e = rfl(rowptr[row + 1]);
    if (e - s > long_thresh) return;
    const uint32_t f0 = blockIdx.y * (64 * VEC) + lane * VEC;
    const bool lane_on = f0 < w;
    A acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) acc[k] = A(0);
    wave_segment<T, VEC>(acc, s, e, colind, vals, X + f0, ldx, lane_on, lane);
    emit<T, VEC>(C + (int64_t)row * ldc, f0, w, acc, accumulate != 0);
}

// Long rows: task t = (row, s, e) -> partial[t, 0:w]
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_long_segments(const uint32_t *__restrict__ tasks /* 3 per task */,
                                                       uint32_t ntasks, const uint32_t *__restrict__ colind,
                                                       const T *__restrict__ vals, const T *__restrict__ X,
                                                       int64_t ldx, T *__restrict__ partial, uint32_t w) {
    using A = typename AccOf<T>::type;
    const int lane = threadIdx.x & 63;
    const uint32_t t = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (t >= ntasks) return;
    const uint32_t s = rfl(tasks[3 * t + 1]), e = rfl(tasks[3 * t + 2]);
    const uint32_t f0 = blockIdx.y * (64 * VEC) + lane * VEC;
    const bool lane_on = f0 < w;
    A acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) acc[k] = A(0);
    wave_segment<T, VEC>(acc, s, e, colind, vals, X + f0, ldx, lane_on, lane);
    if (lane_on) {
        T *p = partial + (int64_t)t * w + f0;  // partial is dense [ntasks, w], element-aligned only
#pragma unroll
        for (int k = 0; k < VEC; k++)
            if (f0 + k < w) p[k] = from_acc<T>(acc[k]);
    }
}

// Long rows: C[row] (+)= sum of its segment partials, in segment order.
// desc = (row, first_task, n_tasks) per long row; one thread per (long row, feature).
template <typename T>
__global__ __launch_bounds__(256) void k_long_reduce(const uint32_t *__restrict__ desc, uint32_t nlong,
                                                     const T *__restrict__ partial, T *__restrict__ C,
                                                     int64_t ldc, uint32_t w, int accumulate) {
    using A = typename AccOf<T>::type;
    const uint32_t i = blockIdx.y;
    if (i >= nlong) return;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= w) return;
    const uint32_t row = desc[3 * i], t0 = desc[3 * i + 1], nt = desc[3 * i + 2];
    A acc = to_acc<T>(partial[(int64_t)t0 * w + k]);
    for (uint32_t t = 1; t < nt; t++) acc += to_acc<T>(partial[(int64_t)(t0 + t) * w + k]);
    T *c = C + (int64_t)row * ldc + k;
    *c = accumulate ? from_acc<T>(to_acc<T>(*c) + acc) : from_acc<T>(acc);
}

// ---------------------------------------------------------------------------
// CSR, several rows per wave ("sub-wave": the row of X needs <= 32 lanes).
// A lane group of 2^log_lpr lanes owns one row; groups walk their rows in
// lock-step up to the longest row of the wave.
// ---------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_csr_sub(const uint32_t *__restrict__ rowptr,
                                                 const uint32_t *__restrict__ colind,
                                                 const T *__restrict__ vals, const T *__restrict__ X,
                                                 int64_t ldx, T *__restrict__ C, int64_t ldc, uint32_t nrows,
                                                 uint32_t w, uint32_t long_thresh, int accumulate, int log_lpr) {
    using A = typename AccOf<T>::type;
    const int lane = threadIdx.x & 63;
    const int lpr = 1 << log_lpr;
    const int li = lane & (lpr - 1);
    const int gbase = lane & ~(lpr - 1);
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t row64 = (uint64_t)wave * (64 >> log_lpr) + (lane >> log_lpr);
    const bool row_ok = row64 < nrows;
    const uint32_t row = row_ok ? (uint32_t)row64 : 0u;
    uint32_t s = 0, len = 0;
    if (row_ok) {
        s = rowptr[row];
        len = rowptr[row + 1] - s;
    }
    const bool is_long = len > long_thresh;
    if (is_long) len = 0;
    const uint32_t f0 = li * VEC;
    const bool lane_on = row_ok && f0 < w;
    // longest row of the wave -> uniform trip count
    uint32_t maxlen = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, off));
    maxlen = rfl(maxlen);
    A acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) acc[k] = A(0);
    const T *xlane = X + f0;
    for (uint32_t e0 = 0; e0 < maxlen; e0 += lpr) {
        const bool have = e0 + li < len;
        const uint32_t mycol = have ? colind[s + e0 + li] : 0u;
        T myval = T(1);
        if (vals) myval = have ? vals[s + e0 + li] : T(0);
        const int n = (len > e0) ? (int)min((uint32_t)lpr, len - e0) : 0;
        for (int j = 0; j < lpr; j++) {
            const uint32_t c = (uint32_t)__shfl((int)mycol, gbase + j);
            const T v = vals ? shfl_lane<T>(myval, gbase + j) : T(1);
            if (lane_on && j < n) {
                T x[VEC];
                load_vec<T, VEC>(xlane + (int64_t)c * ldx, x);
                if (vals) axpy<T, VEC>(acc, to_acc<T>(v), x);
                else add_only<T, VEC>(acc, x);
            }
        }
    }
    if (row_ok && !is_long) emit<T, VEC>(C + (int64_t)row * ldc, f0, w, acc, accumulate != 0);
}

// ---------------------------------------------------------------------------
// CSR "vector" form for the SpMV end of the path (rows of X of at most 4 elements: spmv_sparseP with few
// right-hand sides).  A group of 2^log_g lanes owns one row and its lanes stride over the row's ENTRIES:
// column ids (and weights) are read coalesced, every lane gathers its own tiny row of X (dense, so the whole
// operand sits in L2 / L1), the lanes' partial sums are added in a fixed butterfly.  64 gathers per wave
// instruction instead of the sweep's 8, no padded copy.  Integers exact; floats summed in a different (fixed)
// order than the sequential loop, inside the 1e-5 bound.
// ---------------------------------------------------------------------------
template <typename T> struct Vec2U { typedef T type __attribute__((ext_vector_type(2), aligned(sizeof(T)))); };
template <typename T, int W, bool HAS_VALS>
__global__ __launch_bounds__(256) void k_csr_vec(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
                                                 const T *__restrict__ vals, const T *__restrict__ X, int64_t ldx,
                                                 T *__restrict__ C, int64_t ldc, uint32_t nrows, int accumulate, int log_g) {
    static_assert(W >= 1 && W <= 4, "rows of at most 4 elements");
    using A = typename AccOf<T>::type;
    const int lane = threadIdx.x & 63;
    const uint32_t g = 1u << log_g;
    const uint32_t li = (uint32_t)lane & (g - 1);
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t row64 = wave * (64u >> log_g) + ((uint32_t)lane >> log_g);
    const bool row_ok = row64 < nrows;
    const uint32_t row = row_ok ? (uint32_t)row64 : 0u;
    uint32_t s = 0, e = 0;
    if (row_ok) {
        s = rowptr[row];
        e = rowptr[row + 1];
    }
    A acc[W];
#pragma unroll
    for (int j = 0; j < W; j++) acc[j] = A(0);
    for (uint32_t k = s + li; k < e; k += g) {
        const uint32_t c = __builtin_nontemporal_load(colind + k);
        const T *xr = X + (int64_t)c * ldx;
        // the whole row of X in one (element-aligned) load: one gather per entry whatever W is
        T x[W];
        if constexpr (W == 1) {
            x[0] = xr[0];
        } else if constexpr (W == 2) {
            const typename Vec2U<T>::type q = *reinterpret_cast<const typename Vec2U<T>::type *>(xr);
            x[0] = q[0]; x[1] = q[1];
        } else if constexpr (W == 3) {  // 2 + 1: never reaches past the row
            const typename Vec2U<T>::type q = *reinterpret_cast<const typename Vec2U<T>::type *>(xr);
            x[0] = q[0]; x[1] = q[1]; x[2] = xr[2];
        } else {
            const typename Vec4U<T>::type q = *reinterpret_cast<const typename Vec4U<T>::type *>(xr);
            x[0] = q[0]; x[1] = q[1]; x[2] = q[2]; x[3] = q[3];
        }
        if constexpr (HAS_VALS) {
            const A v = to_acc<T>(__builtin_nontemporal_load(vals + k));
#pragma unroll
            for (int j = 0; j < W; j++) acc[j] += v * to_acc<T>(x[j]);
        } else {
#pragma unroll
            for (int j = 0; j < W; j++) acc[j] += to_acc<T>(x[j]);
        }
    }
    // butterfly over the group's lanes (fixed order -> deterministic)
    for (uint32_t off = g >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int j = 0; j < W; j++) acc[j] += shfl_xor_t<A>(acc[j], (int)off);
    }
    if (row_ok && li == 0) {
        T *c = C + (int64_t)row * ldc;
#pragma unroll
        for (int j = 0; j < W; j++) c[j] = accumulate ? from_acc<T>(to_acc<T>(c[j]) + acc[j]) : from_acc<T>(acc[j]);
    }
}

// ---------------------------------------------------------------------------
// CSR, L2-blocked ("panel") form -- the fast path for wide feature rows.
//
// Why: one gathered row of X is h*sizeof(T) bytes (1 KiB at h=256 f32) and X does
// not fit the 4 MiB L2 of an XCD, so the row-per-wave kernel is bound by the
// Infinity-Cache/HBM gather rate.  Here the work is cut twice:
//   * features into 128-byte SLICES (one cache line per gathered row); slice s is
//     handled by workgroups with blockIdx % nslices == s, which the dispatcher's
//     round-robin places on one XCD when nslices == 8 -- that XCD's L2 then only
//     ever holds its own slice of X (speed only; results do not depend on placement);
//   * columns into PANELS sized so that panel x slice fits the L2; one launch per
//     panel, every launch sweeps all rows restricted to that panel's columns.
// A lane group of 2^LOG_LPR lanes (8 for 16-byte vectors) owns one (row, slice);
// a wave carries 64>>LOG_LPR rows.  Rows are visited in degree-sorted order
// (perm) so the rows that share a wave have similar lengths.  For panels after
// the first the accumulator STARTS from C, and entries are added in stored order,
// so the float result is bit-identical to one sequential pass over the row.
// seg_begin/seg_end: first/last stored entry of (sorted row i, this panel).
// ---------------------------------------------------------------------------
// broadcast lane j of every 8-lane group to the whole group with two DPP moves (no LDS
// crossbar, no lgkmcnt wait): quad broadcast, then the mirrored half for the other quad.
template <int J> __device__ __forceinline__ uint32_t bcast8(uint32_t v) {
    constexpr int q = J & 3;
    constexpr int quad = q | (q << 2) | (q << 4) | (q << 6);  // quad_perm:[q,q,q,q]
    // t: lanes 0-3 of a group hold v[q], lanes 4-7 hold v[4+q]
    // (every lane is written: no "old" value to preserve, so no register initialisation before the move)
    const uint32_t t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, quad, 0xF, 0xF, false);
    // row_half_mirror (0x141): lane i <- lane 7-i inside each group of 8
    if constexpr (J < 4) {
        // want v[q] everywhere: lanes 4-7 (banks 1,3) take the mirror of lanes 0-3
        return (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)t, 0x141, 0xF, 0xA, false);
    } else {
        // want v[4+q] everywhere: lanes 0-3 (banks 0,2) take the mirror of lanes 4-7
        return (uint32_t)__builtin_amdgcn_update_dpp((int)t, (int)t, 0x141, 0xF, 0x5, false);
    }
}
template <typename T, int J> __device__ __forceinline__ T bcast8_t(T v) {
    if constexpr (sizeof(T) == 8) {
        union { T t; uint32_t u[2]; } in, out;
        in.t = v;
        out.u[0] = bcast8<J>(in.u[0]);
        out.u[1] = bcast8<J>(in.u[1]);
        return out.t;
    } else if constexpr (sizeof(T) == 4) {
        union { T t; uint32_t u; } in, out;
        in.t = v;
        out.u = bcast8<J>(in.u);
        return out.t;
    } else {
        return (T)(int32_t)bcast8<J>((uint32_t)(int32_t)v);
    }
}

// ---------------------------------------------------------------------------
// Row accumulator of the panel sweep.  Generic form: one wide register per element (AccOf<T>).
// Packed form (8- and 16-bit integers with unit weights): sums stay packed in 32-bit words and wrap per
// element exactly as val_dt arithmetic does -- 16-bit lanes through v_pk_add_u16, bytes through a SWAR
// add -- instead of 16 unpack+add pairs per gathered 16-byte piece (int8 h=256: 4.3 -> 2.x ms).
// ---------------------------------------------------------------------------
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4_u __attribute__((ext_vector_type(4), aligned(2)));  // 4 panel-local column ids, any 2-byte alignment
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));  // one gathered 16-byte piece, untyped
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// scale of the conv layers' symmetric quantiser (models/quantize.py:22-30): abs_max * 2 / 2^k
__device__ __forceinline__ float quant_scale(uint32_t absmax_bits, int log2_range) {
    return __uint_as_float(absmax_bits) * 2.0f / (float)(1u << log2_range);
}
// the same piece at ANY byte alignment (rows of X / C whose stride is not a multiple of 16 bytes: h = 41, 100 int8 ...):
// still one global_load/store_dwordx4, the memory path takes unaligned addresses
typedef uint32_t u32x4_b __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ uint32_t add_packed_u16(uint32_t a, uint32_t b) {
    union { uint32_t u; u16x2_t v; } x, y;
    x.u = a;
    y.u = b;
    x.v = x.v + y.v;
    return x.u;
}
__device__ __forceinline__ uint32_t add_packed_u8(uint32_t a, uint32_t b) {
    return ((a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu)) ^ ((a ^ b) & 0x80808080u);  // no carry across bytes
}

template <typename T, int VEC, bool PACKED> struct RowAcc;

template <typename T, int VEC> struct RowAcc<T, VEC, false> {
    using A = typename AccOf<T>::type;
    A a[VEC];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < VEC; k++) a[k] = A(0);
    }
    __device__ __forceinline__ void set(int k, T v) { a[k] = to_acc<T>(v); }
    __device__ __forceinline__ T get(int k) const { return from_acc<T>(a[k]); }
    using V = typename VecOf<T, VEC>::type;
    __device__ __forceinline__ void set_raw(u32x4_t r) {
        const V x = __builtin_bit_cast(V, r);
#pragma unroll
        for (int k = 0; k < VEC; k++) a[k] = to_acc<T>(x[k]);
    }
    __device__ __forceinline__ u32x4_t get_raw() const {
        V x;
#pragma unroll
        for (int k = 0; k < VEC; k++) x[k] = from_acc<T>(a[k]);
        return __builtin_bit_cast(u32x4_t, x);
    }
    __device__ __forceinline__ void add(u32x4_t r) {
        const V x = __builtin_bit_cast(V, r);
#pragma unroll
        for (int k = 0; k < VEC; k++) a[k] += to_acc<T>(x[k]);
    }
    __device__ __forceinline__ void fma(T v, u32x4_t r) {
        const V x = __builtin_bit_cast(V, r);
        const A av = to_acc<T>(v);
#pragma unroll
        for (int k = 0; k < VEC; k++) a[k] += av * to_acc<T>(x[k]);
    }
    __device__ __forceinline__ void merge_xor(int off) {
#pragma unroll
        for (int k = 0; k < VEC; k++) a[k] += shfl_xor_t<A>(a[k], off);
    }
};

template <typename T, int VEC> struct RowAcc<T, VEC, true> {
    static_assert(sizeof(T) < 4 && VEC * sizeof(T) == 16, "packed accumulation is for 8/16-bit elements");
    static constexpr int W = VEC * (int)sizeof(T) / 4;
    union { uint32_t w[W]; T e[VEC]; } u;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < W; k++) u.w[k] = 0u;
    }
    __device__ __forceinline__ void set(int k, T v) { u.e[k] = v; }
    __device__ __forceinline__ T get(int k) const { return u.e[k]; }
    __device__ __forceinline__ static uint32_t padd(uint32_t a, uint32_t b) {
        if constexpr (sizeof(T) == 2) return add_packed_u16(a, b);
        else return add_packed_u8(a, b);
    }
    __device__ __forceinline__ void set_raw(u32x4_t r) {
#pragma unroll
        for (int k = 0; k < W; k++) u.w[k] = r[k];
    }
    __device__ __forceinline__ u32x4_t get_raw() const {
        u32x4_t r;
#pragma unroll
        for (int k = 0; k < W; k++) r[k] = u.w[k];
        return r;
    }
    __device__ __forceinline__ void add(u32x4_t r) {
#pragma unroll
        for (int k = 0; k < W; k++) u.w[k] = padd(u.w[k], r[k]);
    }
    __device__ __forceinline__ void fma(T, u32x4_t) {}  // never instantiated with weights
    __device__ __forceinline__ void merge_xor(int off) {
#pragma unroll
        for (int k = 0; k < W; k++) u.w[k] = padd(u.w[k], (uint32_t)__shfl_xor((int)u.w[k], off));
    }
};

// One gathered 16-byte piece of row c.  The base is wave-uniform (X + slice offset: an SGPR pair), the
// per-lane part is a byte offset, so the load is `global_load_dwordx4 v, v_off, s[base]` and the address
// costs one VALU instruction:
//   AMODE 2: rows of exactly 128 bytes below 4 GiB (the slice-major copy): off = (c << 7) + lane_off
//   AMODE 3: as 2, and the column ids come from the 16-bit panel-local array (see load_chunk)
//   AMODE 1: any row stride below 4 GiB: off = c * row_bytes + lane_off (32-bit multiply)
//   AMODE 0: 64-bit offsets
template <int AMODE>
__device__ __forceinline__ u32x4_t gather_raw(const char *__restrict__ xbase, uint32_t lane_off, uint32_t c,
                                              int64_t row_bytes64, uint32_t row_bytes) {
    if constexpr (AMODE >= 2) return *reinterpret_cast<const u32x4_t *>(xbase + ((c << 7) + lane_off));
    else if constexpr (AMODE == 1) return *reinterpret_cast<const u32x4_b *>(xbase + (c * row_bytes + lane_off));
    else return *reinterpret_cast<const u32x4_b *>(xbase + ((int64_t)c * row_bytes64 + lane_off));
}

// One sweep body, two modes.
//   COOP = false: an 8-lane group owns one work item (row x panel) and walks it alone; a wave carries 8 items.
//   COOP = true : the 8 groups of a wave share ONE long item: group g takes the 32-entry chunks g, g+8, ...
//                 and the partial sums are added across the groups at the end (integers: exact; floats:
//                 a different summation order, inside the 1e-5 bound).  Keeps long rows inside the L2-blocked
//                 sweep instead of leaving one lane group with a serial chain of thousands of gathers.
//   DEQ = true (the conv layers' fused quantise -> aggregate -> dequantise, models/quantize.py:20-42): an item that is
//   the LAST of its row (bit 30 of its length word: no later panel holds entries of the row) stores
//   float(sum) * scale to the float matrix Cf instead of the running sum to C.
// Stored values of the sweep: HAS_VALS 0 = none (all ones), 1 = values of the element type, 2 = NARROW values -- an 8-byte
// element type whose values are all exactly representable in 4 bytes (double <- float, int64 <- int32: checked at group
// creation, k_narrow_vals) streams half the value bytes per slice; the widening conversion is exact, so results do not change.
template <typename T> struct NarrowOf { typedef T type; };
template <> struct NarrowOf<double> { typedef float type; };
template <> struct NarrowOf<int64_t> { typedef int32_t type; };
template <typename T, int HV> struct SweepVal { typedef T type; };
template <typename T> struct SweepVal<T, 2> { typedef typename NarrowOf<T>::type type; };

template <typename T, int VEC, int AMODE, int HAS_VALS, bool COOP, bool DEQ>
__device__ __forceinline__ void panel_sweep(uint32_t blk, const uint32_t *__restrict__ item_row,
                                            const uint32_t *__restrict__ item_begin,
                                            const uint32_t *__restrict__ item_len, uint32_t nitems,
                                            const uint32_t *__restrict__ colind, const T *__restrict__ vals,
                                            const T *__restrict__ X, int64_t ldx, int64_t slice_stride,
                                            T *__restrict__ C, int64_t ldc, uint32_t w, uint32_t nslices,
                                            int accumulate, uint32_t col_base, float *__restrict__ Cf, int64_t ldcf,
                                            const uint32_t *__restrict__ absmax_bits, int log2_range,
                                            const float *__restrict__ post_mul, const float *__restrict__ post_add,
                                            int post_relu) {
    static_assert(VEC * sizeof(T) == 16, "the sweep gathers 16-byte pieces");
    constexpr bool PACKED = !HAS_VALS && sizeof(T) < 4;
    constexpr int LPR = 8;   // lanes per 128-byte slice of a row
    constexpr int G = 8;     // lane groups per wave
    const int lane = threadIdx.x & 63;
    const int li = lane & (LPR - 1);
    const uint32_t grp = (uint32_t)lane >> 3;
    // Block -> (feature slice, item block), XCD-aware for ANY slice count.  Blocks b and b + 8 share an
    // XCD (round-robin dispatch; speed only, never correctness).  The item blocks of every slice are dealt
    // into 8 interleaved strands (rb = j, j + 8, ...: each strand sees the whole length spectrum); the
    // 8 * nslices strands are taken slice-major, nslices per XCD, and an XCD walks its strands one after
    // the other.  So an XCD's L2 holds one slice panel at a time, whatever nslices is (8 slices: XCD x =
    // slice x; 4 slices: two XCDs share a slice; 16 slices: an XCD does two slices in turn).
    const uint32_t nwaves = blockDim.x >> 6;  // waves per block (launch parameter)
    const uint32_t items_per_block = COOP ? nwaves : nwaves * G;
    const uint32_t item_blocks = (nitems + items_per_block - 1) / items_per_block;
    const uint32_t strand_len = (item_blocks + 7) >> 3;
    const uint32_t xcd = blk & 7u, kseq = blk >> 3;
    const uint32_t strand = xcd * nslices + kseq / strand_len;
    const uint32_t slice = strand >> 3;
    // Round 5, items in LOCALITY order (bit 1 of `accumulate`; never for the wave-cooperative prefix): a strand is a CONTIGUOUS eighth of the
    // item blocks, so that the ~8 000 items an XCD has in flight are neighbours in the list -- a few communities whose rows of X (261 KB per
    // slice each, products shape) stay in its L2 -- instead of every eighth block of a 64 K-item window (32 communities, 8 MB: L2 hit rate 43 %)
    const bool contiguous = !COOP && (accumulate & 2);
    const uint32_t rb = contiguous ? (strand & 7u) * strand_len + (kseq % strand_len) : (strand & 7u) + 8u * (kseq % strand_len);
    if (rb >= item_blocks) return;
    const uint32_t wv = threadIdx.x >> 6;
    const uint64_t i64 = COOP ? ((uint64_t)rb * nwaves + wv) : (((uint64_t)rb * nwaves + wv) * G + grp);
    const bool row_ok = i64 < nitems;
    const uint32_t i = row_ok ? (uint32_t)i64 : 0u;
    // work item = (row, first entry, length | FIRST flag): the part of one row that falls into
    // this launch's column panel; FIRST = no earlier panel holds entries of the row
    uint32_t s = 0, len = 0, row = 0;
    bool load_c = false, last = false;
    if (row_ok) {
        row = item_row[i];
        s = item_begin[i];
        const uint32_t lf = item_len[i];
        len = lf & 0x3FFFFFFFu;
        load_c = (accumulate & 1) || !(lf >> 31);
        last = (lf >> 30) & 1u;
    }
    const uint32_t f0 = slice * (LPR * VEC) + li * VEC;
    // (an empty row under accumulate has nothing to add: its row of C is neither read nor written)
    const bool lane_on = row_ok && f0 < w && !(len == 0 && (accumulate & 1));
    // X is either the caller's row-major matrix (slice_stride = slice width) or the slice-major
    // copy made by k_slice_pack (slice_stride = rows * slice width, ldx = slice width);
    // lanes past the width re-read the first piece of their own slice (same cache line as lane 0)
    const char *xbase = reinterpret_cast<const char *>(X + (int64_t)slice * slice_stride);  // wave-uniform
    const uint32_t lane_off = f0 < w ? (uint32_t)li * 16u : 0u;
    const int64_t row_bytes64 = ldx * (int64_t)sizeof(T);
    const uint32_t row_bytes = (uint32_t)row_bytes64;
    T *crow = C + (int64_t)row * ldc;
    RowAcc<T, VEC, PACKED> acc;
    acc.zero();
    if (load_c && lane_on && (!COOP || grp == 0)) {
        if (f0 + VEC <= w) {
            acc.set_raw(__builtin_nontemporal_load(reinterpret_cast<const u32x4_b *>(crow + f0)));
        } else {
#pragma unroll
            for (int k = 0; k < VEC; k++)
                if (f0 + k < w) acc.set(k, crow[f0 + k]);
        }
    }
    // chunk geometry: this group's t-th chunk of 32 entries starts at first + t * stride
    constexpr uint32_t CH = 4 * LPR;
    const uint32_t first = COOP ? CH * grp : 0u;
    constexpr uint32_t STRIDE = COOP ? CH * G : CH;
    uint32_t maxlen = len, minlen = len;
    if constexpr (!COOP) {
#pragma unroll
        for (int off = 32; off >= LPR; off >>= 1) {
            maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, off));
            minlen = min(minlen, (uint32_t)__shfl_xor((int)minlen, off));
        }
    }
    maxlen = rfl(maxlen);  // COOP: every lane of the wave holds the same item
    minlen = rfl(minlen);
    const uint32_t nsteps = (maxlen + STRIDE - 1) / STRIDE;

    // Column ids are fetched 32 per lane group at a time (lane li holds ids 4*li .. 4*li+3 of the
    // chunk: one 128-byte request per group instead of four 32-byte ones), then consumed in four
    // batches of 8 gathers.  The next chunk is requested before the current one is gathered.
    uint32_t c4[4], c4n[4];
    using VT = typename SweepVal<T, HAS_VALS>::type;
    const VT *valsv = reinterpret_cast<const VT *>(vals);
    VT v4[4], v4n[4];
    auto load_chunk = [&](uint32_t cbase, uint32_t (&cc)[4], VT (&vv)[4]) {
        const uint32_t base = cbase + 4u * (uint32_t)li;
        if constexpr (AMODE == 3) {
            // 16-bit ids relative to the first column of this launch's panel (same entry order as colind):
            // half the index bytes through L2 per slice; entries past the end read column col_base (valid)
            const unsigned short *c16 = reinterpret_cast<const unsigned short *>(colind);
            if (base + 4u <= len) {
                const u16x4_u q = __builtin_nontemporal_load(reinterpret_cast<const u16x4_u *>(c16 + s + base));
                cc[0] = col_base + q[0]; cc[1] = col_base + q[1]; cc[2] = col_base + q[2]; cc[3] = col_base + q[3];
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    cc[k] = col_base + ((base + k < len) ? (uint32_t)__builtin_nontemporal_load(c16 + s + base + k) : 0u);
            }
        } else if (base + 4u <= len) {
            const u32x4_u q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_u *>(colind + s + base));
            cc[0] = q[0]; cc[1] = q[1]; cc[2] = q[2]; cc[3] = q[3];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) cc[k] = (base + k < len) ? __builtin_nontemporal_load(colind + s + base + k) : 0u;
        }
        if constexpr (HAS_VALS) {
            if (base + 4u <= len) {
                using V4 = typename Vec4U<VT>::type;
                const V4 q = __builtin_nontemporal_load(reinterpret_cast<const V4 *>(valsv + s + base));
                vv[0] = q[0]; vv[1] = q[1]; vv[2] = q[2]; vv[3] = q[3];
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) vv[k] = (base + k < len) ? __builtin_nontemporal_load(valsv + s + base + k) : VT(0);
            }
        }
    };
    if (nsteps > 0) load_chunk(first, c4n, v4n);
    for (uint32_t t = 0; t < nsteps; t++) {
        const uint32_t cbase = first + t * STRIDE;  // this group's chunk
        const uint32_t wbase = t * STRIDE;          // lowest chunk start in the wave
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c4[k] = c4n[k];
            if constexpr (HAS_VALS) v4[k] = v4n[k];
        }
        if (t + 1 < nsteps) load_chunk(cbase + STRIDE, c4n, v4n);
        // entries past a row's end carry column 0 (a valid row of X): gathered, then masked out
#define PYGIM_PANEL_BATCH(B)                                                                               \
        if (wbase + 8u * B < maxlen) {                                                                     \
            const uint32_t eb = cbase + 8u * B;                                                            \
            /* every group of the wave has all 8 entries of this batch? */                                 \
            const bool full = COOP ? (wbase + (STRIDE - CH) + 8u * B + 8u <= maxlen) : (eb + 8u <= minlen); \
            uint32_t cj[LPR];                                                                              \
            cj[0] = bcast8<2 * B>(c4[0]); cj[1] = bcast8<2 * B>(c4[1]);                                    \
            cj[2] = bcast8<2 * B>(c4[2]); cj[3] = bcast8<2 * B>(c4[3]);                                    \
            cj[4] = bcast8<2 * B + 1>(c4[0]); cj[5] = bcast8<2 * B + 1>(c4[1]);                            \
            cj[6] = bcast8<2 * B + 1>(c4[2]); cj[7] = bcast8<2 * B + 1>(c4[3]);                            \
            u32x4_t x[LPR];                                                                                \
            _Pragma("unroll") for (int j = 0; j < LPR; j++)                                                \
                x[j] = gather_raw<AMODE>(xbase, lane_off, cj[j], row_bytes64, row_bytes);                  \
            if constexpr (HAS_VALS) {                                                                      \
                T vj[LPR];                                                                                 \
                vj[0] = (T)bcast8_t<VT, 2 * B>(v4[0]); vj[1] = (T)bcast8_t<VT, 2 * B>(v4[1]);              \
                vj[2] = (T)bcast8_t<VT, 2 * B>(v4[2]); vj[3] = (T)bcast8_t<VT, 2 * B>(v4[3]);              \
                vj[4] = (T)bcast8_t<VT, 2 * B + 1>(v4[0]); vj[5] = (T)bcast8_t<VT, 2 * B + 1>(v4[1]);      \
                vj[6] = (T)bcast8_t<VT, 2 * B + 1>(v4[2]); vj[7] = (T)bcast8_t<VT, 2 * B + 1>(v4[3]);      \
                _Pragma("unroll") for (int j = 0; j < LPR; j++) {                                          \
                    if (full || eb + j < len) acc.fma(vj[j], x[j]);                                       \
                }                                                                                          \
            } else if (full) {                                                                             \
                _Pragma("unroll") for (int j = 0; j < LPR; j++) acc.add(x[j]);                              \
            } else {                                                                                       \
                _Pragma("unroll") for (int j = 0; j < LPR; j++) {                                          \
                    if (eb + j < len) acc.add(x[j]);                                                       \
                }                                                                                          \
            }                                                                                              \
        }
        PYGIM_PANEL_BATCH(0)
        PYGIM_PANEL_BATCH(1)
        PYGIM_PANEL_BATCH(2)
        PYGIM_PANEL_BATCH(3)
#undef PYGIM_PANEL_BATCH
    }

    if constexpr (COOP) {
        // add the 8 groups' partial sums (same feature lanes, lane ^ 8, ^ 16, ^ 32)
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) acc.merge_xor(off);
    }
    if (lane_on && (!COOP || grp == 0)) {
        if constexpr (DEQ) {
            if (last) {
                // scale_edge (1.) * scale_x, as k_dequantize; then the caller's per-column epilogue, if any:
                // y = post_mul[f] * y + post_add[f], optionally max(y, 0)  (bias + eval-mode BatchNorm + ReLU of a GCN layer)
                const float scale = 1.0f * quant_scale(*absmax_bits, log2_range);
                float *frow = Cf + (int64_t)row * ldcf + f0;
                float y[VEC];
#pragma unroll
                for (int k = 0; k < VEC; k++) y[k] = (float)acc.get(k) * scale;
                if (post_mul) {
#pragma unroll
                    for (int k = 0; k < VEC; k++)
                        if (f0 + k < w) {
                            y[k] = post_mul[f0 + k] * y[k] + post_add[f0 + k];
                            if (post_relu) y[k] = fmaxf(y[k], 0.0f);
                        }
                }
                if (f0 + VEC <= w && ((reinterpret_cast<uintptr_t>(frow) & 15u) == 0)) {
#pragma unroll
                    for (int k = 0; k < VEC; k += 4) {
                        f32x4_t o;
                        o.x = y[k];
                        o.y = y[k + 1];
                        o.z = y[k + 2];
                        o.w = y[k + 3];
                        if constexpr (VEC == 4) __builtin_nontemporal_store(o, reinterpret_cast<f32x4_t *>(frow + k));
                        else *reinterpret_cast<f32x4_t *>(frow + k) = o;  // lines completed by several stores: let L2 merge them
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; k++)
                        if (f0 + k < w) frow[k] = y[k];
                }
                return;
            }
        }
        if (f0 + VEC <= w) {
            __builtin_nontemporal_store((u32x4_b)acc.get_raw(), reinterpret_cast<u32x4_b *>(crow + f0));
        } else {
#pragma unroll
            for (int k = 0; k < VEC; k++)
                if (f0 + k < w) crow[f0 + k] = acc.get(k);
        }
    }
}

// column-split LDS plans: C[r][f] = (accumulate ? C[r][f] : 0) + part[0][r][f] + part[1][r][f] + ... in range order (deterministic;
// integers exact, floats: the sum of the column ranges' sequential sums)
template <typename T>
__global__ void k_lds_reduce(const T *__restrict__ part, uint32_t splits, uint64_t nrows, uint32_t w, uint64_t ldp, T *__restrict__ C,
                             int64_t ldc, int accumulate) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * w) return;
    const uint64_t r = i / w;
    const uint32_t f = (uint32_t)(i % w);
    using A = typename AccOf<T>::type;
    A acc = (A)part[r * ldp + f];
    for (uint32_t c = 1; c < splits; c++) acc = (A)(acc + (A)part[((uint64_t)c * nrows + r) * ldp + f]);
    if (accumulate) acc = (A)((A)C[r * ldc + f] + acc);   // (as the kernels' own store: what was there + the product)
    C[r * ldc + f] = (T)acc;
}

// ... with the conv layers' dequantisation (and the optional per-column epilogue) where the ranges' sums meet (round 6: a rank's row share is a column-split
// plan, and its quantised aggregation -- models/pyg_gcn_conv.py:130-137 -- takes the LDS-staged kernel too): out = float(sum) * scale, as the kernels' own
// dequantising store and k_dequantize
template <typename T>
__global__ void k_lds_reduce_deq(const T *__restrict__ part, uint32_t splits, uint64_t nrows, uint32_t w, uint64_t ldp, float *__restrict__ out, int64_t ldo,
                                 const uint32_t *__restrict__ absmax_bits, int log2_range, const float *__restrict__ post_mul, const float *__restrict__ post_add,
                                 int post_relu) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * w) return;
    const uint64_t r = i / w;
    const uint32_t f = (uint32_t)(i % w);
    using A = typename AccOf<T>::type;
    A acc = (A)part[r * ldp + f];
    for (uint32_t c = 1; c < splits; c++) acc = (A)(acc + (A)part[((uint64_t)c * nrows + r) * ldp + f]);
    float y = (float)(T)acc * (1.0f * quant_scale(*absmax_bits, log2_range));
    if (post_mul) {
        y = post_mul[f] * y + post_add[f];
        if (post_relu) y = fmaxf(y, 0.0f);
    }
    out[r * ldo + f] = y;
}

// the same two kernels, four features to a thread (16-byte loads and stores), for widths and strides that are multiples of four elements: 4-byte types
template <typename T, bool DEQ>
__global__ void k_lds_reduce_v4(const T *__restrict__ part, uint32_t splits, uint64_t nrows, uint32_t w, uint64_t ldp, void *__restrict__ C, int64_t ldc, int accumulate,
                                const uint32_t *__restrict__ absmax_bits, int log2_range, const float *__restrict__ post_mul, const float *__restrict__ post_add, int post_relu) {
    static_assert(sizeof(T) == 4, "4-byte element types");
    using A = typename AccOf<T>::type;
    const uint32_t w4 = w >> 2;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * w4) return;
    const uint64_t r = i / w4;
    const uint32_t f = (uint32_t)(i % w4) * 4;
    T v[4];
    load_vec<T, 4>(part + r * ldp + f, v);
    A acc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) acc[k] = to_acc<T>(v[k]);
    for (uint32_t c = 1; c < splits; c++) {
        load_vec<T, 4>(part + ((uint64_t)c * nrows + r) * ldp + f, v);
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = (A)(acc[k] + to_acc<T>(v[k]));
    }
    if constexpr (DEQ) {
        const float scale = 1.0f * quant_scale(*absmax_bits, log2_range);
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            y[k] = (float)from_acc<T>(acc[k]) * scale;
            if (post_mul) {
                y[k] = post_mul[f + k] * y[k] + post_add[f + k];
                if (post_relu) y[k] = fmaxf(y[k], 0.0f);
            }
        }
        store_vec<float, 4>((float *)C + r * ldc + f, y);
    } else {
        T *c = (T *)C + r * ldc + f;
        if (accumulate) {
            load_vec<T, 4>(c, v);
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = (A)(to_acc<T>(v[k]) + acc[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = from_acc<T>(acc[k]);
        store_vec<T, 4>(c, v);
    }
}

// ... and for the 16-bit streams (INT16, and INT8 on its widened features): the ranges' partial sums are 16-bit numbers, two to a dword; their sum wraps modulo 2^16,
// whose low byte is the modular int8 sum.  MODE 0: INT8 result (the low byte; accumulate adds what was there), 1: dequantise the int8 sum, 2: dequantise the int16 sum
template <int MODE>
__global__ void k_lds_reduce16(const int16_t *__restrict__ part, uint32_t splits, uint64_t nrows, uint32_t w, uint64_t ldp, void *__restrict__ C, int64_t ldc, int accumulate,
                               const uint32_t *__restrict__ absmax_bits, int log2_range, const float *__restrict__ post_mul, const float *__restrict__ post_add, int post_relu) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * w) return;
    const uint64_t r = i / w;
    const uint32_t f = (uint32_t)(i % w);
    uint32_t acc = (uint32_t)(int32_t)part[r * ldp + f];
    for (uint32_t c = 1; c < splits; c++) acc += (uint32_t)(int32_t)part[((uint64_t)c * nrows + r) * ldp + f];
    if constexpr (MODE == 0) {
        int8_t *c8 = (int8_t *)C + (int64_t)r * ldc + f;
        if (accumulate) acc += (uint32_t)(int32_t)*c8;
        *c8 = (int8_t)acc;
    } else {
        float y = (float)(MODE == 1 ? (int32_t)(int8_t)acc : (int32_t)(int16_t)acc) * (1.0f * quant_scale(*absmax_bits, log2_range));
        if (post_mul) {
            y = post_mul[f] * y + post_add[f];
            if (post_relu) y = fmaxf(y, 0.0f);
        }
        ((float *)C)[(int64_t)r * ldc + f] = y;
    }
}

// The last rows of a row share, outside the LDS-staged plan (round 6): when tall x slices x S full-height row tiles cover all but a few percent of a share's
// rows with EXACTLY one workgroup per compute unit (a 1/8 share of the Reddit-shaped graph: 16 tiles x 4 slices x 4 ranges = 256 workgroups for 29 184 of its
// 29 471 rows, against 17 x 4 x 3 = 204 for all of them), the plan takes those tiles and these two kernels the remainder.  They gather from the SAME slice-major
// copy the LDS-staged kernel streams (256-byte row slices).  The rows are cut into SEGMENTS of at most LDS_TAIL_SEG entries at plan time (a row of 20 000 entries
// on one workgroup would be a latency-bound chain of its own: the first form of this kernel, a workgroup per row, cost more than the extra column range saved):
//   k_lds_tail_seg  a 512-thread workgroup per (segment, slice): wave v takes entries v, v + 8, ... of the segment (at most 32), all their gathers in flight together; the eight
//                   partial sums are added in wave order and parked in scratch
//   k_lds_tail_fin  a wave per (row, slice): the row's segments in order, then the store -- plain, accumulating, or with the conv layers' dequantisation + epilogue
// Deterministic; integers exact, FLT32 in this fixed order (these shares are column-split plans: the norm-wise contract already).
constexpr uint32_t LDS_TAIL_SEG = 256;
template <typename T>
__global__ __launch_bounds__(512) void k_lds_tail_seg(const uint32_t *__restrict__ segs, uint32_t nseg, const uint32_t *__restrict__ colind, const char *__restrict__ xs,
                                                      uint64_t slice_stride, typename AccOf<T>::type *__restrict__ parked) {
    static_assert(sizeof(T) == 4, "4-byte element types");
    using A = typename AccOf<T>::type;
    __shared__ A partial[8][64];
    const uint32_t seg = blockIdx.x, slice = blockIdx.y;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t e0 = segs[nseg + seg], e1 = segs[2 * nseg + seg];
    const char *xsl = xs + (uint64_t)slice * slice_stride + lane * 4;
    // a segment holds at most LDS_TAIL_SEG = 256 entries: at most 32 for this wave -- all their gathers go out together (one round trip), the sums follow in entry order
    static_assert(LDS_TAIL_SEG == 256, "32 entries per wave");
    uint32_t cols[32];
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const uint32_t e = e0 + wave + 8u * k;
        cols[k] = e < e1 ? colind[e] : 0xFFFFFFFFu;
    }
    T x[32];
#pragma unroll
    for (int k = 0; k < 32; k++) x[k] = cols[k] != 0xFFFFFFFFu ? *(const T *)(xsl + (uint64_t)cols[k] * 256) : (T)0;
    A acc = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) acc = (A)(acc + to_acc<T>(x[k]));   // (entries beyond the segment add an exact zero)
    partial[wave][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    A sum = partial[0][lane];
#pragma unroll
    for (int v = 1; v < 8; v++) sum = (A)(sum + partial[v][lane]);
    parked[((uint64_t)seg * gridDim.y + slice) * 64 + lane] = sum;
}
template <typename T, bool DEQ>
__global__ __launch_bounds__(64) void k_lds_tail_fin(const uint32_t *__restrict__ rowseg, uint32_t row0, const typename AccOf<T>::type *__restrict__ parked, uint32_t w_lanes,
                                                     void *__restrict__ C, int64_t ldc_bytes, int accumulate, const uint32_t *__restrict__ absmax_bits, int log2_range,
                                                     const float *__restrict__ post_mul, const float *__restrict__ post_add, int post_relu) {
    using A = typename AccOf<T>::type;
    const uint32_t t = blockIdx.x, slice = blockIdx.y, lane = threadIdx.x;
    const uint32_t f = slice * 64 + lane;
    if (f >= w_lanes) return;
    A sum = 0;
    for (uint32_t s = rowseg[t]; s < rowseg[t + 1]; s++) sum = (A)(sum + parked[((uint64_t)s * gridDim.y + slice) * 64 + lane]);
    const uint64_t row = (uint64_t)row0 + t;
    if constexpr (DEQ) {
        float y = (float)from_acc<T>(sum) * (1.0f * quant_scale(*absmax_bits, log2_range));
        if (post_mul) {
            y = post_mul[f] * y + post_add[f];
            if (post_relu) y = fmaxf(y, 0.0f);
        }
        *(float *)((char *)C + (int64_t)row * ldc_bytes + (size_t)f * 4) = y;
    } else {
        T *c = (T *)((char *)C + (int64_t)row * ldc_bytes + (size_t)f * 4);
        if (accumulate) sum = (A)(to_acc<T>(*c) + sum);
        *c = from_acc<T>(sum);
    }
}

// dst[i] = src[i], 16 bytes a thread (the upload of the code stream into its executable allocation)
__global__ void k_copy16(const u32x4_t *__restrict__ src, u32x4_t *__restrict__ dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// narrow[i] = vals[i] in the 4-byte type; *flag |= 1 when some value does not survive the round trip (NaNs included)
template <typename T>
__global__ void k_narrow_vals(const T *__restrict__ vals, uint64_t n, typename NarrowOf<T>::type *__restrict__ narrow, int *flag) {
    using NT = typename NarrowOf<T>::type;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T v = vals[i];
    const NT q = (NT)v;
    bool same;
    if constexpr (std::is_same<T, double>::value) same = __double_as_longlong((double)q) == __double_as_longlong(v);
    else same = (T)q == v;
    narrow[i] = q;
    if (!same) atomicOr(flag, 1);
}

// wide[i] = vals[i] sign-extended to 32 bits (valued INT8 / INT16 on the code stream: the encoders read 4-byte value slots)
template <typename T>
__global__ void k_widen_vals(const T *__restrict__ vals, uint64_t n, uint32_t *__restrict__ wide) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wide[i] = (uint32_t)(int32_t)vals[i];
}

// One launch per column panel: the first coop_grid blocks take the panel's LONG items (one wave each,
// dispatched first), the rest the ordinary items (one lane group each).
template <typename T, int VEC, int LOG_LPR, int AMODE, int HAS_VALS, bool DEQ = false>
__global__ __launch_bounds__(256) void k_csr_panel(const uint32_t *__restrict__ item_row,
                                                   const uint32_t *__restrict__ item_begin,
                                                   const uint32_t *__restrict__ item_len, uint32_t nitems,
                                                   const uint32_t *__restrict__ coop_row,
                                                   const uint32_t *__restrict__ coop_begin,
                                                   const uint32_t *__restrict__ coop_len, uint32_t ncoop,
                                                   uint32_t coop_grid, const uint32_t *__restrict__ colind,
                                                   const T *__restrict__ vals, const T *__restrict__ X,
                                                   int64_t ldx, int64_t slice_stride, T *__restrict__ C,
                                                   int64_t ldc, uint32_t w, uint32_t nslices, int accumulate,
                                                   uint32_t col_base, float *__restrict__ Cf, int64_t ldcf,
                                                   const uint32_t *__restrict__ absmax_bits, int log2_range,
                                                   const float *__restrict__ post_mul, const float *__restrict__ post_add,
                                                   int post_relu) {
    static_assert(LOG_LPR == 3, "lane groups of 8 (one 128-byte line per gathered row slice)");
    if (blockIdx.x < coop_grid)
        panel_sweep<T, VEC, AMODE, HAS_VALS, true, DEQ>(blockIdx.x, coop_row, coop_begin, coop_len, ncoop, colind, vals, X,
                                                        ldx, slice_stride, C, ldc, w, nslices, accumulate, col_base, Cf, ldcf,
                                                        absmax_bits, log2_range, post_mul, post_add, post_relu);
    else
        panel_sweep<T, VEC, AMODE, HAS_VALS, false, DEQ>(blockIdx.x - coop_grid, item_row, item_begin, item_len, nitems,
                                                         colind, vals, X, ldx, slice_stride, C, ldc, w, nslices, accumulate,
                                                         col_base, Cf, ldcf, absmax_bits, log2_range, post_mul, post_add, post_relu);
}

// Slice-major copy of X for the panel sweep: Xs[s][j][0:F] = X[j][s*F : (s+1)*F] (zero padded past
// the width).  Why: with row-major X the 128-byte slice an XCD gathers sits at a fixed offset inside
// every row (stride = row bytes, 1 KiB at h = 256 f32); those addresses share their low bits and land
// in a fraction of the L2's sets/channels, so the XCD's panel is evicted long before the L2 is full.
// In the slice-major copy a panel of one slice is one contiguous range.
template <typename T, int VEC, int LOG_LPR>
__global__ void k_slice_pack(const T *__restrict__ X, int64_t ldx, uint32_t nrows, uint32_t w, uint32_t nslices,
                             T *__restrict__ Xs, uint32_t slice_rows,   // slice_rows >= nrows: rows between two slices of the copy
                             const uint32_t *__restrict__ order = nullptr) {  // row j of the copy = X[order[j]] (lds_hybrid_dev.hpp), or X[j]
    constexpr int LPR = 1 << LOG_LPR;
    constexpr uint32_t F = LPR * VEC;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_row = (uint64_t)nslices * LPR;
    const uint64_t j = t / per_row;
    if (j >= nrows) return;
    const uint32_t rem = (uint32_t)(t % per_row);
    const uint32_t sl = rem >> LOG_LPR, li = rem & (LPR - 1);
    const uint32_t f0 = sl * F + li * VEC;
    static_assert(VEC * sizeof(T) == 16, "16-byte pieces");
    T *dst = Xs + ((int64_t)sl * slice_rows + j) * F + li * VEC;
    const int64_t src_row = order ? (int64_t)order[j] : (int64_t)j;
    if (f0 + VEC <= w) {  // X rows may sit at any byte alignment; the copy is 16-byte aligned
        const u32x4_b q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_b *>(X + src_row * ldx + f0));
        *reinterpret_cast<u32x4_t *>(dst) = (u32x4_t)q;
    } else {
        T v[VEC];
#pragma unroll
        for (int k = 0; k < VEC; k++) v[k] = (f0 + k < w) ? X[src_row * ldx + f0 + k] : T(0);
        store_vec<T, VEC>(dst, v);
    }
}

// Half-split staging (round 6, LdsGeometry::half_split): products of at most 32 lanes fold TWO column ranges into the halves of a wave -- row j of the copy is
// [X[j][0 .. 32) | X[j + H][0 .. 32)], 128 bytes each (features beyond the width and rows beyond the matrix: zeros).  One thread per 16-byte piece; 4-byte types.
template <typename T>
__global__ void k_slice_pack_hs(const T *__restrict__ X, int64_t ldx, uint32_t ncols, uint32_t w, uint32_t H, T *__restrict__ Xs) {
    static_assert(sizeof(T) == 4, "4-byte element types");
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j = t >> 4;
    if (j >= H) return;
    const uint32_t pc = (uint32_t)(t & 15);
    const uint64_t src = pc < 8 ? j : j + H;
    const uint32_t f0 = (pc & 7) * 4;
    T v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (src < ncols && f0 + k < w) ? X[(int64_t)src * ldx + f0 + k] : (T)0;
    store_vec<T, 4>(Xs + j * 64 + pc * 4, v);
}

// INT8 features for the code-stream product (round 4): the slice-major copy holds them WIDENED to 16 bits -- a slice is 128 features =
// 256 bytes, as for INT16 -- so that the stream's packed 16-bit adds carry them; the low byte of every 16-bit sum is the int8 sum.
// One thread: 8 features (8 bytes in, 16 bytes out).
__global__ void k_slice_pack_widen8(const int8_t *__restrict__ X, int64_t ldx, uint32_t nrows, uint32_t w, uint32_t nslices,
                                    int16_t *__restrict__ Xs, uint32_t slice_rows, const uint32_t *__restrict__ order = nullptr) {   // (order: as k_slice_pack)
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_row = (uint64_t)nslices * 16;
    const uint64_t j = t / per_row;
    if (j >= nrows) return;
    const uint32_t rem = (uint32_t)(t % per_row);
    const uint32_t sl = rem >> 4, li = rem & 15;
    const uint32_t f0 = sl * 128 + li * 8;
    int16_t v[8];
    const int64_t src_row = order ? (int64_t)order[j] : (int64_t)j;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = (f0 + k < w) ? (int16_t)X[src_row * ldx + f0 + k] : (int16_t)0;
    store_vec<int16_t, 8>(Xs + ((int64_t)sl * slice_rows + j) * 128 + li * 8, v);
}

// The same copy for the fused quantised aggregation: reads the FLOAT features, quantises them as k_quantize does
// (round-half-even of the true quotient v / scale, models/quantize.py:31-33) and writes the slice-major copy in the
// adjacency type -- the row-major quantised matrix is never materialised.
template <typename T, int VEC, int LOG_LPR>
__global__ void k_slice_pack_quant(const float *__restrict__ X, int64_t ldx, uint32_t nrows, uint32_t w, uint32_t nslices,
                                   const uint32_t *__restrict__ absmax_bits, int log2_range, T *__restrict__ Xs,
                                   float *__restrict__ scale_out, uint32_t slice_rows) {  // slice_rows >= nrows, as k_slice_pack
    constexpr int LPR = 1 << LOG_LPR;
    constexpr uint32_t F = LPR * VEC;
    static_assert(VEC * sizeof(T) == 16 && VEC % 4 == 0, "16-byte pieces of at least four elements");
    const float scale = quant_scale(*absmax_bits, log2_range);
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && scale_out) *scale_out = scale;
    const uint64_t per_row = (uint64_t)nslices * LPR;
    const uint64_t j = t / per_row;
    if (j >= nrows) return;
    const uint32_t rem = (uint32_t)(t % per_row);
    const uint32_t sl = rem >> LOG_LPR, li = rem & (LPR - 1);
    const uint32_t f0 = sl * F + li * VEC;
    const float *src = X + (int64_t)j * ldx + f0;
    T v[VEC];
    if (f0 + VEC <= w && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0)) {
#pragma unroll
        for (int k = 0; k < VEC; k += 4) {
            const f32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(src + k));
            v[k] = (T)rintf(q.x / scale);
            v[k + 1] = (T)rintf(q.y / scale);
            v[k + 2] = (T)rintf(q.z / scale);
            v[k + 3] = (T)rintf(q.w / scale);
        }
    } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) v[k] = (f0 + k < w) ? (T)rintf(src[k] / scale) : T(0);
    }
    store_vec<T, VEC>(Xs + ((int64_t)sl * slice_rows + j) * F + li * VEC, v);
}

// panel pointers: pp[p * nrows + i] = first stored entry of sorted row i whose column
// is >= p * panel_cols (p = 0..npanels; pp[0] = row start, pp[npanels] = row end)
__global__ void k_build_panel_ptr(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ rowptr,
                                  const uint32_t *__restrict__ colind, uint32_t nrows, uint32_t npanels,
                                  uint32_t panel_cols, uint32_t *__restrict__ pp) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)nrows * (npanels + 1)) return;
    const uint32_t i = (uint32_t)(t % nrows);
    const uint32_t p = (uint32_t)(t / nrows);
    const uint32_t row = perm ? perm[i] : i;
    uint32_t lo = rowptr[row], hi = rowptr[row + 1];
    if (p == 0) {
        pp[t] = lo;
        return;
    }
    if (p == npanels) {
        pp[t] = hi;
        return;
    }
    const uint64_t bound = (uint64_t)p * panel_cols;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)colind[mid] < bound) lo = mid + 1;
        else hi = mid;
    }
    pp[t] = lo;
}

// ---------------------------------------------------------------------------
// The SpMV end with the dense operand staged in LDS (spmv_sparseP/dpu_kernels/spmv_mul_coo_dpu.c keeps its slice of the
// vector in the DPU's scratchpad the same way).  For rows of X of at most 16 bytes a gather through the cache hierarchy
// costs a whole 128-byte L2 request per stored entry (k_csr_vec: 0.68 ms on the Reddit-shaped graph); here a 1024-thread
// workgroup copies ONE column panel of X (panel_cols rows of W elements, <= 128 KiB) into LDS with coalesced loads and every
// stored entry is one ds_read.  k_spmv_lds: ONE launch for all column panels, no read-modify-write of C, no stores inside
// the software pipeline (0.14 ms; a first form with a launch per panel and the running sum continued through C: 0.38 ms).
//   * a workgroup (one per CU) is given one unit = (panel, slot of nslots): it stages the panel of X once and walks the items
//     slot, slot + nslots, ... of the panel's length-sorted list -- the workgroups of a panel share it like cards dealt in
//     turn, so each sees the whole length spectrum;
//   * the sorted list falls into five length classes (at most 32 / 64 / 128 / 256 entries, and longer) worked by lane groups of
//     4 / 8 / 16 / 32 / 64 lanes, 8 entries per lane, so that an item is ONE pass: the loop over a lane group's items is then
//     perfectly regular and runs as a register pipeline -- descriptors 6 items ahead, the 16 bytes of ids (and the weights)
//     3 items ahead, every load unconditional (indices clamped, never predicated), so the waits are vmcnt(N) for the oldest
//     load only.  Only the whole-wave class has items of several passes (beyond 512 entries: rare), fetched on demand;
//   * gfx9 counts stores in vmcnt and lets them retire out of order with loads: ONE pending store turns every later wait
//     for a load into vmcnt(0) and drains the pipeline (the first form ran 82 % of its wave cycles in such waits).  So a
//     lane group parks (row, sum) in LDS and flushes F of them at a time;
//   * results go to part[panel][row] (each written at most once; the buffer is zeroed before), and k_spmv_reduce adds a
//     row's panels in panel order into C: deterministic, integers exact, floats inside the 1e-5 bound.
// ---------------------------------------------------------------------------
extern __shared__ unsigned char pygim_lds_raw[];
struct SpmvUnit {
    uint32_t item_off;  // the panel's first item in the part's item arrays
    uint32_t n_items;   // its items, longest first
    uint32_t n64, n32, n16, n8;  // the list's leading items by length class: > 256 entries (a wave each, 512 per pass), 129..256
                                 // (32 lanes), 65..128 (16 lanes), 33..64 (8 lanes); the rest, at most 32 entries, take 4 lanes --
                                 // 8 entries per lane in every class
    uint32_t col_base, pcols;
    uint32_t slot, nslots;
    uint32_t panel;
};
typedef unsigned short u16x8_u __attribute__((ext_vector_type(8), aligned(2)));

template <int CTRL, typename A> __device__ __forceinline__ A dpp_get(A v) {
    if constexpr (sizeof(A) == 8) {
        union { A t; int u[2]; } in, out;
        in.t = v;
        out.u[0] = __builtin_amdgcn_update_dpp(0, in.u[0], CTRL, 0xF, 0xF, false);
        out.u[1] = __builtin_amdgcn_update_dpp(0, in.u[1], CTRL, 0xF, 0xF, false);
        return out.t;
    } else {
        union { A t; int u; } in, out;
        in.t = v;
        out.u = __builtin_amdgcn_update_dpp(0, in.u, CTRL, 0xF, 0xF, false);
        return out.t;
    }
}
// sum over the LG lanes of a lane group, in a fixed order, every lane ends with the total
template <int LG, typename A> __device__ __forceinline__ A lanes_sum(A v) {
    static_assert(LG == 4 || LG == 8 || LG == 16 || LG == 32 || LG == 64, "a quad, half a DPP row, a row, two rows or the whole wave");
    v += dpp_get<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);   // quad_perm [2,3,0,1]
    if constexpr (LG >= 8) v += dpp_get<0x141>(v);  // row_half_mirror
    if constexpr (LG >= 16) v += dpp_get<0x140>(v);  // row_mirror
    if constexpr (LG >= 32) v += shfl_xor_t<A>(v, 16);
    if constexpr (LG == 64) v += shfl_xor_t<A>(v, 32);
    return v;
}

template <typename T, int W, bool HAS_VALS, int LG>
__device__ __forceinline__ void spmv_lds_class(const char *__restrict__ ir, const char *__restrict__ ib,
                                               const char *__restrict__ il, uint32_t n, uint32_t first, uint32_t stride,
                                               const char *__restrict__ col16, const T *__restrict__ vals,
                                               const T *__restrict__ xs, uint32_t pcols,
                                               typename AccOf<T>::type *__restrict__ out, uint32_t *st_row,
                                               typename AccOf<T>::type *st_val, uint32_t F, uint32_t li) {
    using A = typename AccOf<T>::type;
    constexpr int D = HAS_VALS ? 2 : 3, R = 6;  // (weights cost 8 more loads and up to 16 registers per item in flight)
    static_assert(R % D == 0 && R >= 2 * D, "descriptor ring: a multiple of the id ring, at least twice as deep");
    constexpr uint32_t PASS = 8u * LG;  // entries per pass
    // this lane group's items are first, first + stride, ...; the wave runs as long as its FIRST group has items
    const uint32_t my_iters = first < n ? (n - first + stride - 1) / stride : 0u;
    const uint32_t iters = rfl(my_iters);  // lane 0 sits in the wave's first group
    if (iters == 0) return;
    uint32_t d_row[R], d_s[R], d_len[R];
    u16x8_u d_q[D];
    T d_v[D][8];
    auto ld_desc = [&](int slot, uint32_t t) {
        const uint32_t ii = min(first + t * stride, n - 1u) << 2;  // clamped: always a real item, never a predicated load
        d_row[slot] = *reinterpret_cast<const uint32_t *>(ir + ii);
        d_s[slot] = *reinterpret_cast<const uint32_t *>(ib + ii);
        d_len[slot] = *reinterpret_cast<const uint32_t *>(il + ii) & 0x3FFFFFFFu;
    };
    auto ld_ids = [&](int qslot, int dslot, uint32_t k0) {  // entries k0 + 8 li .. + 7 of the item in descriptor slot dslot
        const uint32_t len = d_len[dslot], s = d_s[dslot];
        const uint32_t e0 = (k0 + 8u * li < len) ? k0 + 8u * li : 0u;  // lanes past the end re-read the item's first ids
        d_q[qslot] = __builtin_nontemporal_load(reinterpret_cast<const u16x8_u *>(col16 + ((s + e0) << 1)));  // (id array is padded)
        if constexpr (HAS_VALS) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t e = k0 + 8u * li + (uint32_t)u;
                d_v[qslot][u] = __builtin_nontemporal_load(vals + (e < len ? s + e : 0u));  // element 0 stands in: masked below
            }
        }
    };
    auto add_pass = [&](A(&acc)[W], const u16x8_u q, const T(&v)[8], uint32_t k0, uint32_t len) {
        T xv[8][W];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool ok = k0 + 8u * li + (uint32_t)u < len;
            const uint32_t c = ok ? (uint32_t)q[u] : pcols;  // the all-zero row staged behind the panel
#pragma unroll
            for (int j = 0; j < W; j++) xv[u][j] = xs[c * W + j];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (HAS_VALS) {
                const bool ok = k0 + 8u * li + (uint32_t)u < len;
                const A w = ok ? to_acc<T>(v[u]) : A(0);
#pragma unroll
                for (int j = 0; j < W; j++) acc[j] += w * to_acc<T>(xv[u][j]);
            } else {
#pragma unroll
                for (int j = 0; j < W; j++) acc[j] += to_acc<T>(xv[u][j]);
            }
        }
    };
    auto flush = [&](uint32_t staged) {
        __builtin_amdgcn_wave_barrier();
        for (uint32_t f = li; f < staged; f += LG) {
            const uint32_t row = st_row[f];
            if (row != 0xFFFFFFFFu) {
#pragma unroll
                for (int j = 0; j < W; j++) out[(size_t)row * W + j] = st_val[f * W + j];
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll
    for (int t = 0; t < R; t++) ld_desc(t, (uint32_t)t);
#pragma unroll
    for (int t = 0; t < D; t++) ld_ids(t, t, 0u);
    uint32_t staged = 0;
    for (uint32_t t0 = 0; t0 < iters; t0 += R) {
#pragma unroll
        for (int sx = 0; sx < R; sx++) {
            const uint32_t t = t0 + (uint32_t)sx;
            const bool valid = t < my_iters;
            const uint32_t row = d_row[sx], s = d_s[sx];
            const uint32_t len = valid ? d_len[sx] : 0u;
            const u16x8_u q = d_q[sx % D];
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = HAS_VALS ? d_v[sx % D][u] : T(0);
            // keep the pipeline full first: descriptor of item t + R, ids of item t + D
            ld_desc(sx, t + (uint32_t)R);
            ld_ids(sx % D, (sx + D) % R, 0u);
            A acc[W];
#pragma unroll
            for (int j = 0; j < W; j++) acc[j] = A(0);
            add_pass(acc, q, v, 0u, len);
            if constexpr (LG == 64) {  // one item per wave: its length is wave-uniform; passes beyond the first on demand
                const uint32_t ulen = rfl(len);
                for (uint32_t k0 = PASS; k0 < ulen; k0 += PASS) {
                    const uint32_t e0 = (k0 + 8u * li < ulen) ? k0 + 8u * li : 0u;
                    const u16x8_u q2 = __builtin_nontemporal_load(reinterpret_cast<const u16x8_u *>(col16 + ((s + e0) << 1)));
                    T v2[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t e = k0 + 8u * li + (uint32_t)u;
                        v2[u] = HAS_VALS ? __builtin_nontemporal_load(vals + (e < ulen ? s + e : 0u)) : T(0);
                    }
                    add_pass(acc, q2, v2, k0, ulen);
                }
            }
#pragma unroll
            for (int j = 0; j < W; j++) acc[j] = lanes_sum<LG, A>(acc[j]);
            if (li == 0) {
                st_row[staged + sx] = valid ? row : 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < W; j++) st_val[(staged + sx) * W + j] = acc[j];
            }
        }
        staged += R;
        if (staged + R > F || t0 + R >= iters) {
            flush(staged);
            staged = 0;
        }
    }
}

template <typename T, int W, bool HAS_VALS>
__global__ __launch_bounds__(1024) void k_spmv_lds(const SpmvUnit *__restrict__ units, uint32_t nunits,
                                                   const uint32_t *__restrict__ item_row,
                                                   const uint32_t *__restrict__ item_begin,
                                                   const uint32_t *__restrict__ item_len,
                                                   const unsigned short *__restrict__ col16, const T *__restrict__ vals,
                                                   const T *__restrict__ X, int64_t ldx,
                                                   typename AccOf<T>::type *__restrict__ part, uint32_t nrows,
                                                   uint32_t stage_off, uint32_t F16, int merge4) {
    static_assert(W >= 1 && W <= 4, "rows of at most 4 elements");
    using A = typename AccOf<T>::type;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    // staging of (row, sum) pairs: a region per wave of 4 F16 pairs, shared out among the wave's lane groups
    const uint32_t per_wave = 4u * F16 * (4u + (uint32_t)W * (uint32_t)sizeof(A));
    unsigned char *stw = pygim_lds_raw + stage_off + wave * per_wave;
    uint32_t *st_row = reinterpret_cast<uint32_t *>(stw);  // 4 F16 rows, then 4 F16 x W sums
    A *st_val = reinterpret_cast<A *>(stw + 16u * F16);
    for (uint32_t un_i = blockIdx.x; un_i < nunits; un_i += gridDim.x) {
        const SpmvUnit un = units[un_i];
        if (un_i != blockIdx.x) __syncthreads();  // everyone is done with the previous panel before it is overwritten
        // stage the panel: rows [col_base, col_base + pcols) of X, W elements each.  Dense X (ldx == W): 16-byte pieces starting
        // at the 16-byte boundary at or below the panel's first byte (the few leading elements belong to the previous panel and
        // are simply not addressed); else element by element.
        const uint32_t pcols = un.pcols;
        const T *xp = X + (int64_t)un.col_base * ldx;
        const uint32_t total = pcols * (uint32_t)W;
        T *xs = reinterpret_cast<T *>(pygim_lds_raw);
        const uint32_t mis = (uint32_t)((reinterpret_cast<uintptr_t>(xp) & 15u) / sizeof(T));
        if (ldx == (int64_t)W && (reinterpret_cast<uintptr_t>(xp) % sizeof(T)) == 0 && (uint64_t)un.col_base * W >= mis) {  // (never before X)
            constexpr uint32_t PER = 16 / sizeof(T);
            const u32x4_t *src = reinterpret_cast<const u32x4_t *>(xp - mis);
            u32x4_t *dst = reinterpret_cast<u32x4_t *>(xs);
            const uint32_t full = (total + mis) / PER;  // whole 16-byte pieces; the ragged tail element by element (never past X)
            for (uint32_t i = threadIdx.x; i < full; i += blockDim.x) dst[i] = __builtin_nontemporal_load(src + i);
            for (uint32_t i = full * PER + threadIdx.x; i < total + mis; i += blockDim.x) xs[i] = (xp - mis)[i];
            xs += mis;
        } else {
            for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) xs[i] = xp[(int64_t)(i / W) * ldx + (i % W)];
        }
        if (threadIdx.x < (uint32_t)W) xs[pcols * W + threadIdx.x] = T(0);  // the all-zero row (index pcols)
        __syncthreads();
        A *out = part + (size_t)un.panel * nrows * W;
        const char *ir = reinterpret_cast<const char *>(item_row + un.item_off);
        const char *ib = reinterpret_cast<const char *>(item_begin + un.item_off);
        const char *il = reinterpret_cast<const char *>(item_len + un.item_off);
        const char *c16 = reinterpret_cast<const char *>(col16);
        // the five length classes in turn: a lane group of LG lanes per item, 64 / LG groups per wave, each with its share of the
        // wave's staging region
        uint32_t done = 0;
        auto run_class = [&](auto lg_tag, uint32_t n) {
            constexpr int LG = decltype(lg_tag)::value;
            constexpr uint32_t GW = 64u / LG;
            if (n > 0) {
                const uint32_t gw = lane / LG, o = done << 2;
                const uint32_t f_lg = F16 * LG / 16u;
                spmv_lds_class<T, W, HAS_VALS, LG>(ir + o, ib + o, il + o, n, un.slot + un.nslots * (wave * GW + gw),
                                                   un.nslots * nwaves * GW, c16, vals, xs, pcols, out, st_row + gw * f_lg,
                                                   st_val + gw * f_lg * W, f_lg, lane % LG);
            }
            done += n;
        };
        run_class(std::integral_constant<int, 64>{}, un.n64);
        run_class(std::integral_constant<int, 32>{}, un.n32);
        run_class(std::integral_constant<int, 16>{}, un.n16);
        // (merge4: the staging region is too small to give 4-lane groups their own slots -- wide sums -- so the shortest class
        //  rides with the 8-lane one)
        const uint32_t rest = un.n_items - un.n64 - un.n32 - un.n16;
        run_class(std::integral_constant<int, 8>{}, merge4 ? rest : un.n8);
        run_class(std::integral_constant<int, 4>{}, merge4 ? 0u : rest - un.n8);
    }
}

// C[r][j] (+)= sum over the panels of part[p][r][j], in panel order
template <typename T, int W>
__global__ void k_spmv_reduce(const typename AccOf<T>::type *__restrict__ part, uint32_t npanels, uint32_t nrows,
                              T *__restrict__ C, int64_t ldc, int accumulate) {
    using A = typename AccOf<T>::type;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nrows * W) return;
    const size_t stride = (size_t)nrows * W;
    A acc = part[i];
    for (uint32_t p = 1; p < npanels; p++) acc += part[i + p * stride];
    T *c = C + (int64_t)(i / W) * ldc + (i % W);
    *c = accumulate ? from_acc<T>(to_acc<T>(*c) + acc) : from_acc<T>(acc);
}
__global__ void k_zero16(u32x4_t *__restrict__ p, uint64_t n16) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) p[i] = u32x4_t{0, 0, 0, 0};
}

// ---------------------------------------------------------------------------
// COO, equal-nnz split (rows may straddle waves), "wide" layout.
// Wave c owns stored entries [c*chunk, (c+1)*chunk).  Row segments closed on
// both sides are written straight to C (C was zero-filled, or holds the running
// sum when accumulate); a segment open to the left goes to carry[c][0], a
// segment open only to the right goes to carry[c][1].  k_coo_fixup then adds the
// carries of each straddling row in chunk order.  This is the reference's
// nnz-balanced COO split (spmm_default/spmm_mul_coo.c:101-172: "rows may
// straddle", merged by addition :434-437, 478-488), without atomics.
// ---------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_coo_wide(const uint32_t *__restrict__ rowind,
                                                  const uint32_t *__restrict__ colind,
                                                  const T *__restrict__ vals, uint32_t nnz, uint32_t chunk,
                                                  const T *__restrict__ X, int64_t ldx, T *__restrict__ C,
                                                  int64_t ldc, uint32_t w, T *__restrict__ carry,
                                                  int accumulate) {
    using A = typename AccOf<T>::type;
    const int lane = threadIdx.x & 63;
    const uint32_t c = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const uint64_t e0_64 = (uint64_t)c * chunk;
    if (e0_64 >= nnz) return;
    const uint32_t e0 = (uint32_t)e0_64;
    const uint32_t e1 = (uint32_t)min((uint64_t)nnz, e0_64 + chunk);
    const uint32_t f0 = blockIdx.y * (64 * VEC) + lane * VEC;
    const bool lane_on = f0 < w;
    const T *xlane = X + f0;
    const uint32_t first_row = rfl(rowind[e0]);
    const uint32_t last_row = rfl(rowind[e1 - 1]);
    const bool left_open = (e0 > 0) && (rfl(rowind[e0 - 1]) == first_row);
    const bool right_open = (e1 < nnz) && (rfl(rowind[e1]) == last_row);
    T *carry0 = carry + ((int64_t)c * 2) * w;
    T *carry1 = carry0 + w;

    A acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) acc[k] = A(0);
    uint32_t cur = first_row;
    bool seg_is_first = true;

    auto flush = [&](uint32_t r, bool is_first_seg, bool is_last_seg) {
        const bool lo = is_first_seg && left_open;
        const bool ro = is_last_seg && right_open;
        if (lane_on) {
            if (lo || ro) {
                T *p = (lo ? carry0 : carry1) + f0;
#pragma unroll
                for (int k = 0; k < VEC; k++)
                    if (f0 + k < w) p[k] = from_acc<T>(acc[k]);
            } else {
                emit<T, VEC>(C + (int64_t)r * ldc, f0, w, acc, accumulate != 0);
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; k++) acc[k] = A(0);
    };

    for (uint32_t b = e0; b < e1; b += 64) {
        const uint32_t n = min(64u, e1 - b);
        const bool have = (uint32_t)lane < n;
        const uint32_t myrow = have ? __builtin_nontemporal_load(rowind + b + lane) : 0u;
        const uint32_t mycol = have ? __builtin_nontemporal_load(colind + b + lane) : 0u;
        T myval = T(1);
        if (vals) myval = have ? __builtin_nontemporal_load(vals + b + lane) : T(0);
        constexpr int DEPTH = 8;
        uint32_t j = 0;
        while (j < n) {
            // entries of the current row inside this batch: [j, jend)
            const unsigned long long same = __ballot(have && myrow == cur) >> j;
            // count trailing ones of `same` (entries are row-sorted, so they are contiguous)
            const uint32_t run = (same == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~same);
            if (run == 0) {
                flush(cur, seg_is_first, false);
                seg_is_first = false;
                cur = __builtin_amdgcn_readlane(myrow, j);
                continue;
            }
            const uint32_t jend = min(n, j + run);
            for (; j + DEPTH <= jend; j += DEPTH) {
                T x[DEPTH][VEC];
#pragma unroll
                for (int u = 0; u < DEPTH; u++) {
                    const uint32_t cc = __builtin_amdgcn_readlane(mycol, j + u);
                    if (lane_on) load_vec<T, VEC>(xlane + (int64_t)cc * ldx, x[u]);
                }
                if (lane_on) {
#pragma unroll
                    for (int u = 0; u < DEPTH; u++) {
                        if (vals) axpy<T, VEC>(acc, to_acc<T>(bcast_lane<T>(myval, j + u)), x[u]);
                        else add_only<T, VEC>(acc, x[u]);
                    }
                }
            }
            for (; j < jend; j++) {
                const uint32_t cc = __builtin_amdgcn_readlane(mycol, j);
                if (lane_on) {
                    T x[VEC];
                    load_vec<T, VEC>(xlane + (int64_t)cc * ldx, x);
                    if (vals) axpy<T, VEC>(acc, to_acc<T>(bcast_lane<T>(myval, j)), x);
                    else add_only<T, VEC>(acc, x);
                }
            }
        }
    }
    flush(cur, seg_is_first, true);
}

// One wave per chunk; only chunks that START a straddling row do work: they add
// their right-open carry and the left-open carries of the following chunks.
template <typename T>
__global__ __launch_bounds__(256) void k_coo_fixup(const uint32_t *__restrict__ rowind, uint32_t nnz,
                                                   uint32_t chunk, uint32_t nchunks,
                                                   const T *__restrict__ carry, T *__restrict__ C, int64_t ldc,
                                                   uint32_t w, int accumulate) {
    using A = typename AccOf<T>::type;
    const int lane = threadIdx.x & 63;
    const uint32_t c = rfl(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (c >= nchunks) return;
    const uint64_t e0 = (uint64_t)c * chunk;
    const uint64_t e1 = min((uint64_t)nnz, e0 + chunk);
    const uint32_t first_row = rfl(rowind[e0]);
    const uint32_t last_row = rfl(rowind[e1 - 1]);
    const bool left_open = (e0 > 0) && (rfl(rowind[e0 - 1]) == first_row);
    const bool right_open = (e1 < nnz) && (rfl(rowind[e1]) == last_row);
    // starter: its last segment is open to the right and is not the continuation
    // of a row that was already open on the left of this same chunk
    if (!right_open || (left_open && first_row == last_row)) return;
    const uint32_t row = last_row;
    for (uint32_t k = lane; k < w; k += 64) {
        A acc = to_acc<T>(carry[((int64_t)c * 2 + 1) * w + k]);
        uint32_t d = c + 1;
        while (true) {
            acc += to_acc<T>(carry[((int64_t)d * 2) * w + k]);
            const uint64_t d1 = min((uint64_t)nnz, ((uint64_t)d + 1) * chunk);
            // chunk d continues the row iff it is entirely this row and open to the right
            const bool whole = rowind[d1 - 1] == row;
            const bool more = whole && d1 < nnz && rowind[d1] == row;
            if (!more) break;
            d++;
        }
        T *p = C + (int64_t)row * ldc + k;
        *p = accumulate ? from_acc<T>(to_acc<T>(*p) + acc) : from_acc<T>(acc);
    }
}

// ---------------------------------------------------------------------------
// helpers run once per group (create time) or per call on narrow data
// ---------------------------------------------------------------------------
// flag[0] |= 1 if rowind is not non-decreasing; flag[1] |= 1 if an index is out of range
__global__ void k_check_coo(const uint32_t *__restrict__ rowind, const uint32_t *__restrict__ colind,
                            uint32_t nnz, uint32_t nrows, uint32_t ncols, int *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    if (i + 1 < nnz && rowind[i] > rowind[i + 1]) flag[0] = 1;
    if (rowind[i] >= nrows || colind[i] >= ncols) flag[1] = 1;
}
__global__ void k_check_csr(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
                            uint32_t nrows, uint32_t nnz, uint32_t ncols, int *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows && rowptr[i] > rowptr[i + 1]) flag[0] = 1;
    if (i == 0 && (rowptr[0] != 0 || rowptr[nrows] != nnz)) flag[0] = 1;
    if (i < nnz && colind[i] >= ncols) flag[1] = 1;
}
// flag[0] |= 1 when some row's column ids are not non-decreasing (column panels need sorted rows)
__global__ void k_check_sorted_cols(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ colind,
                                    uint32_t nrows, int *flag) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const uint32_t s = rowptr[r], e = rowptr[r + 1];
    for (uint32_t k = s + 1; k < e; k++)
        if (colind[k - 1] > colind[k]) {
            flag[0] = 1;
            return;
        }
}
template <typename T> __global__ void k_check_ones(const T *__restrict__ v, uint32_t n, int *flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(v[i] == T(1))) flag[0] = 1;
}
// entries whose weight is not 1: count, then append (entry index, column, weight - 1) in any order
// (the host sorts the few of them); integer types only (modular arithmetic: v*x == x + (v-1)*x exactly)
template <typename T> __global__ void k_count_non_ones(const T *__restrict__ v, uint32_t n, uint32_t *counter) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool hit = i < n && !(v[i] == T(1));
    const uint64_t m = __ballot(hit);
    if (m && (threadIdx.x & 63) == (uint32_t)__builtin_ctzll(m)) atomicAdd(counter, (uint32_t)__builtin_popcountll(m));
}
template <typename T>
__global__ void k_extract_non_ones(const T *__restrict__ v, const uint32_t *__restrict__ colind, uint32_t n, uint32_t cap,
                                   uint32_t *counter, uint32_t *__restrict__ out_e, uint32_t *__restrict__ out_col,
                                   T *__restrict__ out_val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || v[i] == T(1)) return;
    const uint32_t k = atomicAdd(counter, 1u);
    if (k >= cap) return;
    out_e[k] = (uint32_t)i;
    out_col[k] = colind[i];
    out_val[k] = from_acc<T>(to_acc<T>(v[i]) - to_acc<T>(T(1)));
}
// COO row index -> rowptr (lower bound of each row in the sorted rowind)
__global__ void k_coo_rowptr(const uint32_t *__restrict__ rowind, uint32_t nnz, uint32_t nrows,
                             uint32_t *__restrict__ rowptr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nnz) return;
    const uint32_t hi = (i < nnz) ? rowind[i] : nrows;          // rows < hi end at or before i
    const uint32_t lo = (i == 0) ? 0u : rowind[i - 1] + 1;      // rows >= lo start at or after i
    for (uint32_t r = lo; r <= hi && r <= nrows; r++) rowptr[r] = (uint32_t)i;
    if (i == 0) rowptr[0] = 0;
}
// interleave g separate vectors x_j[n] into Xp[n, g] (SpMV path: the g right-hand
// sides of one call become one narrow dense panel)
template <typename T>
__global__ void k_pack_vectors(const T *const *__restrict__ vecs, uint32_t g, uint64_t n, T *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * g) return;
    const uint64_t r = i / g;
    const uint32_t j = (uint32_t)(i % g);
    out[i] = vecs[j][r];
}


// ---------------------------------------------------------------------------
// quantise -> aggregate -> dequantise around the product: the arithmetic of the
// reference's models/quantize.py:20-42 as the conv layers use it (pyg_gcn_conv.py:130-137), on device.
//   scale = max|x| * 2 / 2^k   (k = 5 / 10 / 20 for int8 / int16 / int32; 20 and a float "quantised"
//   type otherwise);  x_q = round_half_even(x / scale) cast to the type;  out = float(out_q) * scale.
// ---------------------------------------------------------------------------
// max |x| as the bit pattern of a non-negative float (orders like an unsigned integer)
// FLAT = the matrix is one contiguous, 16-byte aligned array whose length is a multiple of 4: float4 accesses
// zero one word (a kernel node instead of a 4-byte memset node: inside a captured HIP graph the memset of the |max| word was
// observed to race with the kernels of the previous aggregation that still read it)
__global__ void k_zero_word(uint32_t *p) { *p = 0u; }
// C[0:nrows, 0:w] = 0 (row stride ldc), as a kernel node for the same reason
template <typename T> __global__ void k_zero_rows(T *__restrict__ C, int64_t ldc, uint64_t nrows, uint32_t w) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows * w) C[(int64_t)(i / w) * ldc + (i % w)] = T(0);
}
template <bool FLAT>
__global__ void k_absmax_bits(const float *__restrict__ x, int64_t ld, uint64_t rows, uint32_t w, uint32_t *out) {
    const uint64_t total = rows * w;
    uint32_t m = 0;
    const uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
    if constexpr (FLAT) {
        const f32x4_t *x4 = reinterpret_cast<const f32x4_t *>(x);
        for (uint64_t i = t0; i < total / 4; i += step) {
            const f32x4_t v = __builtin_nontemporal_load(x4 + i);
            m = max(max(m, __float_as_uint(fabsf(v.x))), __float_as_uint(fabsf(v.y)));
            m = max(max(m, __float_as_uint(fabsf(v.z))), __float_as_uint(fabsf(v.w)));
        }
    } else {
        for (uint64_t i = t0; i < total; i += step) m = max(m, __float_as_uint(fabsf(x[(i / w) * ld + (i % w)])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    // one atomic per BLOCK (thousands of same-address atomics serialise in L2: 0.38 ms for 60 M floats before, 0.05 after)
    __shared__ uint32_t wmax[16];
    const uint32_t wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) wmax[wv] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t k = 1; k < (blockDim.x + 63) / 64; k++) m = max(m, wmax[k]);
        if (m) atomicMax(out, m);
    }
}
template <typename T> struct Pack4 { T v[4]; } __attribute__((aligned(sizeof(T) * 4)));
template <typename T, bool FLAT>
__global__ void k_quantize(const float *__restrict__ x, int64_t ld, uint64_t rows, uint32_t w,
                           const uint32_t *__restrict__ absmax_bits, int log2_range, T *__restrict__ xq,
                           float *__restrict__ scale_out) {
    const float scale = quant_scale(*absmax_bits, log2_range);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && scale_out) *scale_out = scale;
    if constexpr (FLAT) {
        if (i >= rows * w / 4) return;
        const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(x) + i);
        Pack4<T> q;  // torch.round: half to even; true division, as torch's v / scale
        q.v[0] = (T)rintf(v.x / scale);
        q.v[1] = (T)rintf(v.y / scale);
        q.v[2] = (T)rintf(v.z / scale);
        q.v[3] = (T)rintf(v.w / scale);
        reinterpret_cast<Pack4<T> *>(xq)[i] = q;
    } else {
        if (i >= rows * w) return;
        xq[i] = (T)rintf(x[(i / w) * ld + (i % w)] / scale);
    }
}
// the same per-column epilogue as the sweep's fused store, for the unfused path: y = mul[f] * y + add[f], optional ReLU
__global__ void k_post_affine(float *__restrict__ y, uint64_t rows, uint32_t w, const float *__restrict__ mul,
                              const float *__restrict__ add, int relu) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * w) return;
    const uint32_t f = (uint32_t)(i % w);
    float v = mul[f] * y[i] + add[f];
    if (relu) v = fmaxf(v, 0.0f);
    y[i] = v;
}
template <typename T, bool FLAT>
__global__ void k_dequantize(const T *__restrict__ q, uint64_t n, const uint32_t *__restrict__ absmax_bits,
                             int log2_range, float *__restrict__ out) {
    const float scale = 1.0f * quant_scale(*absmax_bits, log2_range);  // scale_edge (1.) * scale_x
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (FLAT) {
        if (i >= n / 4) return;
        const Pack4<T> v = reinterpret_cast<const Pack4<T> *>(q)[i];
        f32x4_t o;
        o.x = (float)v.v[0] * scale;
        o.y = (float)v.v[1] * scale;
        o.z = (float)v.v[2] * scale;
        o.w = (float)v.v[3] * scale;
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4_t *>(out) + i);
    } else {
        if (i < n) out[i] = (float)q[i] * scale;
    }
}

// ---- merging the sparse parts (column blocks with local ids) of a group into one matrix, one-time --------------
// cursor[r] starts as the merged row pointer; after part i has been scattered it is advanced by that part's row length
__global__ void k_merge_count(const uint32_t *__restrict__ rowptr_i, uint32_t nrows, uint32_t *__restrict__ merged_rowptr) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= nrows) merged_rowptr[r] += rowptr_i[r];  // sum of prefix arrays = prefix array of the concatenated rows
}
template <typename T>
__global__ void k_merge_scatter(const uint32_t *__restrict__ rowptr_i, const uint32_t *__restrict__ rowind_i,
                                const uint32_t *__restrict__ col_i, const T *__restrict__ val_i, uint32_t nrows, uint32_t nnz_i,
                                uint32_t col_offset, const uint32_t *__restrict__ cursor, uint32_t *__restrict__ out_col,
                                T *__restrict__ out_val) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz_i) return;
    uint32_t row;
    if (rowind_i) {
        row = rowind_i[e];
    } else {  // last row whose start is <= e
        uint32_t lo = 0, hi = nrows;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo + 1) >> 1);
            if (rowptr_i[mid] <= (uint32_t)e) lo = mid;
            else hi = mid - 1;
        }
        row = lo;
    }
    const uint32_t dst = cursor[row] + ((uint32_t)e - rowptr_i[row]);
    out_col[dst] = col_i[e] + col_offset;
    if (out_val) out_val[dst] = val_i ? val_i[e] : T(1);
}
__global__ void k_merge_advance(const uint32_t *__restrict__ rowptr_i, uint32_t nrows, uint32_t *__restrict__ cursor) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrows) cursor[r] += rowptr_i[r + 1] - rowptr_i[r];
}

// 16-bit panel-local column ids for the panel sweep: col16[e] = colind[e] - (first column of e's panel)
__global__ void k_make_col16(const uint32_t *__restrict__ colind, uint64_t nnz, uint32_t panel_cols,
                             unsigned short *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnz) out[i] = (unsigned short)(colind[i] % panel_cols);
}

}  // namespace pygim
