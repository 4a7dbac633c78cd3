// LDS-staged SpMM prototype (round 2): a workgroup owns R = NWC*8*K rows of one
// 128-byte feature slice with the accumulators in VGPRs and streams the slice-major X through LDS
// in double-buffered column chunks; every stored entry becomes one ds_read_b128 per lane instead of
// one 128-byte L1 request.  Experiment only -- driven by scripts/lds_proto/run_proto.py.
//
// Schedule (built on the device by run_proto.py): rows sorted by length, dealt in octets to the
// consumer waves; for (tile, chunk, wave, j) a segment of `nblk` blocks of 2 steps; one step holds
// the 8 lane groups' LDS line ids (16 bit; the line KC is all zeros = padding).  Rows keep their
// stored order (one lane group sums a row sequentially), so floats match the CPU loop bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

extern __shared__ char lds[];

// consumer part of one workgroup: K accumulators per lane group
template <int K, int NWC>
__device__ __forceinline__ void consume_tile(uint32_t tile, uint32_t slice, uint32_t wave, int grp, int li,
                                             const uint32_t *__restrict__ blk_off, const uint8_t *__restrict__ seg_n,
                                             const uint32_t *__restrict__ rowmap, uint32_t rowmap_base,
                                             float *__restrict__ C, int64_t ldc, uint32_t nchunks, uint32_t xbuf_bytes,
                                             uint32_t ids_base, uint32_t idb) {
    f32x4 acc[K];
#pragma unroll
    for (int j = 0; j < K; j++) acc[j] = f32x4{0, 0, 0, 0};
    __syncthreads();
    for (uint32_t c = 0; c < nchunks; c++) {
        const uint32_t wg = tile * nchunks + c;
        const uint32_t wseg = wg * NWC + wave;  // uniform
        const uint32_t boff = blk_off[wseg] - blk_off[wg * NWC];  // block offset inside this chunk's stream
        const u32x4 nn = *reinterpret_cast<const u32x4 *>(seg_n + (size_t)wseg * 16);
        uint32_t ia = ids_base + (c & 1) * idb + (boff * 8 + grp) * 4;
        const uint32_t lbase = (c & 1) * xbuf_bytes + li * 16;
        uint32_t q0 = *reinterpret_cast<const uint32_t *>(lds + ia);
        uint32_t q1 = *reinterpret_cast<const uint32_t *>(lds + ia + 32);
        ia += 64;
#pragma unroll
        for (int j = 0; j < K; j++) {
            const uint32_t nj = (nn[j >> 2] >> (8 * (j & 3))) & 0xffu;
            uint32_t t = 0;
#pragma nounroll
            for (; t + 2 <= nj; t += 2) {
                const uint32_t c0 = q0, c1 = q1;
                q0 = *reinterpret_cast<const uint32_t *>(lds + ia);
                q1 = *reinterpret_cast<const uint32_t *>(lds + ia + 32);
                ia += 64;
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(lds + ((c0 & 0x3ffu) << 7) + lbase);
                const f32x4 x1 = *reinterpret_cast<const f32x4 *>(lds + ((c0 >> 16) << 7) + lbase);
                const f32x4 x2 = *reinterpret_cast<const f32x4 *>(lds + ((c1 & 0x3ffu) << 7) + lbase);
                const f32x4 x3 = *reinterpret_cast<const f32x4 *>(lds + ((c1 >> 16) << 7) + lbase);
                acc[j] += x0;
                acc[j] += x1;
                acc[j] += x2;
                acc[j] += x3;
            }
            if (t < nj) {
                const uint32_t cur = q0;
                q0 = q1;
                q1 = *reinterpret_cast<const uint32_t *>(lds + ia);
                ia += 32;
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(lds + ((cur & 0x3ffu) << 7) + lbase);
                const f32x4 x1 = *reinterpret_cast<const f32x4 *>(lds + ((cur >> 16) << 7) + lbase);
                acc[j] += x0;
                acc[j] += x1;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < K; j++) {
        const uint32_t row = rowmap[rowmap_base + ((size_t)wave * K + j) * 8 + grp];
        if (row != 0xffffffffu)
            *reinterpret_cast<f32x4 *>(C + (int64_t)row * ldc + slice * 32 + li * 4) = acc[j];
    }
}

// Window variant: the 8 lane groups of a wave no longer share ONE accumulator index.  During phase j a group works on its
// row j or already on its row j + 1 (bit 15 of the step's id says which), so a group that finishes row j early does not wait
// for the slowest one; a phase ends when every group has finished row j.  One 16-bit id per group and step.
// exec-masked adds, hand-written: lanes whose id has bit 15 set (sign bit of the 16-bit load) add to acc[j + 1], the others
// to acc[j]; one compare, four packed adds, three scalar moves -- no branches, no selects
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ADD_SEL(ID, X)                                                                                              \
    {                                                                                                               \
        f32x2 a0l = {acc[j].x, acc[j].y}, a0h = {acc[j].z, acc[j].w};                                               \
        f32x2 a1l = {acc[j + 1].x, acc[j + 1].y}, a1h = {acc[j + 1].z, acc[j + 1].w};                               \
        const f32x2 xl = {(X).x, (X).y}, xh = {(X).z, (X).w};                                                       \
        asm volatile("v_cmp_gt_i32 vcc, 0, %[id]\n\t"                                                               \
                     "s_mov_b64 exec, vcc\n\t"                                                                      \
                     "v_pk_add_f32 %[a1l], %[a1l], %[xl]\n\t"                                                       \
                     "v_pk_add_f32 %[a1h], %[a1h], %[xh]\n\t"                                                       \
                     "s_not_b64 exec, exec\n\t"                                                                     \
                     "v_pk_add_f32 %[a0l], %[a0l], %[xl]\n\t"                                                       \
                     "v_pk_add_f32 %[a0h], %[a0h], %[xh]\n\t"                                                       \
                     "s_mov_b64 exec, -1"                                                                           \
                     : [a0l] "+v"(a0l), [a0h] "+v"(a0h), [a1l] "+v"(a1l), [a1h] "+v"(a1h)                           \
                     : [id] "v"(ID), [xl] "v"(xl), [xh] "v"(xh)                                                     \
                     : "vcc", "scc");                                                                                    \
        acc[j] = f32x4{a0l.x, a0l.y, a0h.x, a0h.y};                                                                 \
        acc[j + 1] = f32x4{a1l.x, a1l.y, a1h.x, a1h.y};                                                             \
    }
template <int K, int NWC>
__device__ __forceinline__ void consume_tile_win(uint32_t tile, uint32_t slice, uint32_t wave, int grp, int li,
                                                 const uint32_t *__restrict__ blk_off, const uint8_t *__restrict__ seg_n,
                                                 const uint32_t *__restrict__ rowmap, uint32_t rowmap_base,
                                                 float *__restrict__ C, int64_t ldc, uint32_t nchunks,
                                                 uint32_t xbuf_bytes, uint32_t ids_base, uint32_t idb) {
    f32x4 acc[K + 1];
#pragma unroll
    for (int j = 0; j <= K; j++) acc[j] = f32x4{0, 0, 0, 0};
    __syncthreads();
    for (uint32_t c = 0; c < nchunks; c++) {
        const uint32_t wg = tile * nchunks + c;
        const uint32_t wseg = wg * NWC + wave;  // uniform
        const uint32_t boff = blk_off[wseg] - blk_off[wg * NWC];  // step offset inside this chunk's stream
        const u32x4 nn = *reinterpret_cast<const u32x4 *>(seg_n + (size_t)wseg * 16);
        uint32_t ia = ids_base + (c & 1) * idb + (boff * 8 + grp) * 2;
        const uint32_t lbase = (c & 1) * xbuf_bytes + li * 16;
        auto ldid = [&](uint32_t off) { return (int32_t) * reinterpret_cast<const int16_t *>(lds + ia + off); };
        auto rdx = [&](int32_t id) { return *reinterpret_cast<const f32x4 *>(lds + ((__builtin_amdgcn_ubfe((uint32_t)id, 0, 10) << 7) + lbase)); };
        int32_t q0 = ldid(0), q1 = ldid(16);
        ia += 32;
#pragma unroll
        for (int j = 0; j < K; j++) {
            const uint32_t nj = (nn[j >> 2] >> (8 * (j & 3))) & 0xffu;
            uint32_t t = 0;
#pragma nounroll
            for (; t + 2 <= nj; t += 2) {
                const int32_t c0 = q0, c1 = q1;
                q0 = ldid(0);
                q1 = ldid(16);
                ia += 32;
                const f32x4 x0 = rdx(c0), x1 = rdx(c1);
                ADD_SEL(c0, x0)
                ADD_SEL(c1, x1)
            }
            if (t < nj) {
                const int32_t c0 = q0;
                q0 = q1;
                q1 = ldid(0);
                ia += 16;
                const f32x4 x0 = rdx(c0);
                ADD_SEL(c0, x0)
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < K; j++) {
        const uint32_t row = rowmap[rowmap_base + ((size_t)wave * K + j) * 8 + grp];
        if (row != 0xffffffffu)
            *reinterpret_cast<f32x4 *>(C + (int64_t)row * ldc + slice * 32 + li * 4) = acc[j];
    }
}

// tileinfo[t] = {K, rowmap offset}
template <int NWC, int NWL, int FLAT>
__global__ __launch_bounds__((NWC + NWL) * 64) void k_lds(const char *__restrict__ Xs, int64_t slice_bytes,
                                                         const uint32_t *__restrict__ blk_off,
                                                         const uint8_t *__restrict__ seg_n,
                                                         const char *__restrict__ stream,
                                                         const uint32_t *__restrict__ rowmap,
                                                         const uint32_t *__restrict__ tileinfo, float *__restrict__ C,
                                                         int64_t ldc, uint32_t nchunks, uint32_t KC, uint32_t idb,
                                                         uint32_t tile0, uint32_t nslices) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 3, li = lane & 7;
    const uint32_t slice = blockIdx.x % nslices;
    const uint32_t tile = tile0 + blockIdx.x / nslices;
    const uint32_t xbuf_bytes = (KC + 1) * 128;
    const uint32_t ids_base = 2 * xbuf_bytes;
    if (threadIdx.x < 16) {  // the zero lines
        const uint32_t b = threadIdx.x >> 3;
        *reinterpret_cast<f32x4 *>(lds + b * xbuf_bytes + KC * 128 + (threadIdx.x & 7) * 16) = f32x4{0, 0, 0, 0};
    }
    if (wave >= NWC) {
        const uint32_t wl = wave - NWC;
        const char *xsl = Xs + (int64_t)slice * slice_bytes;
        const uint32_t chunk_bytes = KC * 128;
        const uint32_t per = chunk_bytes / NWL;  // multiple of 1024
        auto fill = [&](uint32_t c) {
            const char *src = xsl + (int64_t)c * chunk_bytes + wl * per + lane * 16;
            char *dst = lds + (c & 1) * xbuf_bytes + wl * per;
            for (uint32_t o = 0; o < per; o += 1024)
                __builtin_amdgcn_global_load_lds((const GLB_AS void *)(src + o), (LDS_AS void *)(dst + o), 16, 0, 0);
            // the chunk's id stream (all consumer waves, contiguous), 1 KiB pieces dealt to the loader waves
            const uint32_t wg = tile * nchunks + c;
            const uint32_t b0 = blk_off[wg * NWC], b1 = blk_off[(wg + 1) * NWC];
            const uint32_t unit = FLAT ? 16u : 32u;  // bytes per stream unit: one step (window variant) or a block of two
            const uint32_t bytes = (b1 - b0) * unit + 64;  // + the consumers' look-ahead
            const char *isrc = stream + (size_t)b0 * unit + lane * 16;
            char *idst = lds + ids_base + (c & 1) * idb;
            for (uint32_t o = wl * 1024; o < bytes; o += NWL * 1024)
                __builtin_amdgcn_global_load_lds((const GLB_AS void *)(isrc + o), (LDS_AS void *)(idst + o), 16, 0, 0);
        };
        fill(0);
        __syncthreads();
        for (uint32_t c = 0; c < nchunks; c++) {
            if (c + 1 < nchunks) fill(c + 1);
            __syncthreads();
        }
        return;
    }
    const uint32_t K = tileinfo[2 * tile], rmb = tileinfo[2 * tile + 1];
    if constexpr (FLAT) {
#define ARGF tile, slice, wave, grp, li, blk_off, seg_n, rowmap, rmb, C, ldc, nchunks, xbuf_bytes, ids_base, idb
        switch (K) {
        case 1: consume_tile_win<1, NWC>(ARGF); break;
        case 2: consume_tile_win<2, NWC>(ARGF); break;
        case 4: consume_tile_win<4, NWC>(ARGF); break;
        case 8: consume_tile_win<8, NWC>(ARGF); break;
        default: consume_tile_win<16, NWC>(ARGF); break;
        }
#undef ARGF
        return;
    }
#define ARGS tile, slice, wave, grp, li, blk_off, seg_n, rowmap, rmb, C, ldc, nchunks, xbuf_bytes, ids_base, idb
    switch (K) {
    case 1: consume_tile<1, NWC>(ARGS); break;
    case 2: consume_tile<2, NWC>(ARGS); break;
    case 4: consume_tile<4, NWC>(ARGS); break;
    case 8: consume_tile<8, NWC>(ARGS); break;
    default: consume_tile<16, NWC>(ARGS); break;
    }
#undef ARGS
}

template <int NWC, int NWL, int FLAT>
static float launch(const void *Xs, int64_t slice_bytes, const void *blk_off, const void *seg_n, const void *stream,
                    const void *rowmap, const void *tileinfo, void *C, int64_t ldc, uint32_t nchunks, uint32_t KC,
                    uint32_t idb, uint32_t tile0, uint32_t ntiles, uint32_t nslices, int iters) {
    auto kern = k_lds<NWC, NWL, FLAT>;
    const size_t shmem = 2 * (size_t)(KC + 1) * 128 + 2 * (size_t)idb;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) { printf("attr: %s\n", hipGetErrorString(e)); return -1.f; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid(ntiles * nslices), block((NWC + NWL) * 64);
    auto go = [&]() {
        hipLaunchKernelGGL(kern, grid, block, shmem, 0, (const char *)Xs, slice_bytes, (const uint32_t *)blk_off,
                           (const uint8_t *)seg_n, (const char *)stream, (const uint32_t *)rowmap,
                           (const uint32_t *)tileinfo, (float *)C, ldc, nchunks, KC, idb, tile0, nslices);
    };
    go();
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("run: %s\n", hipGetErrorString(e)); return -1.f; }
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; i++) go();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms / iters;
}

extern "C" float proto_run(int FLAT, int NWC, int NWL, const void *Xs, int64_t slice_bytes, const void *blk_off,
                           const void *seg_n, const void *stream, const void *rowmap, const void *tileinfo, void *C,
                           int64_t ldc, uint32_t nchunks, uint32_t KC, uint32_t idb, uint32_t tile0, uint32_t ntiles,
                           uint32_t nslices, int iters) {
#define CASE(c, l)                                                                                               \
    if (NWC == c && NWL == l)                                                                                    \
        return FLAT ? launch<c, l, 1>(Xs, slice_bytes, blk_off, seg_n, stream, rowmap, tileinfo, C, ldc, nchunks, KC, idb, tile0, ntiles, nslices, iters) : launch<c, l, 0>(Xs, slice_bytes, blk_off, seg_n, stream, rowmap, tileinfo, C, ldc, nchunks, KC, idb, \
                            tile0, ntiles, nslices, iters);
    CASE(14, 2) CASE(15, 1) CASE(12, 4)
#undef CASE
    printf("no such variant\n");
    return -2.f;
}
