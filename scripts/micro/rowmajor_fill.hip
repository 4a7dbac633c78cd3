// Can the code-stream product stream X straight from the caller's ROW-MAJOR matrix instead of from a slice-major copy (k_slice_pack: 0.07 ms per product,
// 16 % of a 1/8 row share)?  (Round 6.)  A slice of a row-major [N][256] float matrix is 256 bytes at a fixed offset of every 1 KiB row: the LDS-DMA of one wave
// instruction then reads 4 pieces of 256 bytes at a 1 KiB stride instead of 1 KiB contiguous, and the slices of a row sit side by side in memory.
// Workgroups as in the product: 8 waves, ring of 5 x 32 KiB chunks (128 rows x 256 bytes), 4 pieces per wave and chunk, a consumer loop of `groups` x 8 entries
// per wave and chunk beside the fill, a barrier per chunk; 256 workgroups, XCD x streams slices by the product's rule (sx slices per XCD).
//   LAYOUT 0: slice-major copy (what the library does today)     LAYOUT 1: row-major, 1 KiB rows
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/rowmajor_fill.hip -o /tmp/rowmajor_fill && /tmp/rowmajor_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef const __attribute__((address_space(1))) void *gptr;
typedef __attribute__((address_space(3))) void *lptr;

__device__ __forceinline__ void consumer_group(float (&acc)[16], uint32_t lbase) {
    asm volatile(
        "ds_read2st64_b32 v[40:41], %[b] offset0:3 offset1:77\n"
        "ds_read2st64_b32 v[42:43], %[b] offset0:19 offset1:201\n"
        "ds_read2st64_b32 v[44:45], %[b] offset0:130 offset1:54\n"
        "ds_read2st64_b32 v[46:47], %[b] offset0:99 offset1:240\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_add_f32 %[a0], v40, %[a0]\n v_add_f32 %[a1], v41, %[a1]\n v_add_f32 %[a2], v42, %[a2]\n v_add_f32 %[a3], v43, %[a3]\n"
        "v_add_f32 %[a4], v44, %[a4]\n v_add_f32 %[a5], v45, %[a5]\n v_add_f32 %[a6], v46, %[a6]\n v_add_f32 %[a7], v47, %[a7]\n"
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]), [a6] "+v"(acc[6]), [a7] "+v"(acc[7])
        : [b] "v"(lbase)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | (((N >> 4) & 3) << 14)); }

template <int LAYOUT>
__global__ __launch_bounds__(512) void k_stream(const char *src, uint64_t slice_stride, uint32_t nchunks, uint32_t groups, int sx, float *out) {
    extern __shared__ char lds[];
    constexpr uint32_t NW = 8, PIECES = 4, NBUF = 5, CHUNK = NW * PIECES * 1024;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x, xcd = b & 7, i = b >> 3;
    // the product's rule for 4 slices: xcd_group = 2 * sx XCDs share sx slices
    const uint32_t group = (8 / 4) * sx;
    const uint32_t slice = (xcd / group) * sx + i % sx;
    float acc[16];
    for (int k = 0; k < 16; k++) acc[k] = 0.f;
    const uint32_t lbase = lane * 4;
    auto issue = [&](uint32_t j) {
        const uint32_t lb = (j % NBUF) * CHUNK + wave * PIECES * 1024;
#pragma unroll
        for (uint32_t p = 0; p < PIECES; p++) {
            const char *g;
            if (LAYOUT == 0) g = src + (uint64_t)slice * slice_stride + (uint64_t)j * CHUNK + wave * PIECES * 1024 + p * 1024 + lane * 16;
            else g = src + ((uint64_t)j * 128 + wave * 16 + p * 4 + (lane >> 4)) * 1024 + slice * 256 + (lane & 15) * 16;
            __builtin_amdgcn_global_load_lds((gptr)g, (lptr)(lds + lb + p * 1024), 16, 0, 0);
        }
    };
    for (uint32_t j = 0; j < NBUF - 1 && j < nchunks; j++) issue(j);
    wait_vm<PIECES * (NBUF - 2)>();
    __syncthreads();
    for (uint32_t j = 0; j < nchunks; j++) {
        if (j + NBUF - 1 < nchunks) issue(j + NBUF - 1);
        for (uint32_t g = 0; g < groups; g++) consumer_group(acc, lbase);
        wait_vm<PIECES * (NBUF - 2)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    }
    float t = 0;
    for (int k = 0; k < 16; k++) t += acc[k];
    if (t == 12345.678f) out[threadIdx.x] = t + ((float *)lds)[threadIdx.x];
}

template <int LAYOUT>
static void run(const char *name, const char *src, uint64_t slice_stride, uint32_t nchunks, uint32_t groups, int sx, float *out) {
    auto fn = k_stream<LAYOUT>;
    CHECK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(fn, dim3(256), dim3(512), 163840, 0, src, slice_stride, nchunks, groups, sx, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("%-28s sx %d groups %2u: %7.3f ms  %6.1f GB/s per CU  %6.2f TB/s  %5.3f us per 32 KiB chunk\n", name, sx, groups, best, nchunks * 32768.0 / best * 1e-6,
           256.0 * nchunks * 32768.0 / best * 1e-9, best * 1e3 / nchunks);
    fflush(stdout);
}

int main() {
    CHECK(hipSetDevice(0));
    const uint32_t nrows = 232960;                 // 1 820 chunks of 128 rows
    const uint64_t bytes = (uint64_t)nrows * 1024;
    char *src;
    float *out;
    CHECK(hipMalloc(&src, bytes + (1 << 20)));
    CHECK(hipMemset(src, 0, bytes + (1 << 20)));
    CHECK(hipMalloc(&out, 1 << 16));
    const uint32_t nchunks = nrows / 128;
    for (uint32_t groups : {0u, 6u, 8u}) {
        for (int sx : {1, 2, 4}) {
            run<0>("slice-major copy", src, (uint64_t)nrows * 256, nchunks, groups, sx, out);
            run<1>("row-major, 1 KiB rows", src, 0, nchunks, groups, sx, out);
        }
    }
    return 0;
}
