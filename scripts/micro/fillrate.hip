// How fast can ONE compute unit land a contiguous stream in its LDS, by which mechanism, and what does it cost the LDS reads that
// run beside it?  (Round 4: the code-stream product k_lds_code_* is bound by the chunk fills -- ~70 GB/s per CU through
// global_load_lds_dwordx4 -- profiles/r03_lds_kernel.md.)
//
// A workgroup of NW waves streams NCHUNKS chunks of NW * PIECES KiB from a slice of `src` (the 32 workgroups of an XCD read the same
// slice, nearly in step, like the product's) through an LDS ring of NBUF chunk buffers:
//     MODE 0: LDS-DMA            global_load_lds_dwordx4, 1 KiB per wave instruction
//     MODE 1: register staging   global_load_dwordx4 -> ds_write_b128 (4 VGPRs per piece in flight)
//     MODE 2: half and half      pieces [0, PIECES / 2) by DMA, the rest through registers
//     MODE 3: no fill            (the consumer loop alone)
// Beside the fill every wave runs `groups` groups of 4 x ds_read2st64_b32 + 8 x v_add_f32 per chunk (the product's entry loop from a
// cached loop: 8 stored entries per group), and the workgroup meets at a barrier per chunk (barrier = 1) as the product does.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/fillrate.hip -o /tmp/fillrate && /tmp/fillrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef const __attribute__((address_space(1))) void *gptr;
typedef __attribute__((address_space(3))) void *lptr;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void consumer_group(float (&acc)[16], uint32_t lbase) {
    float x0, x1, x2, x3, x4, x5, x6, x7;
    asm volatile(
        "ds_read2st64_b32 v[40:41], %[b] offset0:3 offset1:77\n"
        "ds_read2st64_b32 v[42:43], %[b] offset0:19 offset1:201\n"
        "ds_read2st64_b32 v[44:45], %[b] offset0:130 offset1:54\n"
        "ds_read2st64_b32 v[46:47], %[b] offset0:99 offset1:240\n"
        "s_waitcnt lgkmcnt(4)\n"
        "v_add_f32 %[a0], v48, %[a0]\n v_add_f32 %[a1], v49, %[a1]\n v_add_f32 %[a2], v50, %[a2]\n v_add_f32 %[a3], v51, %[a3]\n"
        "v_add_f32 %[a4], v52, %[a4]\n v_add_f32 %[a5], v53, %[a5]\n v_add_f32 %[a6], v54, %[a6]\n v_add_f32 %[a7], v55, %[a7]\n"
        "ds_read2st64_b32 v[48:49], %[b] offset0:7 offset1:177\n"
        "ds_read2st64_b32 v[50:51], %[b] offset0:219 offset1:21\n"
        "ds_read2st64_b32 v[52:53], %[b] offset0:30 offset1:154\n"
        "ds_read2st64_b32 v[54:55], %[b] offset0:199 offset1:40\n"
        "s_waitcnt lgkmcnt(4)\n"
        "v_add_f32 %[a8], v40, %[a8]\n v_add_f32 %[a9], v41, %[a9]\n v_add_f32 %[a10], v42, %[a10]\n v_add_f32 %[a11], v43, %[a11]\n"
        "v_add_f32 %[a12], v44, %[a12]\n v_add_f32 %[a13], v45, %[a13]\n v_add_f32 %[a14], v46, %[a14]\n v_add_f32 %[a15], v47, %[a15]\n"
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [a4] "+v"(acc[4]), [a5] "+v"(acc[5]), [a6] "+v"(acc[6]),
          [a7] "+v"(acc[7]), [a8] "+v"(acc[8]), [a9] "+v"(acc[9]), [a10] "+v"(acc[10]), [a11] "+v"(acc[11]), [a12] "+v"(acc[12]),
          [a13] "+v"(acc[13]), [a14] "+v"(acc[14]), [a15] "+v"(acc[15])
        : [b] "v"(lbase)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "memory");
    (void)x0; (void)x1; (void)x2; (void)x3; (void)x4; (void)x5; (void)x6; (void)x7;
}

template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | (((N >> 4) & 3) << 14)); }

// MODE / NW / PIECES / NBUF as above; NT: nt bit on the fill loads
template <int MODE, int NW, int PIECES, int NBUF, int NT>
__global__ __launch_bounds__(NW * 64) void k_stream(const char *src, uint64_t slice_stride, uint32_t nchunks, uint32_t groups, int barrier, int nslices, float *out) {
    extern __shared__ char lds[];
    constexpr uint32_t CHUNK = NW * PIECES * 1024;
    constexpr int PD = MODE == 0 ? PIECES : (MODE == 2 ? PIECES / 2 : 0);   // pieces by DMA
    constexpr int PR = (MODE == 1 || MODE == 2) ? PIECES - PD : 0;          // pieces through registers
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t xcd = blockIdx.x & 7;
    const char *s = src + (uint64_t)(xcd % nslices) * slice_stride + (uint64_t)wave * PIECES * 1024;
    const uint32_t lw = wave * PIECES * 1024;
    float acc[16];
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    const uint32_t lbase = lane * 4;
    u32x4 r[PR > 0 ? PR : 1];
    auto issue = [&](uint32_t j) {   // chunk j -> buffer j % NBUF (DMA pieces), registers (the rest)
        const char *p = s + (uint64_t)j * CHUNK;
        const uint32_t lb = (j % NBUF) * CHUNK + lw;
#pragma unroll
        for (int i = 0; i < PD; i++)
            __builtin_amdgcn_global_load_lds((gptr)(p + i * 1024 + lane * 16), (lptr)(lds + lb + i * 1024), 16, 0, NT ? 2 : 0);
#pragma unroll
        for (int i = 0; i < PR; i++) {
            if (NT) r[i] = __builtin_nontemporal_load((const u32x4 *)(p + (PD + i) * 1024 + lane * 16));
            else r[i] = *(const u32x4 *)(p + (PD + i) * 1024 + lane * 16);
        }
    };
    auto land_regs = [&](uint32_t j) {
        const uint32_t lb = (j % NBUF) * CHUNK + lw;
#pragma unroll
        for (int i = 0; i < PR; i++) *(u32x4 *)(lds + lb + (PD + i) * 1024 + lane * 16) = r[i];
    };
    if (MODE != 3) {
        // register staging holds ONE chunk's pieces: NBUF - 1 chunks ahead only for the DMA-only mode
        constexpr int AHEAD = MODE == 0 ? NBUF - 1 : 1;
        for (uint32_t j = 0; j < (uint32_t)AHEAD && j < nchunks; j++) {
            issue(j);
            if (PR) { __builtin_amdgcn_s_waitcnt(0x0F70); land_regs(j); }
        }
        wait_vm<PD * (AHEAD - 1)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __syncthreads();
        for (uint32_t j = 0; j < nchunks; j++) {
            if (j + AHEAD < nchunks) issue(j + AHEAD);
            for (uint32_t g = 0; g < groups; g += 2) consumer_group(acc, lbase + (j % NBUF) * 0);   // (reads anywhere in the ring: timing only)
            if (PR && j + AHEAD < nchunks) { __builtin_amdgcn_s_waitcnt(0x0F70); land_regs(j + AHEAD); }
            wait_vm<PD * (AHEAD - 1)>();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (barrier) __builtin_amdgcn_s_barrier();
        }
    } else {
        for (uint32_t j = 0; j < nchunks; j++) {
            for (uint32_t g = 0; g < groups; g += 2) consumer_group(acc, lbase);
            if (barrier) __builtin_amdgcn_s_barrier();
        }
    }
    float t = 0;
    for (int i = 0; i < 16; i++) t += acc[i];
    if (t == 12345.678f) out[threadIdx.x] = t + ((float *)lds)[threadIdx.x];
}

template <int MODE, int NW, int PIECES, int NBUF, int NT>
static void run(const char *name, const char *src, uint64_t slice_stride, uint32_t nchunks, uint32_t groups, int barrier, int nslices, int grid, float *out) {
    auto fn = k_stream<MODE, NW, PIECES, NBUF, NT>;
    constexpr uint32_t CHUNK = NW * PIECES * 1024;
    CHECK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), 163840, 0, src, slice_stride, nchunks, groups, barrier, nslices, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)nchunks * CHUNK, entries = (double)nchunks * groups * 8 * NW;
    printf("%-34s NW %2d chunk %3u KiB x%d nt%d grid %3d groups %3u bar %d: %7.3f ms  fill %6.1f GB/s/CU (%5.1f B/clk @2.4)  %5.2f CU-clk/entry  %6.0f clk/chunk\n",
           name, NW, CHUNK >> 10, NBUF, NT, grid, groups, barrier, best, MODE == 3 ? 0.0 : bytes / best * 1e-6, MODE == 3 ? 0.0 : bytes / (best * 1e-3 * 2.4e9),
           entries > 0 ? best * 1e-3 * 2.4e9 / entries : 0.0, best * 1e-3 * 2.4e9 / nchunks);
    fflush(stdout);
}

int main() {
    CHECK(hipSetDevice(0));
    const uint64_t slice = 59648000ull / 81920 * 81920 + 81920 * 4;   // ~59.6 MB, whole chunks of every geometry below
    char *src;
    float *out;
    CHECK(hipMalloc(&src, slice * 4 + (1 << 20)));
    CHECK(hipMemset(src, 0, slice * 4 + (1 << 20)));
    CHECK(hipMalloc(&out, 1 << 16));
    const uint32_t n80 = 728, n64 = 910, n48 = 1213, n40 = 1456;
    for (int grid : {256, 8}) {
        printf("=== %d workgroups (%s)\n", grid, grid == 256 ? "every CU busy, 32 per XCD on one slice" : "one per XCD");
        // fill alone
        run<0, 16, 5, 2, 0>("DMA", src, slice, n80, 0, 1, 4, grid, out);
        run<0, 16, 5, 2, 0>("DMA no barrier", src, slice, n80, 0, 0, 4, grid, out);
        run<0, 16, 5, 2, 1>("DMA nt", src, slice, n80, 0, 1, 4, grid, out);
        run<0, 16, 3, 3, 0>("DMA 3 x 48K", src, slice, n48, 0, 1, 4, grid, out);
        run<0, 16, 2, 5, 0>("DMA 5 x 32K", src, slice, slice / 32768, 0, 1, 4, grid, out);
        run<0, 8, 10, 2, 0>("DMA 8 waves", src, slice, n80, 0, 1, 4, grid, out);
        run<0, 8, 5, 4, 0>("DMA 8 waves 4 x 40K", src, slice, n40, 0, 1, 4, grid, out);
        run<0, 4, 20, 2, 0>("DMA 4 waves", src, slice, n80, 0, 1, 4, grid, out);
        run<1, 16, 5, 2, 0>("registers", src, slice, n80, 0, 1, 4, grid, out);
        run<1, 16, 5, 2, 0>("registers no barrier", src, slice, n80, 0, 0, 4, grid, out);
        run<1, 16, 5, 2, 1>("registers nt", src, slice, n80, 0, 1, 4, grid, out);
        run<1, 8, 10, 2, 0>("registers 8 waves", src, slice, n80, 0, 1, 4, grid, out);
        run<2, 16, 5, 2, 0>("half DMA half registers", src, slice, n80, 0, 1, 4, grid, out);
        run<2, 8, 10, 2, 0>("half and half 8 waves", src, slice, n80, 0, 1, 4, grid, out);
        run<2, 16, 4, 2, 0>("half and half 64K", src, slice, n64, 0, 1, 4, grid, out);
        // the consumer loop alone (cached code): 16 waves x 48 entries, 8 waves x 152 (2-round tiles)
        run<3, 16, 5, 2, 0>("consumer alone", src, slice, n80, 6, 1, 4, grid, out);
        run<3, 16, 5, 2, 0>("consumer alone no barrier", src, slice, n80, 6, 0, 4, grid, out);
        run<3, 8, 10, 2, 0>("consumer alone 8 waves", src, slice, n80, 20, 1, 4, grid, out);
        // both
        run<0, 16, 5, 2, 0>("DMA + consumer", src, slice, n80, 6, 1, 4, grid, out);
        run<0, 16, 5, 2, 0>("DMA + consumer no barrier", src, slice, n80, 6, 0, 4, grid, out);
        run<0, 16, 3, 3, 0>("DMA 3 x 48K + consumer", src, slice, n48, 4, 1, 4, grid, out);
        run<0, 8, 10, 2, 0>("DMA + consumer 8 waves", src, slice, n80, 20, 1, 4, grid, out);
        run<0, 8, 5, 4, 0>("DMA 4 x 40K + consumer 8 waves", src, slice, n40, 10, 1, 4, grid, out);
        run<1, 16, 5, 2, 0>("registers + consumer", src, slice, n80, 6, 1, 4, grid, out);
        run<2, 16, 5, 2, 0>("half/half + consumer", src, slice, n80, 6, 1, 4, grid, out);
        run<2, 8, 10, 2, 0>("half/half + consumer 8 waves", src, slice, n80, 20, 1, 4, grid, out);
    }
    (void)n64;
    return 0;
}
