// Can the reference's default call (CPU tensors in and out, spmm_test.py:29-35) hide the product behind the two PCIe copies?  (Round 6.)
// The upload of X and the download of C are 238.6 MB each for the Reddit shape (h = 256, FLT32).  PCIe is full duplex: with FEATURE WINDOWS
// (a window of X = `w` columns of every row: `w * 4` bytes at a pitch of 1 KiB) window k + 1 can go up while window k's product runs and
// window k - 1's result comes down -- and a feature window keeps every row's stored order of sums (bit-identical FLT32).
// What this measures, before the library is touched:
//   1. one-dimensional copies: pageable -> device, device -> pinned, alone and both at once (two streams, two host threads);
//   2. the same as 2-D copies of a window (width 256 / 512 bytes, pitch 1 KiB), from pageable and from page-locked memory;
//   3. hipHostRegister / hipHostUnregister of the caller's X;
//   4. a kernel that reads a window straight out of page-locked host memory (no DMA engine);
//   5. a host-side gather of a window into a pinned staging buffer (threads) + a contiguous copy.
//   hipcc --offload-arch=gfx950 -O2 -pthread scripts/micro/pcie_windows.hip -o /tmp/pcie_windows && /tmp/pcie_windows
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_window_from_host(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t rows, uint32_t pitch16, uint32_t w16, uint32_t off16) {
    // one 16-byte piece per thread; a row's window is w16 pieces
    const uint64_t n = (uint64_t)rows * w16;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = i / w16, c = i % w16;
        dst[(uint64_t)r * pitch16 + off16 + c] = src[(uint64_t)r * pitch16 + off16 + c];
    }
}


// the same copy with four independent 16-byte pieces in flight per thread (reads of host memory wait ~2 us each)
__global__ void k_window_x4(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint32_t rows, uint32_t pitch16, uint32_t w16, uint32_t off16) {
    const uint64_t n = (uint64_t)rows * w16, G = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += 4 * G) {
        uint4 v[4];
        uint64_t o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t j = i + u * G;
            const uint32_t r = j < n ? j / w16 : 0, c = j < n ? j % w16 : 0;
            o[u] = (uint64_t)r * pitch16 + off16 + c;
            if (j < n) v[u] = src[o[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i + u * G < n) dst[o[u]] = v[u];
    }
}

// a stand-in for the product of one feature window: `grid` workgroups of 512 threads that own a compute unit each (all of its LDS) for `ticks` of the 100 MHz clock
__global__ __launch_bounds__(512) void k_fake_product(uint64_t ticks, float *sink) {
    extern __shared__ char lds[];
    const uint64_t t0 = wall_clock64();
    float a = threadIdx.x;
    while (wall_clock64() - t0 < ticks) a = a * 1.0001f + 1.f;
    if (a == 12345.678f) sink[0] = a + lds[threadIdx.x];
}

template <class F> static double median_ms(int reps, F f) {
    std::vector<double> t;
    for (int i = 0; i < reps; i++) {
        const double t0 = now_ms();
        f();
        t.push_back(now_ms() - t0);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const size_t N = 232965, H = 256, ES = 4, PITCH = H * ES, BYTES = N * PITCH;
    char *x_page = (char *)aligned_alloc(4096, (BYTES + 4095) & ~(size_t)4095);
    memset(x_page, 1, BYTES);
    char *c_pin, *x_pin, *stage_pin;
    CHECK(hipHostMalloc((void **)&c_pin, BYTES, hipHostMallocDefault));
    CHECK(hipHostMalloc((void **)&x_pin, BYTES, hipHostMallocDefault));
    CHECK(hipHostMalloc((void **)&stage_pin, BYTES, hipHostMallocDefault));
    memset(c_pin, 0, BYTES);
    memset(x_pin, 1, BYTES);
    char *c_page = (char *)aligned_alloc(4096, (BYTES + 4095) & ~(size_t)4095);
    memset(c_page, 0, BYTES);
    char *dx, *dc;
    CHECK(hipMalloc((void **)&dx, BYTES));
    CHECK(hipMalloc((void **)&dc, BYTES));
    CHECK(hipMemset(dc, 2, BYTES));
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const int R = 7;
    auto gbs = [&](double ms, size_t b) { return b / ms * 1e-6; };

    // warm
    CHECK(hipMemcpyAsync(dx, x_page, BYTES, hipMemcpyHostToDevice, sa));
    CHECK(hipMemcpyAsync(c_pin, dc, BYTES, hipMemcpyDeviceToHost, sb));
    CHECK(hipDeviceSynchronize());

    double t;
    printf("# %zu rows x %zu bytes = %.1f MB each way\n", N, PITCH, BYTES * 1e-6);
    t = median_ms(R, [&] { CHECK(hipMemcpyAsync(dx, x_page, BYTES, hipMemcpyHostToDevice, sa)); CHECK(hipStreamSynchronize(sa)); });
    printf("1-D  pageable -> device                      %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
    t = median_ms(R, [&] { CHECK(hipMemcpyAsync(dx, x_pin, BYTES, hipMemcpyHostToDevice, sa)); CHECK(hipStreamSynchronize(sa)); });
    printf("1-D  pinned   -> device                      %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
    t = median_ms(R, [&] { CHECK(hipMemcpyAsync(c_pin, dc, BYTES, hipMemcpyDeviceToHost, sb)); CHECK(hipStreamSynchronize(sb)); });
    printf("1-D  device   -> pinned                      %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
    t = median_ms(3, [&] { CHECK(hipMemcpyAsync(c_page, dc, BYTES, hipMemcpyDeviceToHost, sb)); CHECK(hipStreamSynchronize(sb)); });
    printf("1-D  device   -> pageable                    %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
    // is the pageable async copy asynchronous for the host?
    {
        const double t0 = now_ms();
        CHECK(hipMemcpyAsync(dx, x_page, BYTES, hipMemcpyHostToDevice, sa));
        const double t1 = now_ms();
        CHECK(hipStreamSynchronize(sa));
        printf("     pageable -> device: the call returns after %.3f ms, the copy ends after %.3f ms\n", t1 - t0, now_ms() - t0);
    }
    // both directions at once: one host thread (pinned both ways)
    t = median_ms(R, [&] {
        CHECK(hipMemcpyAsync(dx, x_pin, BYTES, hipMemcpyHostToDevice, sa));
        CHECK(hipMemcpyAsync(c_pin, dc, BYTES, hipMemcpyDeviceToHost, sb));
        CHECK(hipStreamSynchronize(sa));
        CHECK(hipStreamSynchronize(sb));
    });
    printf("1-D  pinned -> device || device -> pinned    %7.3f ms  %6.1f GB/s each way\n", t, gbs(t, BYTES));
    // pageable up from a second thread, pinned down
    t = median_ms(R, [&] {
        std::thread up([&] { CHECK(hipMemcpyAsync(dx, x_page, BYTES, hipMemcpyHostToDevice, sa)); CHECK(hipStreamSynchronize(sa)); });
        CHECK(hipMemcpyAsync(c_pin, dc, BYTES, hipMemcpyDeviceToHost, sb));
        CHECK(hipStreamSynchronize(sb));
        up.join();
    });
    printf("1-D  pageable -> device (thread) || -> pinned %6.3f ms  %6.1f GB/s each way\n", t, gbs(t, BYTES));

    for (size_t w : {(size_t)256, (size_t)512}) {
        const size_t nwin = PITCH / w;
        printf("# windows of %zu bytes at a pitch of %zu (%zu windows)\n", w, PITCH, nwin);
        t = median_ms(R, [&] {
            for (size_t k = 0; k < nwin; k++) CHECK(hipMemcpy2DAsync(dx + k * w, PITCH, x_page + k * w, PITCH, w, N, hipMemcpyHostToDevice, sa));
            CHECK(hipStreamSynchronize(sa));
        });
        printf("2-D  pageable -> device, all windows         %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
        t = median_ms(R, [&] {
            for (size_t k = 0; k < nwin; k++) CHECK(hipMemcpy2DAsync(dx + k * w, PITCH, x_pin + k * w, PITCH, w, N, hipMemcpyHostToDevice, sa));
            CHECK(hipStreamSynchronize(sa));
        });
        printf("2-D  pinned   -> device, all windows         %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
        t = median_ms(R, [&] {
            for (size_t k = 0; k < nwin; k++) CHECK(hipMemcpy2DAsync(c_pin + k * w, PITCH, dc + k * w, PITCH, w, N, hipMemcpyDeviceToHost, sb));
            CHECK(hipStreamSynchronize(sb));
        });
        printf("2-D  device   -> pinned, all windows         %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
        t = median_ms(R, [&] {
            for (size_t k = 0; k < nwin; k++) {
                CHECK(hipMemcpy2DAsync(dx + k * w, PITCH, x_pin + k * w, PITCH, w, N, hipMemcpyHostToDevice, sa));
                CHECK(hipMemcpy2DAsync(c_pin + k * w, PITCH, dc + k * w, PITCH, w, N, hipMemcpyDeviceToHost, sb));
            }
            CHECK(hipStreamSynchronize(sa));
            CHECK(hipStreamSynchronize(sb));
        });
        printf("2-D  pinned -> device || device -> pinned    %7.3f ms  %6.1f GB/s each way\n", t, gbs(t, BYTES));
        // kernel copies through mapped page-locked memory
        t = median_ms(R, [&] {
            for (size_t k = 0; k < nwin; k++)
                hipLaunchKernelGGL(k_window_from_host, dim3(1024), dim3(256), 0, sa, (const uint4 *)x_pin, (uint4 *)dx, (uint32_t)N, (uint32_t)(PITCH / 16), (uint32_t)(w / 16), (uint32_t)(k * w / 16));
            CHECK(hipStreamSynchronize(sa));
        });
        printf("kern pinned   -> device, all windows         %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
        t = median_ms(R, [&] {
            for (size_t k = 0; k < nwin; k++)
                hipLaunchKernelGGL(k_window_from_host, dim3(1024), dim3(256), 0, sb, (const uint4 *)dc, (uint4 *)c_pin, (uint32_t)N, (uint32_t)(PITCH / 16), (uint32_t)(w / 16), (uint32_t)(k * w / 16));
            CHECK(hipStreamSynchronize(sb));
        });
        printf("kern device   -> pinned, all windows         %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
        t = median_ms(R, [&] {
            for (size_t k = 0; k < nwin; k++) {
                hipLaunchKernelGGL(k_window_from_host, dim3(512), dim3(256), 0, sa, (const uint4 *)x_pin, (uint4 *)dx, (uint32_t)N, (uint32_t)(PITCH / 16), (uint32_t)(w / 16), (uint32_t)(k * w / 16));
                hipLaunchKernelGGL(k_window_from_host, dim3(512), dim3(256), 0, sb, (const uint4 *)dc, (uint4 *)c_pin, (uint32_t)N, (uint32_t)(PITCH / 16), (uint32_t)(w / 16), (uint32_t)(k * w / 16));
            }
            CHECK(hipStreamSynchronize(sa));
            CHECK(hipStreamSynchronize(sb));
        });
        printf("kern pinned -> device || device -> pinned    %7.3f ms  %6.1f GB/s each way\n", t, gbs(t, BYTES));
        // host gather of a window into pinned staging with T threads, then a contiguous copy
        for (int T : {4, 8}) {
            t = median_ms(R, [&] {
                for (size_t k = 0; k < nwin; k++) {
                    std::vector<std::thread> th;
                    char *dst = stage_pin + k * (N * w);
                    for (int q = 0; q < T; q++)
                        th.emplace_back([&, q] {
                            const size_t r0 = N * q / T, r1 = N * (q + 1) / T;
                            for (size_t r = r0; r < r1; r++) memcpy(dst + r * w, x_page + r * PITCH + k * w, w);
                        });
                    for (auto &q : th) q.join();
                    CHECK(hipMemcpyAsync(dx + k * (N * w), dst, N * w, hipMemcpyHostToDevice, sa));
                }
                CHECK(hipStreamSynchronize(sa));
            });
            printf("host gather (%d threads) + 1-D copy, all windows %5.3f ms  %6.1f GB/s\n", T, t, gbs(t, BYTES));
        }
    }

    // ---- the whole pipeline with a stand-in product: window k + 1 goes up while window k's product runs and window k - 1's result comes down ----
    {
        CHECK(hipFuncSetAttribute((const void *)k_fake_product, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipStream_t sc;
        CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
        hipEvent_t ein[8], eprod[8];
        for (int k = 0; k < 8; k++) { CHECK(hipEventCreateWithFlags(&ein[k], hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&eprod[k], hipEventDisableTiming)); }
        float *sink;
        CHECK(hipMalloc((void **)&sink, 256));
        CHECK(hipHostRegister(x_page, BYTES, hipHostRegisterDefault));
        char *x_reg_dev = nullptr;
        CHECK(hipHostGetDevicePointer((void **)&x_reg_dev, x_page, 0));
        struct V { const char *name; size_t w; int up; int down; unsigned grid; double prod_ms; unsigned cgrid; };
        // up: 0 = hipMemcpy2DAsync from pageable (registered here, still the runtime's path), 1 = kernel from registered memory, 2 = hipMemcpy2DAsync from pinned
        // down: 0 = hipMemcpy2DAsync to pinned, 1 = kernel to pinned
        const V vs[] = {
            {"4 x 256 B: 2-D copy up, kernel down (64 wg)", 256, 0, 1, 128, 1.0, 64},
            {"4 x 256 B: kernel up, kernel down (64 wg)", 256, 1, 1, 128, 1.0, 64},
            {"4 x 256 B: kernel up, kernel down (16 wg)", 256, 1, 1, 128, 1.0, 16},
            {"4 x 256 B: kernel up, kernel down (128 wg)", 256, 1, 1, 128, 1.0, 128},
            {"4 x 256 B: pinned 2-D copy up, kernel down (64 wg)", 256, 2, 1, 128, 1.0, 64},
            {"2 x 512 B: 2-D copy up, kernel down (64 wg), product on 256 CUs", 512, 0, 1, 256, 1.07, 64},
            {"2 x 512 B: 2-D copy up, kernel down (8 wg), product on 256 CUs", 512, 0, 1, 256, 1.07, 8},
            {"2 x 512 B: 2-D copy up, 2-D copy down, product on 256 CUs", 512, 0, 0, 256, 1.07, 0},
            {"4 x 256 B: 2-D copy up, 2-D copy down", 256, 0, 0, 128, 1.0, 0},
            {"4 x 256 B: x4 kernel up, x4 kernel down (32 wg)", 256, 3, 3, 128, 1.0, 32},
            {"4 x 256 B: x4 kernel up, x4 kernel down (64 wg)", 256, 3, 3, 128, 1.0, 64},
            {"4 x 256 B: x4 kernel up, x4 kernel down (128 wg)", 256, 3, 3, 128, 1.0, 128},
            {"4 x 256 B: x4 kernel up, x4 kernel down (256 wg)", 256, 3, 3, 128, 1.0, 256},
            {"4 x 256 B: 2-D copy up, x4 kernel down (128 wg)", 256, 0, 3, 128, 1.0, 128},
            {"4 x 256 B: 2-D copy up, kernel down (128 wg)", 256, 0, 1, 128, 1.0, 128},
            {"4 x 256 B: 2-D copy up, kernel down (256 wg)", 256, 0, 1, 128, 1.0, 256},
            {"2 x 512 B: 2-D copy up, x4 kernel down (64 wg), product on 256 CUs", 512, 0, 3, 256, 1.07, 64},
            {"2 x 512 B: 2-D copy up, x4 kernel down (128 wg), product on 256 CUs", 512, 0, 3, 256, 1.07, 128},
            {"8 x 128 B: x4 kernel up, x4 kernel down (128 wg), product 0.5 ms", 128, 3, 3, 128, 0.5, 128},
            {"4 x 256 B: x4 kernel both (128 wg), NO product", 256, 3, 3, 1, 0.001, 128},
            {"4 x 256 B: 2-D copy up, x4 kernel down (128 wg), NO product", 256, 0, 3, 1, 0.001, 128},
        };
        for (const V &v : vs) {
            const size_t nwin = PITCH / v.w;
            const char *xsrc = v.up == 2 ? x_pin : x_page;
            t = median_ms(R, [&] {
                for (size_t k = 0; k < nwin; k++) {
                    if (v.up == 3)
                        hipLaunchKernelGGL(k_window_x4, dim3(v.cgrid), dim3(256), 0, sa, (const uint4 *)x_reg_dev, (uint4 *)dx, (uint32_t)N, (uint32_t)(PITCH / 16), (uint32_t)(v.w / 16), (uint32_t)(k * v.w / 16));
                    else if (v.up == 1)
                        hipLaunchKernelGGL(k_window_from_host, dim3(v.cgrid), dim3(256), 0, sa, (const uint4 *)x_reg_dev, (uint4 *)dx, (uint32_t)N, (uint32_t)(PITCH / 16), (uint32_t)(v.w / 16), (uint32_t)(k * v.w / 16));
                    else
                        CHECK(hipMemcpy2DAsync(dx + k * v.w, PITCH, xsrc + k * v.w, PITCH, v.w, N, hipMemcpyHostToDevice, sa));
                    CHECK(hipEventRecord(ein[k], sa));
                    CHECK(hipStreamWaitEvent(sc, ein[k], 0));
                    hipLaunchKernelGGL(k_fake_product, dim3(v.grid), dim3(512), 160 * 1024, sc, (uint64_t)(v.prod_ms * 1e5), sink);
                    CHECK(hipEventRecord(eprod[k], sc));
                    CHECK(hipStreamWaitEvent(sb, eprod[k], 0));
                    if (v.down == 3)
                        hipLaunchKernelGGL(k_window_x4, dim3(v.cgrid), dim3(256), 0, sb, (const uint4 *)dc, (uint4 *)c_pin, (uint32_t)N, (uint32_t)(PITCH / 16), (uint32_t)(v.w / 16), (uint32_t)(k * v.w / 16));
                    else if (v.down == 1)
                        hipLaunchKernelGGL(k_window_from_host, dim3(v.cgrid), dim3(256), 0, sb, (const uint4 *)dc, (uint4 *)c_pin, (uint32_t)N, (uint32_t)(PITCH / 16), (uint32_t)(v.w / 16), (uint32_t)(k * v.w / 16));
                    else
                        CHECK(hipMemcpy2DAsync(c_pin + k * v.w, PITCH, dc + k * v.w, PITCH, v.w, N, hipMemcpyDeviceToHost, sb));
                }
                CHECK(hipStreamSynchronize(sb));
                CHECK(hipStreamSynchronize(sa));
                CHECK(hipStreamSynchronize(sc));
            });
            printf("pipeline  %-66s %7.3f ms\n", v.name, t);
        }
        // the serial call of today: up, product (two rounds of 256 workgroups ~ 2.05 ms), down
        t = median_ms(R, [&] {
            CHECK(hipMemcpyAsync(dx, x_page, BYTES, hipMemcpyHostToDevice, sc));
            hipLaunchKernelGGL(k_fake_product, dim3(256), dim3(512), 160 * 1024, sc, (uint64_t)(2.05 * 1e5), sink);
            CHECK(hipMemcpyAsync(c_pin, dc, BYTES, hipMemcpyDeviceToHost, sc));
            CHECK(hipStreamSynchronize(sc));
        });
        printf("serial    up, product of 2.05 ms, down                                        %7.3f ms\n", t);
        CHECK(hipHostUnregister(x_page));
    }
    // a FRESH pageable buffer: what page-locking it costs the first time, and reading it from a kernel
    {
        char *fresh = (char *)aligned_alloc(4096, (BYTES + 4095) & ~(size_t)4095);
        memset(fresh, 3, BYTES);
        double t0 = now_ms();
        CHECK(hipHostRegister(fresh, BYTES, hipHostRegisterDefault));
        const double treg = now_ms() - t0;
        char *fd = nullptr;
        CHECK(hipHostGetDevicePointer((void **)&fd, fresh, 0));
        t = median_ms(R, [&] {
            for (size_t k = 0; k < 4; k++)
                hipLaunchKernelGGL(k_window_from_host, dim3(64), dim3(256), 0, sa, (const uint4 *)fd, (uint4 *)dx, (uint32_t)N, (uint32_t)(PITCH / 16), 16u, (uint32_t)(k * 16));
            CHECK(hipStreamSynchronize(sa));
        });
        t0 = now_ms();
        CHECK(hipHostUnregister(fresh));
        const double tun = now_ms() - t0;
        printf("fresh buffer: hipHostRegister %.3f ms, kernel read of 4 windows (64 wg) %.3f ms, hipHostUnregister %.3f ms\n", treg, t, tun);
        char *fresh2 = (char *)aligned_alloc(4096, (BYTES + 4095) & ~(size_t)4095);
        memset(fresh2, 3, BYTES);
        t0 = now_ms();
        CHECK(hipMemcpyAsync(dx, fresh2, BYTES, hipMemcpyHostToDevice, sa));
        CHECK(hipStreamSynchronize(sa));
        printf("fresh buffer: first 1-D pageable -> device %.3f ms\n", now_ms() - t0);
        t0 = now_ms();
        CHECK(hipMemcpy2DAsync(dx, PITCH, fresh2, PITCH, 256, N, hipMemcpyHostToDevice, sa));
        CHECK(hipStreamSynchronize(sa));
        printf("fresh buffer: then a 2-D window of 256 B     %.3f ms (a quarter of the bytes)\n", now_ms() - t0);
    }
    // page-locking the caller's buffer
    {
        std::vector<double> reg, unreg;
        for (int i = 0; i < 5; i++) {
            double t0 = now_ms();
            CHECK(hipHostRegister(x_page, BYTES, hipHostRegisterDefault));
            reg.push_back(now_ms() - t0);
            if (i == 4) {
                t = median_ms(R, [&] { CHECK(hipMemcpyAsync(dx, x_page, BYTES, hipMemcpyHostToDevice, sa)); CHECK(hipStreamSynchronize(sa)); });
                printf("1-D  registered -> device                    %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
                t = median_ms(R, [&] {
                    for (size_t k = 0; k < 2; k++) CHECK(hipMemcpy2DAsync(dx + k * 512, PITCH, x_page + k * 512, PITCH, 512, N, hipMemcpyHostToDevice, sa));
                    CHECK(hipStreamSynchronize(sa));
                });
                printf("2-D  registered -> device, 2 windows         %7.3f ms  %6.1f GB/s\n", t, gbs(t, BYTES));
            }
            t0 = now_ms();
            CHECK(hipHostUnregister(x_page));
            unreg.push_back(now_ms() - t0);
        }
        std::sort(reg.begin(), reg.end());
        std::sort(unreg.begin(), unreg.end());
        printf("hipHostRegister of %.1f MB: %.3f ms (median of 5), hipHostUnregister %.3f ms\n", BYTES * 1e-6, reg[2], unreg[2]);
    }
    return 0;
}
