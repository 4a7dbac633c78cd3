// Is the rate of a pitched (2-D) device -> page-locked-host copy a property of the STREAM it is issued on?  (Round 6: the host-operand pipeline's downloads ran at
// 55, 38 or 21 GB/s from one group to the next.)  16 streams, three copies of one 119 MB window (512-byte pieces at a 1 KiB pitch) on each, then 1-D copies of the
// same bytes; then the same with other streams created and destroyed in between.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/rect_lottery.hip -o /tmp/rect_lottery && /tmp/rect_lottery
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t N = 232965, PITCH = 1024, W = 512, BYTES = N * PITCH;
    char *c_pin, *dc;
    CHECK(hipHostMalloc((void **)&c_pin, BYTES, hipHostMallocDefault));
    memset(c_pin, 0, BYTES);
    CHECK(hipMalloc((void **)&dc, BYTES));
    CHECK(hipMemset(dc, 2, BYTES));
    CHECK(hipDeviceSynchronize());
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("# priority range: least %d greatest %d\n", least, greatest);
    for (int round = 0; round < 2; round++) {
        std::vector<hipStream_t> ss(16);
        for (int i = 0; i < 16; i++) {
            if (round == 0) CHECK(hipStreamCreateWithFlags(&ss[i], hipStreamNonBlocking));
            else CHECK(hipStreamCreateWithPriority(&ss[i], hipStreamNonBlocking, i % 3 == 0 ? greatest : i % 3 == 1 ? least : 0));
        }
        for (int i = 0; i < 16; i++) {
            double t2[3], t1 = 0;
            for (int r = 0; r < 3; r++) {
                const double t0 = now_ms();
                CHECK(hipMemcpy2DAsync(c_pin, PITCH, dc, PITCH, W, N, hipMemcpyDeviceToHost, ss[i]));
                CHECK(hipStreamSynchronize(ss[i]));
                t2[r] = now_ms() - t0;
            }
            {
                const double t0 = now_ms();
                CHECK(hipMemcpyAsync(c_pin, dc, N * W, hipMemcpyDeviceToHost, ss[i]));
                CHECK(hipStreamSynchronize(ss[i]));
                t1 = now_ms() - t0;
            }
            printf("round %d stream %2d: 2-D window down %6.3f %6.3f %6.3f ms (%5.1f GB/s)   1-D, same bytes %6.3f ms\n", round, i, t2[0], t2[1], t2[2], N * W / t2[2] * 1e-6, t1);
        }
        for (auto s : ss) CHECK(hipStreamDestroy(s));
    }
    // does it depend on WHERE the device buffer lives?  Churn the device heap (what bench.py's papers100M leg does: tens of GB allocated and freed), then a NEW source
    // buffer beside the one made at the start; and a new page-locked destination
    {
        std::vector<void *> big;
        for (int i = 0; i < 12; i++) { void *q = nullptr; if (hipMalloc(&q, (size_t)12 << 30) == hipSuccess) { CHECK(hipMemset(q, 1, (size_t)12 << 30)); big.push_back(q); } }
        std::vector<void *> small;
        for (int i = 0; i < 64; i++) { void *q = nullptr; if (hipMalloc(&q, (size_t)37 << 20) == hipSuccess) small.push_back(q); }
        CHECK(hipDeviceSynchronize());
        for (size_t i = 0; i < big.size(); i++) CHECK(hipFree(big[i]));
        for (size_t i = 0; i < small.size(); i += 2) CHECK(hipFree(small[i]));
        printf("# churned: %zu x 12 GB allocated, written and freed; every other of %zu x 37 MB freed\n", big.size(), small.size());
        char *dc2, *c_pin2;
        CHECK(hipMalloc((void **)&dc2, BYTES));
        CHECK(hipMemset(dc2, 3, BYTES));
        CHECK(hipHostMalloc((void **)&c_pin2, BYTES, hipHostMallocDefault));
        memset(c_pin2, 0, BYTES);
        hipStream_t s;
        CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const char *names[4] = {"old device buffer -> old host buffer", "old device buffer -> new host buffer", "new device buffer -> old host buffer", "new device buffer -> new host buffer"};
        for (int v = 0; v < 4; v++) {
            char *src = (v & 2) ? dc2 : dc, *dst = (v & 1) ? c_pin2 : c_pin;
            double t2[3];
            for (int r = 0; r < 3; r++) {
                const double t0 = now_ms();
                CHECK(hipMemcpy2DAsync(dst, PITCH, src, PITCH, W, N, hipMemcpyDeviceToHost, s));
                CHECK(hipStreamSynchronize(s));
                t2[r] = now_ms() - t0;
            }
            printf("after the churn, %s: %6.3f %6.3f %6.3f ms (%5.1f GB/s)\n", names[v], t2[0], t2[1], t2[2], N * W / t2[2] * 1e-6);
        }
    }
    return 0;
}
