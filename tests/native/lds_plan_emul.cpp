// Test infrastructure: a CPU emulator of k_lds_spmm's token walk over the schedule lds_plan.hpp builds.
// It reads the plan exactly as the kernel does (tile table, chunk lists, batch counts, token streams, row map)
// so that tests/test_lds_plan.py can compare "what the schedule says" with the oracle on the CPU.
#include "../../pygim_amd/csrc/lds_plan.hpp"

#include <cstring>

using namespace pygim;

template <typename T>
static int emulate(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const T *X, uint32_t h, T *C,
                   uint32_t KA, uint32_t batch, uint32_t threads, uint64_t *stats, uint32_t nw, const T *vals) {
    LdsGeometry geo;
    geo.NW = nw;
    geo.KA = KA;
    geo.BATCH = batch;
    LdsPlanHost plan;
    static_assert(sizeof(T) == 4, "4-byte values");
    lds_plan_build(rowptr, col, nrows, ncols, geo, plan, threads, reinterpret_cast<const uint32_t *>(vals));
    if (plan.header_overflow) return 13;
    if ((vals != nullptr) != !plan.wts.empty() || (vals && plan.wts.size() != plan.tok.size())) return 9;
    const uint32_t NW = geo.NW, KC = geo.KC;
    const uint32_t nslices = (h + 63) / 64;
    if (stats) {
        stats[0] = plan.ntiles;
        stats[1] = plan.slots;
        stats[2] = plan.ntokens;
        stats[3] = plan.tok.size();
    }
    std::vector<char> written((size_t)nrows, 0);
    for (uint32_t ti = 0; ti < plan.ntiles; ti++) {
        const LdsTile &t = plan.tiles[ti];
        for (uint32_t s = 0; s < nslices; s++) {
            const uint32_t wvalid = std::min(64u, h - s * 64);
            for (uint32_t w = 0; w < NW; w++) {
                std::vector<T> acc((size_t)(KA + 1) * 64, T(0));
                uint64_t at = (uint64_t)t.tokstart[w] * batch;
                uint32_t chunk = t.chunk0, chunk_next = 0;
                for (uint32_t j = 0; j < t.nch; j++, chunk = chunk_next) {
                    // the kernel's scalar side reads nothing but the token stream: batch count / next count / next chunk id
                    // sit in the upper halves of the first three tokens of the slot's first batch
                    if (at + 2 >= plan.tok.size()) return 2;
                    const uint32_t nb = plan.tok[at] >> LDS_HDR_SHIFT, nb_next = plan.tok[at + 1] >> LDS_HDR_SHIFT;
                    chunk_next = plan.tok[at + 2] >> LDS_HDR_SHIFT;
                    if (nb == 0 || nb != plan.nb[t.nb_off + (size_t)j * NW + w] || chunk != plan.chunks[t.chunk_off + j]) return 11;
                    if (nb_next != (j + 1 < t.nch ? plan.nb[t.nb_off + (size_t)(j + 1) * NW + w] : 0u)) return 12;
                    for (uint32_t b = 0; b < nb * batch; b++, at++) {
                        if (at >= plan.tok.size()) return 2;
                        const uint32_t tk = plan.tok[at];
                        const uint32_t k = tk & 0xFF, ldsrow = (tk >> 8) & 0x3FF;
                        if ((tk >> LDS_HDR_SHIFT) && b > 2) return 3;   // bits 18.. are header space in tokens 0..2 of the first batch only
                        // the kernel's LDS ring: slot j streams its chunk into buffer j & 1 = rows [KC * (j & 1), KC * (j & 1) + KC);
                        // a real token must point into THAT buffer (padding reads row 0 of buffer 0 into the dummy accumulator)
                        if (k < KA && ldsrow / KC != (j & 1)) return 15;
                        if (ldsrow >= 2 * KC) return 16;
                        const uint32_t c = ldsrow % KC;
                        if (k > KA) return 4;
                        const uint64_t xr = (uint64_t)chunk * KC + c;
                        if (k < KA && xr >= ncols) return 5;
                        T wv = T(1);
                        if (vals) std::memcpy(&wv, &plan.wts[at], 4);
                        if (vals && k == KA && plan.wts[at] != 0) return 10;  // padding carries the value 0
                        for (uint32_t l = 0; l < wvalid; l++) {
                            T x = xr < ncols ? X[xr * h + s * 64 + l] : T(0);
                            T &a = acc[(size_t)k * 64 + l];
                            if constexpr (std::is_integral<T>::value) {
                                if (vals) x = (T)((uint32_t)wv * (uint32_t)x);
                                a = (T)((uint32_t)a + (uint32_t)x);
                            } else {
                                if (vals) {
                                    volatile T prod = wv * x;  // product and sum round separately (no FMA), as the kernel and the oracle do
                                    x = prod;
                                }
                                a = a + x;
                            }
                        }
                    }
                }
                if (plan.nb[t.nb_off + (size_t)t.nch * NW + w] != 0) return 6;  // closing row of zeros
                for (uint32_t k = 0; k < KA; k++) {
                    const uint32_t row = plan.rowmap[((size_t)ti * NW + w) * KA + k];
                    if (row == 0xFFFFFFFFu) continue;
                    if (row >= nrows) return 7;
                    if (s == 0) written[row]++;
                    for (uint32_t l = 0; l < wvalid; l++) C[(size_t)row * h + s * 64 + l] = acc[(size_t)k * 64 + l];
                }
            }
        }
    }
    for (uint32_t r = 0; r < nrows; r++)
        if (written[r] != 1) return 8;
    return 0;
}

extern "C" {
int lds_emul_f32(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const float *X, uint32_t h, float *C,
                 uint32_t KA, uint32_t batch, uint32_t threads, uint64_t *stats, uint32_t nw, const float *vals) {
    return emulate<float>(rowptr, col, nrows, ncols, X, h, C, KA, batch, threads, stats, nw, vals);
}
int lds_emul_i32(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const int32_t *X, uint32_t h, int32_t *C,
                 uint32_t KA, uint32_t batch, uint32_t threads, uint64_t *stats, uint32_t nw, const int32_t *vals) {
    return emulate<int32_t>(rowptr, col, nrows, ncols, X, h, C, KA, batch, threads, stats, nw, vals);
}
}

extern "C" uint32_t lds_emul_rows_per_tile(uint32_t nrows, uint32_t rmax, uint32_t nslices, uint32_t cus) {
    return lds_rows_per_tile(nrows, rmax, nslices, cus);
}
