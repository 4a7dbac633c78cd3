// Test infrastructure: the host-side schedule builder and gfx950 instruction encoder (pygim_amd/csrc/lds_plan.hpp) under
// AddressSanitizer + UndefinedBehaviorSanitizer.  What the encoder writes ends up in EXECUTABLE GPU memory: an out-of-bounds index or a
// mis-packed instruction word there is garbage the GPU would run, so the same plans the parity tests use are built, compiled into code
// streams and interpreted here with every access checked (no GPU sanitizer exists on this pool).
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -pthread tests/native/lds_plan_san_main.cpp -o /tmp/lds_san && /tmp/lds_san [cases]
#include "lds_plan_emul.cpp"

#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>

struct Csr {
    std::vector<uint32_t> rowptr, col;
};

static Csr random_csr(std::mt19937_64 &rng, uint32_t nrows, uint32_t ncols, double avg_deg, double empty_frac, uint32_t long_row, uint32_t long_len) {
    Csr m;
    m.rowptr.assign(nrows + 1, 0);
    std::vector<uint8_t> mark(ncols);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (uint32_t r = 0; r < nrows; r++) {
        uint32_t deg = 0;
        if (r == long_row && long_len) deg = std::min(long_len, ncols);
        else if (u(rng) >= empty_frac) deg = (uint32_t)std::min<double>(ncols, std::max(0.0, -avg_deg * std::log(1.0 - u(rng) * 0.999)));
        std::fill(mark.begin(), mark.end(), 0);
        uint32_t got = 0;
        if (deg * 2 > ncols) {   // dense row: choose what to leave out
            std::fill(mark.begin(), mark.end(), 1);
            got = ncols;
            while (got > deg) {
                const uint32_t c = (uint32_t)(rng() % ncols);
                if (mark[c]) { mark[c] = 0; got--; }
            }
        } else {
            while (got < deg) {
                const uint32_t c = (uint32_t)(rng() % ncols);
                if (!mark[c]) { mark[c] = 1; got++; }
            }
        }
        for (uint32_t c = 0; c < ncols; c++)
            if (mark[c]) m.col.push_back(c);
        m.rowptr[r + 1] = (uint32_t)m.col.size();
    }
    return m;
}

template <typename T> static std::vector<T> reference(const Csr &m, uint32_t nrows, const std::vector<T> &x, uint32_t h, const float *vals) {
    std::vector<T> c((size_t)nrows * h, T(0));
    for (uint32_t r = 0; r < nrows; r++)
        for (uint32_t e = m.rowptr[r]; e < m.rowptr[r + 1]; e++)
            for (uint32_t f = 0; f < h; f++) {
                T xv = x[(size_t)m.col[e] * h + f];
                T &a = c[(size_t)r * h + f];
                if constexpr (std::is_integral<T>::value) a = (T)((typename std::make_unsigned<T>::type)a + (typename std::make_unsigned<T>::type)xv);
                else {
                    if (vals) {
                        volatile float prod = vals[e] * xv;
                        xv = prod;
                    }
                    a = a + xv;
                }
            }
    return c;
}

int main(int argc, char **argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 40;
    std::mt19937_64 rng(12345);
    // (waves, columns per chunk, ring buffers, staged columns per group, x-register sets)
    const uint32_t geos[][5] = {{16, 320, 2, 8, 2}, {16, 192, 3, 8, 2}, {16, 192, 3, 6, 3}, {8, 320, 2, 10, 2}, {8, 192, 3, 10, 2}, {8, 160, 4, 10, 2},
                                {8, 128, 5, 10, 2}, {8, 128, 5, 6, 3}, {8, 96, 6, 10, 2}, {8, 64, 8, 2, 2}, {16, 128, 4, 12, 2}, {8, 32, 5, 4, 3}};
    int done = 0;
    for (int c = 0; c < cases; c++) {
        const uint32_t nrows = 1 + (uint32_t)(rng() % 3000), ncols = 1 + (uint32_t)(rng() % 4000), h = 1 + (uint32_t)(rng() % 200);
        const double deg = 1 + (double)(rng() % 60);
        const Csr m = random_csr(rng, nrows, ncols, deg, (rng() % 4) * 0.2, (uint32_t)(rng() % nrows), (rng() % 3) ? 0 : (uint32_t)(rng() % 3000));
        const auto &g = geos[rng() % (sizeof geos / sizeof geos[0])];
        const uint32_t splits = (rng() % 5 == 0) ? 2 + (uint32_t)(rng() % 3) : 1;
        const uint32_t rpt = (rng() % 4 == 0) ? 16 + (uint32_t)(rng() % 1500) : 0;
        const uint32_t bnd = (uint32_t)(rng() % 2);   // rings of >= 3 buffers: meet in the middle of a slot, or at its boundary
        uint64_t stats[4];
        if (c % 5 == 4) {   // the 8-byte element form: 512-byte rows, register pairs (8 waves x 114 rows)
            const uint32_t g8[][4] = {{64, 5, 5, 2}, {48, 6, 3, 3}, {96, 3, 2, 2}, {16, 5, 5, 2}, {32, 10, 4, 2}};
            const auto &q = g8[rng() % 5];
            if (c % 2 && splits == 1 && rng() % 2 == 0) {   // valued DBL64 (round 5): real-valued weights and features, bit-identical to the sequential loop
                std::vector<double> x((size_t)ncols * h), out((size_t)nrows * h, 77.0), vals(m.col.size());
                std::uniform_real_distribution<double> ud(-1.0, 1.0);
                for (auto &v : x) v = ud(rng);
                for (auto &v : vals) v = ud(rng);
                const int rc = lds_code_f64_val_geo(m.rowptr.data(), m.col.data(), nrows, ncols, x.data(), h, out.data(), 3, stats, q[0], q[1], vals.data(), q[2], q[3], rpt, bnd);
                if (rc) { printf("case %d: interpreter code %d (valued f64 geo %u %u %u %u)\n", c, rc, q[0], q[1], q[2], q[3]); return 1; }
                std::vector<double> want((size_t)nrows * h, 0.0);
                for (uint32_t r = 0; r < nrows; r++)
                    for (uint32_t e = m.rowptr[r]; e < m.rowptr[r + 1]; e++)
                        for (uint32_t f = 0; f < h; f++) {
                            volatile double prod = vals[e] * x[(size_t)m.col[e] * h + f];
                            want[(size_t)r * h + f] = want[(size_t)r * h + f] + prod;
                        }
                if (memcmp(out.data(), want.data(), want.size() * 8) != 0) { printf("case %d: valued f64 result differs\n", c); return 2; }
            } else if (c % 2) {
                std::vector<double> x((size_t)ncols * h), out((size_t)nrows * h, 77.0);
                for (auto &v : x) v = (double)((int64_t)(rng() % 17) - 8);
                const int rc = lds_code_f64_geo(m.rowptr.data(), m.col.data(), nrows, ncols, x.data(), h, out.data(), 3, stats, q[0], q[1], splits, q[2], q[3], rpt, bnd);
                if (rc) { printf("case %d: interpreter code %d (f64 geo %u %u %u %u)\n", c, rc, q[0], q[1], q[2], q[3]); return 1; }
                if (out != reference<double>(m, nrows, x, h, nullptr)) { printf("case %d: f64 result differs\n", c); return 2; }
            } else if (splits == 1 && rng() % 2 == 0) {   // valued INT64 (round 5): inline values, values that fit int32 (through s94), any 64 bits (s[94:95])
                std::vector<int64_t> x((size_t)ncols * h), out((size_t)nrows * h, 77), vals(m.col.size());
                for (auto &v : x) v = (int64_t)rng();
                const uint32_t cls = (uint32_t)(rng() % 3);   // every value inline / every value fits int32 / any 64 bits (both halves through s[94:95])
                for (auto &v : vals) v = cls == 0 ? (int64_t)(rng() % 81) - 16 : cls == 1 ? (int64_t)(int32_t)rng() : (rng() % 4 ? (int64_t)rng() : (int64_t)(int32_t)rng());
                const int rc = lds_code_i64_val_geo(m.rowptr.data(), m.col.data(), nrows, ncols, x.data(), h, out.data(), 3, stats, q[0], q[1], vals.data(), q[2], q[3], rpt, bnd);
                if (rc) { printf("case %d: interpreter code %d (valued i64 geo %u %u %u %u)\n", c, rc, q[0], q[1], q[2], q[3]); return 1; }
                std::vector<int64_t> want((size_t)nrows * h, 0);
                for (uint32_t r = 0; r < nrows; r++)
                    for (uint32_t e = m.rowptr[r]; e < m.rowptr[r + 1]; e++)
                        for (uint32_t f = 0; f < h; f++)
                            want[(size_t)r * h + f] = (int64_t)((uint64_t)want[(size_t)r * h + f] + (uint64_t)vals[e] * (uint64_t)x[(size_t)m.col[e] * h + f]);
                if (out != want) { printf("case %d: valued i64 result differs\n", c); return 2; }
            } else {
                std::vector<int64_t> x((size_t)ncols * h), out((size_t)nrows * h, 77);
                for (auto &v : x) v = (int64_t)rng();
                const int rc = lds_code_i64_geo(m.rowptr.data(), m.col.data(), nrows, ncols, x.data(), h, out.data(), 3, stats, q[0], q[1], splits, q[2], q[3], rpt, bnd);
                if (rc) { printf("case %d: interpreter code %d (i64 geo %u %u %u %u)\n", c, rc, q[0], q[1], q[2], q[3]); return 1; }
                if (out != reference<int64_t>(m, nrows, x, h, nullptr)) { printf("case %d: i64 result differs\n", c); return 2; }
            }
            done++;
            continue;
        }
        if (c % 2 == 0) {
            std::vector<float> x((size_t)ncols * h), out((size_t)nrows * h, 77.f), vals;
            for (auto &v : x) v = (float)((int64_t)(rng() % 17) - 8);   // small integers: sums exact, so split plans compare too
            const bool valued = splits == 1 && rng() % 3 == 0;
            if (valued) {
                vals.resize(m.col.size());
                for (auto &v : vals) v = (float)((int64_t)(rng() % 5) - 2);
            }
            const int rc = lds_code_f32_geo(m.rowptr.data(), m.col.data(), nrows, ncols, x.data(), h, out.data(), 3, stats, g[1], g[2], valued ? vals.data() : nullptr,
                                            splits, g[0], g[3], g[4], rpt, bnd);
            if (rc) { printf("case %d: interpreter code %d (geo %u %u %u %u %u, %u x %u, h %u)\n", c, rc, g[0], g[1], g[2], g[3], g[4], nrows, ncols, h); return 1; }
            if (out != reference<float>(m, nrows, x, h, valued ? vals.data() : nullptr)) { printf("case %d: result differs\n", c); return 2; }
            std::vector<float> out2((size_t)nrows * h, 77.f);
            const int rc2 = lds_emul_f32(m.rowptr.data(), m.col.data(), nrows, ncols, x.data(), h, out2.data(), g[0] == 8 ? 192 : 96, g[0] == 8 ? 16 : 8, 3, stats, g[0],
                                         valued ? vals.data() : nullptr, splits);
            if (rc2 || out2 != out) { printf("case %d: token walk code %d / differs\n", c, rc2); return 3; }
        } else if (g[0] == 8 && rng() % 3 == 0) {   // INT16 (two features to a lane), unit weights or valued (round 5: v_pk_mul_lo_u16, the value inline or through s94)
            std::vector<int16_t> x((size_t)ncols * h), out((size_t)nrows * h, 77), vals;
            for (auto &v : x) v = (int16_t)rng();
            const bool valued = splits == 1 && (rng() & 1);
            if (valued) {
                vals.resize(m.col.size());
                const bool small = rng() & 1;
                for (auto &v : vals) v = small ? (int16_t)((int32_t)(rng() % 81) - 16) : (int16_t)rng();
            }
            const int rc = lds_code_i16_geo(m.rowptr.data(), m.col.data(), nrows, ncols, x.data(), h, out.data(), 3, stats, g[1], g[2], valued ? vals.data() : nullptr, splits, g[3], g[4], rpt, bnd);
            if (rc) { printf("case %d: interpreter code %d (i16 valued %d, geo %u %u %u %u %u)\n", c, rc, (int)valued, g[0], g[1], g[2], g[3], g[4]); return 1; }
            std::vector<int16_t> want((size_t)nrows * h, 0);
            for (uint32_t r = 0; r < nrows; r++)
                for (uint32_t e = m.rowptr[r]; e < m.rowptr[r + 1]; e++)
                    for (uint32_t f = 0; f < h; f++)
                        want[(size_t)r * h + f] = (int16_t)((uint32_t)want[(size_t)r * h + f] + (valued ? (uint32_t)(int32_t)vals[e] : 1u) * (uint32_t)(int32_t)x[(size_t)m.col[e] * h + f]);
            if (out != want) { printf("case %d: i16 result differs (valued %d)\n", c, (int)valued); return 2; }
        } else if (splits == 1 && rng() % 3 == 0) {   // valued INT32 (round 5): small values ride as inline constants, any others through an SGPR
            std::vector<int32_t> x((size_t)ncols * h), out((size_t)nrows * h, 77), vals(m.col.size());
            for (auto &v : x) v = (int32_t)rng();
            const bool small = rng() & 1;
            for (auto &v : vals) v = small ? (int32_t)(rng() % 81) - 16 : (int32_t)rng();
            const int rc = lds_code_i32_val_geo(m.rowptr.data(), m.col.data(), nrows, ncols, x.data(), h, out.data(), 3, stats, g[1], g[2], vals.data(), g[0], g[3], g[4], rpt, bnd);
            if (rc) { printf("case %d: interpreter code %d (valued i32, geo %u %u %u %u %u)\n", c, rc, g[0], g[1], g[2], g[3], g[4]); return 1; }
            std::vector<int32_t> want((size_t)nrows * h, 0);
            for (uint32_t r = 0; r < nrows; r++)
                for (uint32_t e = m.rowptr[r]; e < m.rowptr[r + 1]; e++)
                    for (uint32_t f = 0; f < h; f++)
                        want[(size_t)r * h + f] = (int32_t)((uint32_t)want[(size_t)r * h + f] + (uint32_t)vals[e] * (uint32_t)x[(size_t)m.col[e] * h + f]);
            if (out != want) { printf("case %d: valued i32 result differs\n", c); return 2; }
        } else {
            std::vector<int32_t> x((size_t)ncols * h), out((size_t)nrows * h, 77);
            for (auto &v : x) v = (int32_t)rng();
            const int rc = lds_code_i32_geo(m.rowptr.data(), m.col.data(), nrows, ncols, x.data(), h, out.data(), 3, stats, g[1], g[2], splits, g[0], g[3], g[4], rpt, bnd);
            if (rc) { printf("case %d: interpreter code %d (geo %u %u %u %u %u, %u x %u, h %u)\n", c, rc, g[0], g[1], g[2], g[3], g[4], nrows, ncols, h); return 1; }
            if (out != reference<int32_t>(m, nrows, x, h, nullptr)) { printf("case %d: result differs\n", c); return 2; }
        }
        done++;
    }
    // the tile-height rule at the edges
    for (uint32_t nrows : {0u, 1u, 15u, 1824u, 1825u, 232965u, 4000000000u})
        for (uint32_t nsl : {1u, 3u, 4u, 8u})
            (void)lds_rows_per_tile(nrows, 1824, nsl, 256);
    printf("sanitized: %d cases, all geometries, no finding\n", done);
    return 0;
}
