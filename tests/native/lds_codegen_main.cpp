// Test infrastructure: the data-parallel form of the code-stream encoder (pygim_amd/csrc/lds_codegen.hpp: the bodies the device runs as
// HIP kernels) against the host encoder (lds_plan.hpp: lds_plan_build + lds_code_from_plan), BYTE FOR BYTE -- the instruction words of every
// (tile, wave) stream, the stream offsets, the row map, the tile order and the statistics -- over random shapes and every geometry.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -pthread tests/native/lds_codegen_main.cpp -o /tmp/cg && /tmp/cg [cases]
#include "../../pygim_amd/csrc/lds_codegen.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

using namespace pygim;

struct Csr {
    std::vector<uint32_t> rowptr, col;
};

static Csr random_csr(std::mt19937_64 &rng, uint32_t nrows, uint32_t ncols, double avg_deg, double empty_frac, uint32_t long_row, uint32_t long_len, bool dups,
                      bool clustered) {
    Csr m;
    m.rowptr.assign(nrows + 1, 0);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    std::vector<uint32_t> row;
    for (uint32_t r = 0; r < nrows; r++) {
        uint32_t deg = 0;
        if (r == long_row && long_len) deg = long_len;
        else if (u(rng) >= empty_frac) deg = (uint32_t)std::max(0.0, -avg_deg * std::log(1.0 - u(rng) * 0.999));
        if (!dups) deg = std::min(deg, ncols);
        row.clear();
        const uint32_t centre = (uint32_t)((uint64_t)r * ncols / std::max(1u, nrows)), width = std::max(8u, ncols / 12);
        while (row.size() < deg) {
            uint32_t c = (uint32_t)(rng() % ncols);
            if (clustered) c = (uint32_t)((centre + rng() % width) % ncols);
            if (!dups && std::find(row.begin(), row.end(), c) != row.end()) {
                if (clustered && row.size() >= std::min(width, ncols)) break;
                continue;
            }
            row.push_back(c);
        }
        std::sort(row.begin(), row.end());
        m.col.insert(m.col.end(), row.begin(), row.end());
        m.rowptr[r + 1] = (uint32_t)m.col.size();
    }
    return m;
}

int main(int argc, char **argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 60;
    std::mt19937_64 rng(argc > 2 ? (uint64_t)atoll(argv[2]) : 777);
    // (waves, accumulators, columns per chunk, ring buffers, staged columns per group, x-register sets, row bytes)
    const uint32_t geos[][7] = {{8, 228, 128, 5, 10, 2, 256}, {8, 228, 128, 5, 0, 0, 256}, {16, 96, 320, 2, 8, 2, 256}, {16, 96, 192, 3, 8, 2, 256}, {8, 228, 320, 2, 10, 2, 256},
                                {8, 228, 192, 3, 6, 3, 256}, {8, 228, 160, 4, 12, 2, 256}, {8, 228, 64, 8, 2, 2, 256}, {8, 228, 32, 5, 4, 3, 256}, {8, 7, 128, 5, 10, 2, 256},
                                {16, 5, 128, 4, 12, 2, 256}, {8, 114, 64, 5, 5, 2, 512}, {8, 114, 48, 6, 3, 3, 512}, {8, 114, 96, 3, 2, 2, 512}, {8, 9, 16, 5, 5, 2, 512}};
    int half_cases = 0;
    for (int c = 0; c < cases; c++) {
        const uint32_t nrows = 1 + (uint32_t)(rng() % 2500), ncols = 1 + (uint32_t)(rng() % 3500);
        const double deg = 1 + (double)(rng() % 50);
        const bool dups = rng() % 4 == 0, clustered = rng() % 3 == 0;
        const Csr m = random_csr(rng, nrows, ncols, deg, (rng() % 4) * 0.2, (uint32_t)(rng() % nrows), (rng() % 3) ? 0 : (uint32_t)(rng() % 3000), dups, clustered);
        if (m.col.empty()) continue;
        const auto &q = geos[rng() % (sizeof geos / sizeof geos[0])];
        LdsGeometry geo;
        geo.NW = q[0]; geo.KA = q[1]; geo.KC = q[2]; geo.NBUF = q[3]; geo.row_bytes = q[6]; geo.BATCH = 8;
        geo.boundary = 1;
        geo.keep_tile_order = (rng() % 4 == 0) ? 1 : 0;   // tiles in row order (the dense half of a density split) instead of heaviest first
        geo.rows_per_tile = (rng() % 4 == 0) ? 16 + (uint32_t)(rng() % 1500) : 0;
        geo.col_splits = (rng() % 4 == 0) ? 2 + (uint32_t)(rng() % 7) : 1;   // tiles split into column ranges (short row shares on N GPUs)
        const bool wide = q[6] == 512;
        const uint32_t ops4[] = {0x02000000u, 0x68000000u, LDS_CODE_PK_ADD_U16};
        const uint32_t op = wide ? ((rng() & 1) ? LDS_CODE_ADD_F64 : LDS_CODE_ADD_U64) : ops4[rng() % 3];
        const bool valued = !wide && geo.col_splits == 1 && rng() % 3 == 0;   // FLT32, INT32 (round 5), INT16 (round 5: v_pk_mul_lo_u16)
        std::vector<uint32_t> vals;
        if (valued) {
            vals.resize(m.col.size());
            const bool small = rng() & 1;   // (INT32: values that ride as inline constants, or any)
            for (auto &v : vals) v = small ? (uint32_t)((int32_t)(rng() % 81) - 16) : op == LDS_CODE_PK_ADD_U16 ? (uint32_t)(int32_t)(int16_t)rng() : (uint32_t)rng();
        }
        // valued DBL64 (round 5): the plan's value slot carries the entry index, the values travel beside it
        const bool valued64 = wide && geo.col_splits == 1 && rng() % 2 == 0;   // DBL64: any bits; INT64: inline values, values that fit int32, or any 64 bits
        std::vector<uint64_t> vals64;
        std::vector<uint32_t> eidx;
        if (valued64) {
            vals64.resize(m.col.size());
            eidx.resize(m.col.size());
            const uint32_t cls = (uint32_t)(rng() % 3);   // INT64: every value inline / every value fits int32 / any 64 bits
            for (size_t i = 0; i < vals64.size(); i++) {
                eidx[i] = (uint32_t)i;
                if (op == LDS_CODE_ADD_F64) vals64[i] = rng();
                else vals64[i] = cls == 2 ? (rng() % 4 ? rng() : (uint64_t)(int64_t)(int32_t)rng()) : (uint64_t)(int64_t)(cls == 0 ? (int32_t)(rng() % 81) - 16 : (int32_t)rng());
            }
        }
        // half-split plans (round 6): two column ranges folded into the halves of a wave -- the host encoder sees the VIRTUAL matrix (columns c mod H, every row's two
        // sorted runs merged, the lower range first on ties; the entry's half in its value slot), the data-parallel form the stored columns and H
        const bool half = !wide && !valued && (op == 0x02000000u || op == 0x68000000u) && geo.NW == 8 && rng() % 3 == 0;
        uint32_t half_H = 0;
        std::vector<uint32_t> col2, hv;
        if (half) {
            half_H = (((ncols + 1) / 2 + geo.KC - 1) / geo.KC) * geo.KC;
            geo.half_split = 1;
            half_cases++;
            col2.resize(m.col.size());
            hv.resize(m.col.size());
            for (uint32_t r = 0; r < nrows; r++) {
                const uint32_t e0 = m.rowptr[r], e1 = m.rowptr[r + 1];
                const uint32_t mid = (uint32_t)(std::lower_bound(m.col.begin() + e0, m.col.begin() + e1, half_H) - m.col.begin());
                uint32_t a = e0, b = mid, o = e0;
                while (a < mid || b < e1) {
                    const bool low = b >= e1 || (a < mid && m.col[a] <= m.col[b] - half_H);
                    if (low) { col2[o] = m.col[a++]; hv[o++] = 0u; }
                    else { col2[o] = m.col[b++] - half_H; hv[o++] = 1u; }
                }
            }
        }
        // tiles of consecutive rows, or of rows in an arbitrary order (similarity tiles)
        std::vector<uint32_t> rorder;
        if (rng() % 3 == 0) {
            rorder.resize(nrows);
            for (uint32_t i = 0; i < nrows; i++) rorder[i] = i;
            std::shuffle(rorder.begin(), rorder.end(), rng);
        }
        const uint32_t *ro = rorder.empty() ? nullptr : rorder.data();
        // the host encoder
        LdsPlanHost plan;
        if (half) lds_plan_build(m.rowptr.data(), col2.data(), nrows, half_H, geo, plan, 2, hv.data(), ro);
        else lds_plan_build(m.rowptr.data(), m.col.data(), nrows, ncols, geo, plan, 2, valued ? vals.data() : (valued64 ? eidx.data() : nullptr), ro);
        LdsCodeHost ch;
        lds_code_from_plan(plan, op, ch, 2, q[4], q[5], 0, valued64 ? vals64.data() : nullptr);
        // the data-parallel form
        CgHostResult r;
        cg_run_on_host(m.rowptr.data(), m.col.data(), valued ? vals.data() : nullptr, nrows, half ? half_H : ncols, geo, op, r, q[4], q[5], ro, valued64 ? vals64.data() : nullptr, half_H);
        auto bad = [&](const char *what) {
            printf("case %d: %s differs (geo %u %u %u %u %u %u %u, %u x %u, nnz %zu, rpt %u, splits %u, op %08x, valued %d, dups %d, clustered %d, half H %u)\n", c, what, q[0], q[1], q[2], q[3],
                   q[4], q[5], q[6], nrows, ncols, m.col.size(), geo.rows_per_tile, geo.col_splits, op, (int)valued, (int)dups, (int)clustered, half_H);
            return 1;
        };
        if (r.rows.rowmap != plan.rowmap) return bad("row map");
        if (r.start != ch.start) return bad("stream offsets");
        if (r.code.size() != ch.code.size()) return bad("code size");
        if (memcmp(r.code.data(), ch.code.data(), ch.code.size() * 4) != 0) {
            size_t at = 0;
            while (r.code[at] == ch.code[at]) at++;
            size_t s = 0;
            while (s + 1 < ch.start.size() && ch.start[s + 1] / 4 <= at) s++;
            printf("first difference at dword %zu (stream %zu + %zu): %08x vs host %08x\n", at, s, at - (size_t)(ch.start[s] / 4), r.code[at], ch.code[at]);
            return bad("code");
        }
        for (uint32_t t = 0; t < plan.ntiles; t++)
            if (plan.tiles[t].nch != r.chunks.nch[t] || plan.tiles[t].row0 != r.rows.tile_row0[t] || plan.tiles[t].nnz != r.rows.tile_nnz[t]) return bad("tile table");
        if (r.entries != ch.entries || r.pairs != ch.pairs || r.shared != ch.shared || r.chunks.slots != plan.slots) return bad("statistics");
    }
    printf("codegen: %d cases (%d of them half-split plans), data-parallel form == host encoder, byte for byte\n", cases, half_cases);
    return 0;
}
