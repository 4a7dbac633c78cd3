// Test infrastructure: a CPU emulator of k_lds_spmm's token walk over the schedule lds_plan.hpp builds.
// It reads the plan exactly as the kernel does (tile table, chunk lists, batch counts, token streams, row map)
// so that tests/test_lds_plan.py can compare "what the schedule says" with the oracle on the CPU.
#include "../../pygim_amd/csrc/lds_plan.hpp"

#include <cstring>
#include <type_traits>

using namespace pygim;

template <typename T>
static int emulate(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const T *X, uint32_t h, T *Cout,
                   uint32_t KA, uint32_t batch, uint32_t threads, uint64_t *stats, uint32_t nw, const T *vals, uint32_t col_splits = 1) {
    // column-split plans: tile (t, c) writes the partial sums of column range c to row r + c * nrows; summed in range order at the end
    const uint32_t nrows_real = nrows, S = col_splits ? col_splits : 1;
    std::vector<T> part;
    T *C = Cout;
    if (S > 1) {
        part.assign((size_t)S * nrows * h, T(0));
        C = part.data();
    }
    LdsGeometry geo;
    geo.col_splits = S;
    geo.NW = nw;
    geo.KA = KA;
    geo.BATCH = batch;
    LdsPlanHost plan;
    static_assert(sizeof(T) == 4, "4-byte values");
    lds_plan_build(rowptr, col, nrows, ncols, geo, plan, threads, reinterpret_cast<const uint32_t *>(vals));
    if (plan.header_overflow) return 13;
    if ((vals != nullptr) != !plan.wts.empty() || (vals && plan.wts.size() != plan.tok.size())) return 9;
    nrows = S * nrows_real;   // rows of the (partial-sum) result the plan writes
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
                        if (k < KA && ldsrow / KC != (j % geo.NBUF)) return 15;
                        if (ldsrow >= geo.NBUF * KC) return 16;
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
                    const uint32_t row = plan.rowmap[((size_t)ti * NW + w) * geo.ka_stride() + k];
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
    if (S > 1)
        for (uint32_t r = 0; r < nrows_real; r++)
            for (uint32_t f = 0; f < h; f++) {
                T acc = part[(size_t)r * h + f];
                for (uint32_t c = 1; c < S; c++) {
                    const T v = part[((size_t)c * nrows_real + r) * h + f];
                    if constexpr (std::is_integral<T>::value) acc = (T)((uint32_t)acc + (uint32_t)v);
                    else acc = acc + v;
                }
                Cout[(size_t)r * h + f] = acc;
            }
    return 0;
}

extern "C" {
int lds_emul_f32(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const float *X, uint32_t h, float *C,
                 uint32_t KA, uint32_t batch, uint32_t threads, uint64_t *stats, uint32_t nw, const float *vals, uint32_t col_splits) {
    return emulate<float>(rowptr, col, nrows, ncols, X, h, C, KA, batch, threads, stats, nw, vals, col_splits);
}
int lds_emul_i32(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const int32_t *X, uint32_t h, int32_t *C,
                 uint32_t KA, uint32_t batch, uint32_t threads, uint64_t *stats, uint32_t nw, const int32_t *vals, uint32_t col_splits) {
    return emulate<int32_t>(rowptr, col, nrows, ncols, X, h, C, KA, batch, threads, stats, nw, vals, col_splits);
}
}

extern "C" uint32_t lds_emul_rows_per_tile(uint32_t nrows, uint32_t rmax, uint32_t nslices, uint32_t cus) {
    return lds_rows_per_tile(nrows, rmax, nslices, cus);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The code-stream form (lds_code_from_plan): a CPU interpreter of exactly the instructions the stream may contain.  It models
// what the hardware would do with them -- the two 80 KiB chunk buffers in LDS filled by the DMA sequences (chunk id and
// destination decoded from the literals), the x registers written by ds_read_b32 / ds_read2st64_b32 (base register + offsets),
// the accumulators named by the add's register field, lgkmcnt waits (an add must not read an x register whose read is still
// "in flight": in-order LDS returns, so a wait for N outstanding reads retires all but the N youngest) -- and rejects anything else.
// ---------------------------------------------------------------------------------------------------------------------------
template <typename T>
static int run_code(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const T *X, uint32_t h, T *Cout,
                    uint32_t threads, uint64_t *stats, uint32_t kc, uint32_t nbuf, const T *vals = nullptr, uint32_t col_splits = 1,
                    uint32_t nw = 16, uint32_t gsize = 0, uint32_t nsets = 0, uint32_t rows_per_tile = 0, uint32_t boundary = 0) {
    const uint32_t nrows_real = nrows, S = col_splits ? col_splits : 1;
    std::vector<T> part;
    T *C = Cout;
    if (S > 1) {
        part.assign((size_t)S * nrows * h, T(0));
        C = part.data();
    }
    LdsGeometry geo;
    geo.col_splits = S;
    constexpr bool WIDE = sizeof(T) == 8;
    constexpr bool PK16 = sizeof(T) == 2;      // INT16: two features to a lane (v_pk_add_u16 / v_pk_mul_lo_u16), 128 features to a 256-byte row
    constexpr uint32_t F = PK16 ? 128 : 64;     // features of a slice
    if (PK16 && nw != 8) return 12;   // INT64 / DBL64: 512-byte rows, ds_read_b64, register pairs (8 waves x 114 accumulators)
    if (WIDE && nw != 8) return 12;
    geo.NW = nw == 8 ? 8 : 16;          // round 4: 8 waves x 228 accumulators (2 waves per SIMD) beside 16 x 96
    geo.KA = nw == 8 ? (WIDE ? LDS_CODE8_KA64 : LDS_CODE8_KA) : 96;
    geo.row_bytes = WIDE ? 512 : 256;
    geo.BATCH = 8;
    geo.rows_per_tile = rows_per_tile;
    geo.boundary = boundary;   // rings of >= 3 buffers: the workgroup meets at the slot boundary instead of in the middle of a slot
    geo.KC = kc;       // the code-stream ring (pygim_hip.hip build_lds_plan): 2 x 320 columns, or 3 x 192 (two chunks in flight)
    geo.NBUF = nbuf;
    LdsPlanHost plan;
    std::vector<uint32_t> eidx;   // 8-byte values (valued DBL64): the plan's 4-byte value slot carries the entry index
    if (WIDE && vals) {
        eidx.resize(rowptr[nrows]);
        for (uint32_t i = 0; i < eidx.size(); i++) eidx[i] = i;
    }
    std::vector<uint32_t> vals32;  // INT16 values: sign-extended to the plan's 4-byte value slot
    if (PK16 && vals) {
        vals32.resize(rowptr[nrows]);
        for (uint32_t i = 0; i < vals32.size(); i++) vals32[i] = (uint32_t)(int32_t)vals[i];
    }
    lds_plan_build(rowptr, col, nrows, ncols, geo, plan, threads, WIDE ? (vals ? eidx.data() : nullptr) : PK16 ? (vals ? vals32.data() : nullptr) : reinterpret_cast<const uint32_t *>(vals));
    if (plan.header_overflow) return 13;
    nrows = S * nrows_real;   // rows of the (partial-sum) result the plan writes
    const uint32_t NBUF = geo.NBUF;
    const uint32_t opcode = std::is_same<T, float>::value ? 0x02000000u : std::is_same<T, double>::value ? LDS_CODE_ADD_F64 :
                            std::is_same<T, int64_t>::value ? LDS_CODE_ADD_U64 : PK16 ? LDS_CODE_PK_ADD_U16 : 0x68000000u;
    LdsCodeHost ch;
    lds_code_from_plan(plan, opcode, ch, threads, gsize, nsets, 0, (WIDE && vals) ? reinterpret_cast<const uint64_t *>(vals) : nullptr);
    const LdsCodeRegs R = ch.regs;
    if (R.nx() > 32 || NBUF > 16) return 14;
    const uint32_t NW = geo.NW, KA = geo.KA, KC = geo.KC, RB = geo.row_bytes, chunk_bytes = KC * RB, RPB = 65536 / RB;
    const uint32_t nslices = (h + F - 1) / F;
    if (stats) {
        stats[0] = plan.ntiles;
        stats[1] = ch.code.size() * 4;
        stats[2] = ch.entries;
        stats[3] = ch.pairs;
    }
    uint64_t entries_seen = 0;
    std::vector<char> written((size_t)nrows, 0);
    for (uint32_t ti = 0; ti < plan.ntiles; ti++) {
        for (uint32_t s = 0; s < nslices; s++) {
            const uint32_t wvalid = std::min(F, h - s * F);
            // LDS as the 16 waves of the workgroup see it: which chunk each buffer holds (the DMA of all waves lands the same chunk)
            uint64_t barriers_w0 = 0;
            for (uint32_t w = 0; w < NW; w++) {
                std::vector<T> acc((size_t)KA * F, T(0));
                int64_t buf_chunk[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};      // what a buffer holds, as far as this wave may rely on it
                int64_t landed_chunk[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};   // landed (waited for), visible to everybody after the next barrier
                bool dirty[16] = {false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false}; // a DMA into the buffer has been issued and not yet fenced (wait + barrier)
                // a DMA may only go into a buffer that NOBODY reads any more: every wave issues the same DMAs between the same two
                // barriers, so it is enough that this wave had no read of that buffer in flight at its last barrier and has issued
                // none since
                bool inflight_at_barrier[16] = {false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false};
                bool read_since_barrier[16] = {false, false, false, false, false, false, false, false, false, false, false, false, false, false, false, false};
                uint64_t barriers = 0;
                struct VLoad { int buf; int64_t cid; bool last; };
                std::vector<VLoad> vfifo;                     // vector loads in flight, oldest first (DMA pieces and touches)
                struct XReg { bool valid = false, inflight = false, has_mul = false; uint32_t ldsrow = 0, mulbits = 0; uint64_t mulbits64 = 0; int64_t chunk = -1;
                              int mstage = 0; int32_t mval = 0; uint32_t macc = 0; };   // (valued INT64: where the entry's multiply sequence stands)
                XReg x[32];
                std::vector<uint32_t> fifo;                   // x registers with a read in flight, oldest first
                uint64_t pc = ch.start[(size_t)ti * NW + w] / 4;
                uint64_t pa = 0;                              // DMA source offset inside the slice (bytes), from the literal
                int64_t dma_buf = -1, dma_cid = -1;
                uint32_t pieces_seen = 0;
                const uint32_t pieces = (KC * RB / 1024) / NW;
                bool done = false;
                int64_t lo_half = -1;                         // INT64: the v_add_co_u32 of a pair has been seen for this accumulator
                uint32_t sval = 0;                            // valued INT32: the SGPR a general value travels in
                bool sval_set = false;
                uint32_t sval_hi = 0;                         // valued DBL64: s95, the high half (s94 = sval the low one)
                bool sval_hi_set = false;
                bool mad_add = false;                  // valued INT64: a v_mad_u64_u32 was decoded, its accumulate follows at once
                uint32_t mad_vd = 0, mad_vx = 0;
                for (uint64_t guard = 0; !done; guard++) {
                    if (pc >= ch.code.size() || guard > (1ull << 32)) return 20;
                    // vmcnt is a 6-bit counter: a wave has at most 63 vector loads in flight (the next one is not issued before the oldest
                    // has returned), so whatever lies more than 63 loads back HAS landed -- s_waitcnt vmcnt(63) is the no-op it encodes as
                    while (vfifo.size() > 63) {
                        if (vfifo.front().last) landed_chunk[vfifo.front().buf] = vfifo.front().cid;
                        vfifo.erase(vfifo.begin());
                    }
                    const uint32_t i0 = ch.code[pc];
                    if (i0 == 0xBF800000u) { pc++; continue; }                                  // s_nop 0
                    if (i0 == 0xBF8A0000u) {                                                    // s_barrier: landed chunks become visible
                        for (uint32_t b = 0; b < NBUF; b++)
                            if (landed_chunk[b] >= 0) { buf_chunk[b] = landed_chunk[b]; landed_chunk[b] = -1; dirty[b] = false; }
                        for (uint32_t b = 0; b < NBUF; b++) inflight_at_barrier[b] = read_since_barrier[b] = false;
                        for (uint32_t r : fifo)
                            for (int q = 0; q < 2; q++)
                                if (r + q < 32 && x[r + q].inflight) inflight_at_barrier[x[r + q].ldsrow / KC] = true;
                        barriers++;
                        pc++;
                        continue;
                    }
                    if ((i0 & 0xFFFFF0FFu) == 0xBF8CC07Fu) {                                    // s_waitcnt lgkmcnt(n)
                        const uint32_t n = (i0 >> 8) & 0xF;
                        while (fifo.size() > n) { x[fifo.front()].inflight = false; fifo.erase(fifo.begin()); }
                        pc++;
                        continue;
                    }
                    if ((i0 & 0xFFFF3FF0u) == 0xBF8C0F70u) {                                    // s_waitcnt vmcnt(N): all but the N youngest loads
                        const uint32_t n = (i0 & 15) | (((i0 >> 14) & 3) << 4);
                        while (vfifo.size() > n) {
                            if (vfifo.front().last) landed_chunk[vfifo.front().buf] = vfifo.front().cid;
                            vfifo.erase(vfifo.begin());
                        }
                        pc++;
                        continue;
                    }
                    if (i0 == (0xBE801D00u | R.s_ret)) {                                        // s_setpc_b64: back to the kernel
                        if (!fifo.empty() || !vfifo.empty()) return 42;                         // (something still in flight)
                        done = true;
                        continue;
                    }
                    if ((i0 & 0xFF00FFFFu) == (0x8000FF00u | R.s_xs) && ((i0 >> 16) & 0xFF) == R.s_pa) {   // s_add_u32 pa, xs, literal (chunk)
                        pa = ch.code[pc + 1];
                        if (pa % chunk_bytes) return 21;
                        dma_cid = pa / chunk_bytes;
                        pieces_seen = 0;
                        pc += 2;
                        continue;
                    }
                    if ((i0 & 0xFF00FFFFu) == (0x8000FF00u | R.s_pa) && ((i0 >> 16) & 0xFF) == R.s_pa) {   // s_add_u32 pa, pa, 0x1000
                        if (ch.code[pc + 1] != 0x1000) return 22;
                        pc += 2;
                        continue;
                    }
                    if ((i0 & 0xFF00FFFFu) == (0x8000FF00u | R.s_cb) && ((i0 >> 16) & 0xFF) == R.s_cb) { pc += 2; continue; }   // touch pointer
                    if ((i0 & 0xFF00FFFFu) == (0x8000FF00u | R.s_ldsw) && ((i0 >> 16) & 0xFF) == 124) {    // s_add_u32 m0, ldsw, literal
                        const uint32_t lit = ch.code[pc + 1];
                        dma_buf = lit / chunk_bytes;
                        if (dma_buf >= (int64_t)NBUF || (lit % chunk_bytes) != (pieces_seen / 4) * 0x1000) return 23;
                        pc += 2;
                        continue;
                    }
                    if ((i0 & 0xFF00FF00u) == 0x82008000u) { pc++; continue; }                  // s_addc_u32 x, x, 0
                    if ((i0 & 0xFFFF8000u) == 0xDDF48000u) {                                    // global_load_lds_dwordx4 (one DMA piece)
                        if (ch.code[pc + 1] != ((R.s_pa << 16) | R.vl16) || (i0 & 0x1FFF) != (pieces_seen % 4) * 1024 || dma_buf < 0) return 24;
                        // the buffer being filled must not be one that is read before the fence: checked at the reads below
                        if (inflight_at_barrier[dma_buf] || read_since_barrier[dma_buf]) return 36;   // somebody may still be reading it
                        dirty[dma_buf] = true;
                        buf_chunk[dma_buf] = -1;
                        ++pieces_seen;
                        vfifo.push_back({(int)dma_buf, dma_cid, pieces_seen == pieces});
                        pc += 2;
                        continue;
                    }
                    if ((i0 & 0xFFFF8000u) == 0xDC508000u) {                                    // global_load_dword (touch)
                        if (ch.code[pc + 1] != ((R.vjunk << 24) | (R.s_cb << 16) | R.vtouch)) return 37;
                        vfifo.push_back({-1, -1, false});
                        pc += 2;
                        continue;
                    }
                    if (WIDE && (i0 & 0xFFFF0000u) == 0xD8EC0000u) {                            // ds_read_b64 x[0:1], base offset:row * 512
                        const uint32_t i1 = ch.code[pc + 1], vdst = i1 >> 24, vaddr = i1 & 0xFF, off = i0 & 0xFFFF;
                        int blk = -1;
                        for (int b = 0; b < 3; b++) if (vaddr == R.vbase[b]) blk = b;
                        if (blk < 0 || vdst < R.x0 || vdst + 1 >= R.x0 + R.nx() || ((vdst - R.x0) & 1) || off % RB) return 25;
                        XReg &xr = x[vdst - R.x0];
                        if (xr.inflight) return 27;
                        xr.valid = xr.inflight = true;
                        xr.has_mul = false;
                        xr.ldsrow = blk * RPB + off / RB;
                        if (xr.ldsrow >= NBUF * KC) return 28;
                        if (dirty[xr.ldsrow / KC]) return 29;
                        if (buf_chunk[xr.ldsrow / KC] < 0) return 38;
                        xr.chunk = buf_chunk[xr.ldsrow / KC];
                        read_since_barrier[xr.ldsrow / KC] = true;
                        fifo.push_back(vdst - R.x0);
                        pc += 2;
                        continue;
                    }
                    if (WIDE && vals && i0 == (0xBE8000FFu | (LDS_CODE_S_VAL << 16))) { sval = ch.code[pc + 1]; sval_set = true; pc += 2; continue; }           // s_mov_b32 s94, <low half>
                    if (WIDE && vals && i0 == (0xBE8000FFu | ((LDS_CODE_S_VAL + 1) << 16))) { sval_hi = ch.code[pc + 1]; sval_hi_set = true; pc += 2; continue; }   // s_mov_b32 s95, <high half>
                    if (WIDE && vals && opcode == LDS_CODE_ADD_U64) {
                        // valued INT64 (values that fit int32): v_mul_lo_u32 x_hi, V, x_hi ; [v_sub_u32 x_hi, x_hi, x_lo] ; v_add_u32 acc_hi, x_hi, acc_hi ;
                        // v_mad_u64_u32 acc[0:1], vcc, V, x_lo, acc[0:1] -- V an inline constant or s94; the last one is the entry's accumulate
                        auto operand = [&](uint32_t src0, int32_t *val) -> bool {
                            if (src0 == LDS_CODE_S_VAL) { if (!sval_set) return false; *val = (int32_t)sval; return true; }
                            if (src0 >= 128 && src0 <= 192) { *val = (int32_t)(src0 - 128); return true; }
                            if (src0 >= 193 && src0 <= 208) { *val = -(int32_t)(src0 - 192); return true; }
                            return false;
                        };
                        if ((i0 & 0xFFFFFF00u) == 0xD2850000u) {                                // v_mul_lo_u32 x_hi, V, x_hi : the sequence starts
                            const uint32_t d1 = ch.code[pc + 1], vd = i0 & 0xFF, src0 = d1 & 0x1FF, vs1 = ((d1 >> 9) & 0x1FF) - 256;
                            if ((d1 >> 18) || vd < R.x0 || vd >= R.x0 + R.nx() || !((vd - R.x0) & 1)) return 33;
                            const uint32_t base = (vd - R.x0) & ~1u;
                            XReg &xr = x[base];
                            int32_t v = 0;
                            if (src0 != LDS_CODE_S_VAL + 1 && !operand(src0, &v)) return 46;
                            if (src0 == LDS_CODE_S_VAL + 1) {                                 // full 64-bit values: v_mul_lo_u32 x_hi, s95, x_lo (x_lo * v_hi)
                                if (!sval_hi_set || vs1 != vd - 1 || xr.mstage != 3) return 47;
                                xr.mstage = 5;
                                pc += 2;
                                continue;
                            }
                            bool infl = false;
                            for (uint32_t r : fifo) if (r == base) infl = true;
                            if (vs1 != vd || !xr.valid || infl || xr.has_mul || xr.mstage != 0) return 34;
                            xr.inflight = false;
                            xr.mstage = 1;
                            xr.mval = v;
                            pc += 2;
                            continue;
                        }
                        if ((i0 & 0xFFFFFF00u) == 0xD1E86A00u) {                                // v_mad_u64_u32 acc[0:1], vcc, V, x_lo, acc[0:1]
                            const uint32_t d1 = ch.code[pc + 1], vd = i0 & 0xFF, src0 = d1 & 0x1FF, vs1 = ((d1 >> 9) & 0x1FF) - 256, vs2 = ((d1 >> 18) & 0x1FF) - 256;
                            if ((d1 >> 27) || vs2 != vd || vs1 < R.x0 || vs1 >= R.x0 + R.nx() || ((vs1 - R.x0) & 1)) return 47;
                            XReg &xr = x[vs1 - R.x0];
                            int32_t v;
                            if (!operand(src0, &v)) return 46;
                            if ((xr.mstage != 3 && xr.mstage != 6) || v != xr.mval || xr.macc != vd) return 47;
                            if ((xr.mstage == 6) != sval_hi_set) return 47;                    // (s95 is set exactly in the full form)
                            xr.mulbits64 = xr.mstage == 6 ? (((uint64_t)sval_hi << 32) | (uint32_t)xr.mval) : (uint64_t)(int64_t)xr.mval;
                            xr.mstage = 0;
                            xr.has_mul = true;
                            if (src0 == LDS_CODE_S_VAL) sval_set = false;
                            sval_hi_set = false;
                            mad_vd = vd;
                            mad_vx = vs1;
                            mad_add = true;
                            pc += 2;
                        }
                        if (!mad_add && (i0 & 0xFE000000u) == 0x6A000000u) {                    // v_sub_u32 x_hi, x_hi, x_lo (negative values only)
                            const uint32_t vd = (i0 >> 17) & 0xFF, vs1 = (i0 >> 9) & 0xFF, s0 = (i0 & 0x1FF) - 256;
                            if (vd < R.x0 + 1 || vd >= R.x0 + R.nx() || !((vd - R.x0) & 1) || s0 != vd || vs1 != vd - 1) return 47;
                            XReg &xr = x[vd - 1 - R.x0];
                            if (xr.mstage != 1 || xr.mval >= 0) return 47;
                            xr.mstage = 2;
                            pc++;
                            continue;
                        }
                        if (!mad_add && (i0 & 0xFE000000u) == 0x68000000u) {                    // v_add_u32 acc_hi, x_hi, acc_hi
                            const uint32_t vd = (i0 >> 17) & 0xFF, vs1 = (i0 >> 9) & 0xFF, s0 = (i0 & 0x1FF) - 256;
                            if (vd != vs1 || vd < R.acc0 + 1 || vd >= R.acc0 + 2 * KA || !((vd - R.acc0) & 1) || s0 < R.x0 + 1 || s0 >= R.x0 + R.nx() || !((s0 - R.x0) & 1)) return 47;
                            XReg &xr = x[s0 - 1 - R.x0];
                            if (xr.mstage == 5) {                        // full 64-bit values: the second sum into the high half
                                if (xr.macc != vd - 1) return 48;
                                xr.mstage = 6;
                                pc++;
                                continue;
                            }
                            // (one-SGPR / inline form: a negative value passes through the v_sub_u32; the full form has no correction)
                            if (xr.mstage != ((xr.mval < 0 && !sval_hi_set) ? 2 : 1)) return 47;
                            xr.mstage = 3;
                            xr.macc = vd - 1;
                            pc++;
                            continue;
                        }
                    }
                    if (WIDE && vals && opcode == LDS_CODE_ADD_F64 && (i0 & 0xFFFFFF00u) == 0xD2810000u) {                                    // v_mul_f64 x[0:1], s[94:95], x[0:1]
                        const uint32_t d1 = ch.code[pc + 1], vd = i0 & 0xFF;
                        if ((d1 & 0x1FF) != LDS_CODE_S_VAL || ((d1 >> 9) & 0x1FF) != 256 + vd || (d1 >> 18) || !sval_set || !sval_hi_set) return 45;
                        if (vd < R.x0 || vd + 1 >= R.x0 + R.nx() || ((vd - R.x0) & 1)) return 33;
                        XReg &xr = x[vd - R.x0];
                        bool infl = false;
                        for (uint32_t r : fifo) if (r == vd - R.x0) infl = true;
                        if (!xr.valid || infl || xr.has_mul) return 34;
                        xr.inflight = false;
                        xr.has_mul = true;
                        xr.mulbits64 = ((uint64_t)sval_hi << 32) | sval;
                        sval_set = sval_hi_set = false;
                        pc += 2;
                        continue;
                    }
                    if (WIDE) {
                        // v_add_f64 acc[0:1], x[0:1], acc[0:1]   |   v_add_co_u32 acc0, vcc, x0, acc0 ; v_addc_co_u32 acc1, vcc, x1, acc1, vcc
                        uint32_t vd = 0, vx = 0;
                        bool is_add = false;
                        if (mad_add) {                      // (valued INT64: the v_mad_u64_u32 decoded above)
                            vd = mad_vd;
                            vx = mad_vx;
                            is_add = true;
                            mad_add = false;
                        } else if (opcode == LDS_CODE_ADD_F64 && (i0 & 0xFFFFFF00u) == 0xD2800000u) {
                            const uint32_t d1 = ch.code[pc + 1];
                            vd = i0 & 0xFF;
                            vx = (d1 & 0x1FF) - 256;
                            if (((d1 >> 9) & 0x1FF) != 256 + vd || (d1 >> 18)) return 30;
                            is_add = true;
                            pc += 2;
                        } else if (opcode == LDS_CODE_ADD_U64 && (i0 & 0xFE000000u) == 0x32000000u) {
                            if (lo_half >= 0) return 44;
                            lo_half = i0;
                            pc++;
                            continue;
                        } else if (opcode == LDS_CODE_ADD_U64 && (i0 & 0xFE000000u) == 0x38000000u) {
                            if (lo_half < 0) return 44;
                            const uint32_t l0 = (uint32_t)lo_half;
                            vd = (l0 >> 17) & 0xFF;
                            vx = (l0 & 0x1FF) - 256;
                            if (((l0 >> 9) & 0xFF) != vd || ((i0 >> 17) & 0xFF) != vd + 1 || ((i0 >> 9) & 0xFF) != vd + 1 || (i0 & 0x1FF) != 256 + vx + 1) return 30;
                            lo_half = -1;
                            is_add = true;
                            pc++;
                        }
                        if (is_add) {
                            if (vd < R.acc0 || vd + 1 >= R.acc0 + 2 * KA || ((vd - R.acc0) & 1) || vx < R.x0 || vx + 1 >= R.x0 + R.nx() || ((vx - R.x0) & 1)) return 30;
                            XReg &xr = x[vx - R.x0];
                            bool infl = false;
                            for (uint32_t r : fifo) if (r == vx - R.x0) infl = true;
                            if (!xr.valid || infl) return 31;
                            xr.inflight = false;
                            if (xr.chunk < 0) return 32;
                            const uint64_t xrow = (uint64_t)xr.chunk * KC + xr.ldsrow % KC;
                            const uint32_t k = (vd - R.acc0) / 2;
                            if ((vals != nullptr) != xr.has_mul) return 35;                     // every entry of a valued matrix is multiplied once
                            if (vals && std::is_integral<T>::value && xr.macc != vd) return 48;  // (INT64: its high half went to this accumulator)
                            xr.has_mul = false;
                            const bool had_mul = vals != nullptr;
                            for (uint32_t l = 0; l < wvalid; l++) {
                                T xv = xrow < ncols ? X[xrow * h + s * 64 + l] : T(0);
                                T &a = acc[(size_t)k * 64 + l];
                                if constexpr (std::is_integral<T>::value) a = (T)((uint64_t)a + (had_mul ? xr.mulbits64 * (uint64_t)xv : (uint64_t)xv));
                                else {
                                    if (had_mul) {
                                        double mv;
                                        std::memcpy(&mv, &xr.mulbits64, 8);
                                        volatile double prod = mv * xv;   // product and sum round separately (no FMA)
                                        xv = prod;
                                    }
                                    a = a + xv;
                                }
                            }
                            if (s == 0) entries_seen++;
                            continue;
                        }
                    }
                    if ((i0 & 0xFFFF0000u) == 0xD8700000u || (i0 & 0xFFFF0000u) == 0xD86C0000u) {   // ds_read2st64_b32 / ds_read_b32
                        const bool two = (i0 & 0xFFFF0000u) == 0xD8700000u;
                        const uint32_t i1 = ch.code[pc + 1], vdst = i1 >> 24, vaddr = i1 & 0xFF;
                        int blk = -1;
                        for (int b = 0; b < 3; b++) if (vaddr == R.vbase[b]) blk = b;
                        if (blk < 0 || vdst < R.x0 || vdst + (two ? 1 : 0) >= R.x0 + R.nx() || (two && ((vdst - R.x0) & 1))) return 25;
                        const uint32_t rows[2] = {two ? (i0 & 0xFF) : ((i0 & 0xFFFF) >> 8), two ? ((i0 >> 8) & 0xFF) : 0};
                        if (!two && (i0 & 0xFF)) return 26;
                        for (int q = 0; q < (two ? 2 : 1); q++) {
                            XReg &xr = x[vdst - R.x0 + q];
                            if (xr.inflight) return 27;                                         // overwritten before it was consumed
                            xr.valid = true;
                            xr.inflight = true;
                            xr.has_mul = false;
                            xr.ldsrow = blk * 256 + rows[q];
                            if (xr.ldsrow >= NBUF * KC) return 28;
                            if (dirty[xr.ldsrow / KC]) return 29;                               // reading a buffer whose DMA has not been fenced
                            if (buf_chunk[xr.ldsrow / KC] < 0) return 38;                       // ... or one that holds nothing this wave may rely on
                            xr.chunk = buf_chunk[xr.ldsrow / KC];                               // (the add may come after the buffer has been handed on)
                            read_since_barrier[xr.ldsrow / KC] = true;
                        }
                        fifo.push_back(vdst - R.x0);   // (a pair retires as one LDS instruction: both registers with the first index)
                        if (two) x[vdst - R.x0 + 1].inflight = true;
                        pc += 2;
                        continue;
                    }
                    if (vals && (std::is_same<T, int32_t>::value || PK16) && i0 == (0xBE8000FFu | (LDS_CODE_S_VAL << 16))) {   // s_mov_b32 s94, <literal value>
                        sval = ch.code[pc + 1];
                        sval_set = true;
                        pc += 2;
                        continue;
                    }
                    if (vals && ((std::is_same<T, int32_t>::value && (i0 & 0xFFFFFF00u) == 0xD2850000u) ||          // v_mul_lo_u32 x, <inline | s94>, x
                                 (PK16 && (i0 & 0xFFFFFF00u) == 0xD3814000u))) {                               // v_pk_mul_lo_u16 x, <inline | s94>, x op_sel_hi:[0,1]
                        const uint32_t d1 = ch.code[pc + 1], vd = i0 & 0xFF, src0 = d1 & 0x1FF, vs1 = (d1 >> 9) & 0x1FF;
                        if (vs1 != 256 + vd || (d1 >> 18) != (PK16 ? 0x400u : 0u) || vd < R.x0 || vd >= R.x0 + R.nx()) return 33;
                        uint32_t value;
                        if (src0 == LDS_CODE_S_VAL) {
                            if (!sval_set) return 45;                                           // the SGPR was not loaded for this multiply
                            value = sval;
                            sval_set = false;
                        } else if (src0 >= 128 && src0 <= 192) value = src0 - 128;              // inline 0 .. 64
                        else if (src0 >= 193 && src0 <= 208) value = (uint32_t)(-(int32_t)(src0 - 192));   // inline -1 .. -16
                        else return 46;
                        XReg &xr = x[vd - R.x0];
                        bool infl = false;
                        for (uint32_t r : fifo) if (r == vd - R.x0 || (r + 1 == vd - R.x0 && x[r + 1].inflight && x[r].inflight)) infl = true;
                        if (!xr.valid || infl || xr.has_mul) return 34;
                        xr.inflight = false;
                        xr.has_mul = true;
                        xr.mulbits = value;
                        pc += 2;
                        continue;
                    }
                    if (vals && (i0 & 0xFE0001FFu) == 0x0A0000FFu) {                            // v_mul_f32 x, <literal>, x (valued matrices)
                        const uint32_t vd = (i0 >> 17) & 0xFF, vs1 = (i0 >> 9) & 0xFF;
                        if (vd != vs1 || vd < R.x0 || vd >= R.x0 + R.nx()) return 33;
                        XReg &xr = x[vd - R.x0];
                        bool infl = false;
                        for (uint32_t r : fifo) if (r == vd - R.x0 || (r + 1 == vd - R.x0 && x[r + 1].inflight && x[r].inflight)) infl = true;
                        if (!xr.valid || infl || xr.has_mul) return 34;                         // multiplied before its read landed, or twice
                        xr.inflight = false;
                        xr.has_mul = true;
                        xr.mulbits = ch.code[pc + 1];
                        pc += 2;
                        continue;
                    }
                    if ((!PK16 && (i0 & 0xFE000000u) == opcode) || (PK16 && (i0 & 0xFFFFFF00u) == 0xD38A4000u)) {   // v_add acc[k], x, acc[k]  |  v_pk_add_u16 acc[k], x, acc[k]
                        uint32_t vd = (i0 >> 17) & 0xFF, vs1 = (i0 >> 9) & 0xFF, src0 = i0 & 0x1FF;
                        if (PK16) {
                            const uint32_t d1 = ch.code[pc + 1];
                            vd = i0 & 0xFF;
                            vs1 = ((d1 >> 9) & 0x1FF) - 256;
                            src0 = d1 & 0x1FF;
                            if ((d1 >> 18) != 0x600u) return 30;          // op_sel_hi of both sources
                            pc++;
                        }
                        if (vd != vs1 || vd < R.acc0 || vd >= R.acc0 + KA || src0 < 256 + R.x0 || src0 >= 256 + R.x0 + R.nx()) return 30;
                        XReg &xr = x[src0 - 256 - R.x0];
                        // a pair's second register retires with the pair: find whether its instruction is still in the fifo
                        bool infl = false;
                        for (uint32_t r : fifo) if (r == src0 - 256 - R.x0 || (r + 1 == src0 - 256 - R.x0 && x[r + 1].inflight && x[r].inflight)) infl = true;
                        if (!xr.valid || infl) return 31;                                       // the read has not been waited for
                        xr.inflight = false;
                        const int64_t chunk = xr.chunk;   // what the buffer held when the read was issued (and in flight: no DMA may touch it, 36)
                        if (chunk < 0) return 32;
                        const uint64_t xrow = (uint64_t)chunk * KC + xr.ldsrow % KC;
                        const uint32_t k = vd - R.acc0;
                        if ((vals != nullptr) != xr.has_mul) return 35;                         // every entry of a valued matrix is multiplied once
                        for (uint32_t l = 0; l < wvalid; l++) {
                            T xv = xrow < ncols ? X[xrow * h + s * F + l] : T(0);
                            T &a = acc[(size_t)k * F + l];
                            if constexpr (std::is_integral<T>::value) a = (T)((uint32_t)a + (xr.has_mul ? xr.mulbits * (uint32_t)xv : (uint32_t)xv));
                            else {
                                if (xr.has_mul) {
                                    float mv;
                                    std::memcpy(&mv, &xr.mulbits, 4);
                                    volatile float prod = mv * xv;   // product and sum round separately (no FMA)
                                    xv = prod;
                                }
                                a = a + xv;
                            }
                        }
                        if (s == 0) entries_seen++;
                        pc++;
                        continue;
                    }
                    return 40;   // an instruction the stream must not contain
                }
                if (w == 0) barriers_w0 = barriers;
                else if (barriers != barriers_w0) return 43;   // the waves of a workgroup meet at the same barriers
                for (uint32_t k = 0; k < KA; k++) {
                    const uint32_t row = plan.rowmap[((size_t)ti * NW + w) * geo.ka_stride() + k];
                    if (row == 0xFFFFFFFFu) continue;
                    if (row >= nrows) return 7;
                    if (s == 0) written[row]++;
                    for (uint32_t l = 0; l < wvalid; l++) C[(size_t)row * h + s * F + l] = acc[(size_t)k * F + l];
                }
            }
        }
    }
    for (uint32_t r = 0; r < nrows; r++)
        if (written[r] != 1) return 8;
    if (entries_seen != (uint64_t)rowptr[nrows_real]) return 41;   // every stored entry exactly once, no padding
    if (S > 1)
        for (uint32_t r = 0; r < nrows_real; r++)
            for (uint32_t f = 0; f < h; f++) {
                T acc = part[(size_t)r * h + f];
                for (uint32_t c = 1; c < S; c++) {
                    const T v = part[((size_t)c * nrows_real + r) * h + f];
                    if constexpr (std::is_integral<T>::value) acc = (T)((typename std::make_unsigned<T>::type)acc + (typename std::make_unsigned<T>::type)v);
                    else acc = acc + v;
                }
                Cout[(size_t)r * h + f] = acc;
            }
    return 0;
}

extern "C" {
int lds_code_f32(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const float *X, uint32_t h, float *C,
                 uint32_t threads, uint64_t *stats, uint32_t kc, uint32_t nbuf, const float *vals, uint32_t col_splits) {
    return run_code<float>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, vals, col_splits);
}
// ... with the geometry as arguments (round 4): waves per workgroup (16 x 96 or 8 x 228 accumulators), entries per group, x-register sets
int lds_code_f32_geo(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const float *X, uint32_t h, float *C,
                     uint32_t threads, uint64_t *stats, uint32_t kc, uint32_t nbuf, const float *vals, uint32_t col_splits, uint32_t nw,
                     uint32_t gsize, uint32_t nsets, uint32_t rows_per_tile, uint32_t boundary) {
    return run_code<float>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, vals, col_splits, nw, gsize, nsets, rows_per_tile, boundary);
}
// 8-byte element types (INT64 / DBL64): the 8-wave geometry with 512-byte rows
int lds_code_f64_geo(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const double *X, uint32_t h, double *C,
                     uint32_t threads, uint64_t *stats, uint32_t kc, uint32_t nbuf, uint32_t col_splits, uint32_t gsize, uint32_t nsets, uint32_t rows_per_tile, uint32_t boundary) {
    return run_code<double>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, nullptr, col_splits, 8, gsize, nsets, rows_per_tile, boundary);
}
int lds_code_i64_geo(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const int64_t *X, uint32_t h, int64_t *C,
                     uint32_t threads, uint64_t *stats, uint32_t kc, uint32_t nbuf, uint32_t col_splits, uint32_t gsize, uint32_t nsets, uint32_t rows_per_tile, uint32_t boundary) {
    return run_code<int64_t>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, nullptr, col_splits, 8, gsize, nsets, rows_per_tile, boundary);
}
int lds_code_i32_geo(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const int32_t *X, uint32_t h, int32_t *C,
                     uint32_t threads, uint64_t *stats, uint32_t kc, uint32_t nbuf, uint32_t col_splits, uint32_t nw, uint32_t gsize,
                     uint32_t nsets, uint32_t rows_per_tile, uint32_t boundary) {
    return run_code<int32_t>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, nullptr, col_splits, nw, gsize, nsets, rows_per_tile, boundary);
}
// valued INT64 (round 5), values that fit int32: the 64-bit product from 32-bit pieces
int lds_code_i64_val_geo(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const int64_t *X, uint32_t h, int64_t *C, uint32_t threads,
                         uint64_t *stats, uint32_t kc, uint32_t nbuf, const int64_t *vals, uint32_t gsize, uint32_t nsets, uint32_t rows_per_tile, uint32_t boundary) {
    return run_code<int64_t>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, vals, 1, 8, gsize, nsets, rows_per_tile, boundary);
}
// valued DBL64 (round 5): s_mov_b32 x 2 + v_mul_f64 with the value in s[94:95]
int lds_code_f64_val_geo(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const double *X, uint32_t h, double *C, uint32_t threads,
                         uint64_t *stats, uint32_t kc, uint32_t nbuf, const double *vals, uint32_t gsize, uint32_t nsets, uint32_t rows_per_tile, uint32_t boundary) {
    return run_code<double>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, vals, 1, 8, gsize, nsets, rows_per_tile, boundary);
}
// valued INT32 (round 5): the value inline in a v_mul_lo_u32 (all values in [-16, 64]) or through an SGPR
int lds_code_i32_val_geo(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const int32_t *X, uint32_t h, int32_t *C,
                         uint32_t threads, uint64_t *stats, uint32_t kc, uint32_t nbuf, const int32_t *vals, uint32_t nw, uint32_t gsize,
                         uint32_t nsets, uint32_t rows_per_tile, uint32_t boundary) {
    return run_code<int32_t>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, vals, 1, nw, gsize, nsets, rows_per_tile, boundary);
}
int lds_code_i32(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const int32_t *X, uint32_t h, int32_t *C,
                 uint32_t threads, uint64_t *stats, uint32_t kc, uint32_t nbuf, uint32_t col_splits) {
    return run_code<int32_t>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, nullptr, col_splits);
}
// INT16 (two features to a lane), unit weights or valued (round 5: v_pk_mul_lo_u16 with the value inline or through s94); INT8 rides the same stream
int lds_code_i16_geo(const uint32_t *rowptr, const uint32_t *col, uint32_t nrows, uint32_t ncols, const int16_t *X, uint32_t h, int16_t *C,
                     uint32_t threads, uint64_t *stats, uint32_t kc, uint32_t nbuf, const int16_t *vals, uint32_t col_splits, uint32_t gsize,
                     uint32_t nsets, uint32_t rows_per_tile, uint32_t boundary) {
    return run_code<int16_t>(rowptr, col, nrows, ncols, X, h, C, threads, stats, kc, nbuf, vals, col_splits, 8, gsize, nsets, rows_per_tile, boundary);
}
}
