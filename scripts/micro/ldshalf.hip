// Does an LDS read whose EXEC mask covers only half of the wave cost half the LDS-array cycles?  (Round 4: the code-stream product is
// bound by the LDS array -- 2 cycles per ds_read_b32 of a 256-byte row.  If a ds_read_b64 under EXEC = lanes 0..31 (one lane group of
// 32 x 8 bytes = the same 256-byte row) takes ONE cycle, a wave could keep two rows per accumulator register pair -- one per half -- and
// read every entry's row in half the LDS time.)
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/ldshalf.hip -o /tmp/ldshalf && /tmp/ldshalf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: ds_read_b32, all lanes                      (256 B per instruction)
// MODE 1: ds_read_b64, all lanes                      (512 B)
// MODE 2: ds_read_b64, EXEC = lanes 0..31             (256 B)
// MODE 3: ds_read_b64, EXEC alternating halves, switched every 4 reads
// MODE 4: ds_read2st64_b32, all lanes                 (2 x 256 B)
// MODE 5: MODE 3 + a v_pk_add_f32 per read under the same half mask (the product's entry: read + add)
// MODE 6: MODE 4 + two v_add_f32 per read2 (today's entry pair)
// MODE 7: ds_read_b128, EXEC = lanes 0..15            (256 B by 16 lanes)
template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, float *out) {
    extern __shared__ char lds[];
    const uint32_t lane = threadIdx.x & 63;
    uint32_t a32 = lane * 4, a64 = (lane & 31) * 8 + (lane >> 5) * 16384, a128 = (lane & 15) * 16;
    if (MODE == 1) a64 = lane * 8;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0)
            asm volatile("ds_read_b32 v40, %0 offset:256\n ds_read_b32 v41, %0 offset:2816\n ds_read_b32 v42, %0 offset:9472\n ds_read_b32 v43, %0 offset:1024\n"
                         "ds_read_b32 v44, %0 offset:7424\n ds_read_b32 v45, %0 offset:12544\n ds_read_b32 v46, %0 offset:4352\n ds_read_b32 v47, %0 offset:15104\n"
                         "s_waitcnt lgkmcnt(4)\n" : : "v"(a32) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
        if (MODE == 1 || MODE == 2)
            asm volatile("ds_read_b64 v[40:41], %0 offset:256\n ds_read_b64 v[42:43], %0 offset:2816\n ds_read_b64 v[44:45], %0 offset:9472\n ds_read_b64 v[46:47], %0 offset:1024\n"
                         "ds_read_b64 v[48:49], %0 offset:7424\n ds_read_b64 v[50:51], %0 offset:12544\n ds_read_b64 v[52:53], %0 offset:4352\n ds_read_b64 v[54:55], %0 offset:15104\n"
                         "s_waitcnt lgkmcnt(4)\n" : : "v"(a64) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "memory");
        if (MODE == 3)
            asm volatile("s_mov_b64 exec, 0xffffffff\n"
                         "ds_read_b64 v[40:41], %0 offset:256\n ds_read_b64 v[42:43], %0 offset:2816\n ds_read_b64 v[44:45], %0 offset:9472\n ds_read_b64 v[46:47], %0 offset:1024\n"
                         "s_not_b64 exec, exec\n"
                         "ds_read_b64 v[40:41], %0 offset:7424\n ds_read_b64 v[42:43], %0 offset:12544\n ds_read_b64 v[44:45], %0 offset:4352\n ds_read_b64 v[46:47], %0 offset:15104\n"
                         "s_mov_b64 exec, -1\n"
                         "s_waitcnt lgkmcnt(4)\n" : : "v"(a64) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
        if (MODE == 4)
            asm volatile("ds_read2st64_b32 v[40:41], %0 offset0:1 offset1:11\n ds_read2st64_b32 v[42:43], %0 offset0:37 offset1:4\n"
                         "ds_read2st64_b32 v[44:45], %0 offset0:29 offset1:49\n ds_read2st64_b32 v[46:47], %0 offset0:17 offset1:59\n"
                         "s_waitcnt lgkmcnt(2)\n" : : "v"(a32) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
        if (MODE == 5)
            asm volatile("s_mov_b64 exec, 0xffffffff\n"
                         "ds_read_b64 v[40:41], %0 offset:256\n ds_read_b64 v[42:43], %0 offset:2816\n ds_read_b64 v[44:45], %0 offset:9472\n ds_read_b64 v[46:47], %0 offset:1024\n"
                         "s_not_b64 exec, exec\n"
                         "ds_read_b64 v[48:49], %0 offset:7424\n ds_read_b64 v[50:51], %0 offset:12544\n ds_read_b64 v[52:53], %0 offset:4352\n ds_read_b64 v[54:55], %0 offset:15104\n"
                         "s_not_b64 exec, exec\n"
                         "s_waitcnt lgkmcnt(4)\n"
                         "v_pk_add_f32 v[60:61], v[40:41], v[60:61]\n v_pk_add_f32 v[62:63], v[42:43], v[62:63]\n v_pk_add_f32 v[64:65], v[44:45], v[64:65]\n v_pk_add_f32 v[66:67], v[46:47], v[66:67]\n"
                         "s_not_b64 exec, exec\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "v_pk_add_f32 v[68:69], v[48:49], v[68:69]\n v_pk_add_f32 v[70:71], v[50:51], v[70:71]\n v_pk_add_f32 v[72:73], v[52:53], v[72:73]\n v_pk_add_f32 v[74:75], v[54:55], v[74:75]\n"
                         "s_mov_b64 exec, -1\n"
                         : : "v"(a64) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
                             "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "memory");
        if (MODE == 6)
            asm volatile("ds_read2st64_b32 v[40:41], %0 offset0:1 offset1:11\n ds_read2st64_b32 v[42:43], %0 offset0:37 offset1:4\n"
                         "ds_read2st64_b32 v[44:45], %0 offset0:29 offset1:49\n ds_read2st64_b32 v[46:47], %0 offset0:17 offset1:59\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         "v_add_f32 v60, v40, v60\n v_add_f32 v61, v41, v61\n v_add_f32 v62, v42, v62\n v_add_f32 v63, v43, v63\n"
                         "v_add_f32 v64, v44, v64\n v_add_f32 v65, v45, v65\n v_add_f32 v66, v46, v66\n v_add_f32 v67, v47, v67\n"
                         : : "v"(a32) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "memory");
        if (MODE == 7)
            asm volatile("s_mov_b64 exec, 0xffff\n"
                         "ds_read_b128 v[40:43], %0 offset:256\n ds_read_b128 v[44:47], %0 offset:2816\n ds_read_b128 v[48:51], %0 offset:9472\n ds_read_b128 v[52:55], %0 offset:1024\n"
                         "ds_read_b128 v[40:43], %0 offset:7424\n ds_read_b128 v[44:47], %0 offset:12544\n ds_read_b128 v[48:51], %0 offset:4352\n ds_read_b128 v[52:55], %0 offset:15104\n"
                         "s_mov_b64 exec, -1\n"
                         "s_waitcnt lgkmcnt(4)\n" : : "v"(a128) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n" ::: "memory");
    if (iters < 0) out[threadIdx.x] = ((float *)lds)[threadIdx.x];
}

template <int MODE> static void run(const char *name, int waves, int per_iter, int bytes_per_instr, float *out) {
    CHECK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 65536, 0, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    const double instr = (double)iters * per_iter * waves;   // LDS instructions per CU
    printf("%-58s %2d waves: %7.3f ms  %5.2f CU cycles per LDS instruction (2.4 GHz), %6.1f B/clk/CU\n", name, waves, best, best * 1e-3 * 2.4e9 / instr,
           instr * bytes_per_instr / (best * 1e-3 * 2.4e9));
    fflush(stdout);
}

int main() {
    CHECK(hipSetDevice(0));
    float *out;
    CHECK(hipMalloc(&out, 4096));
    for (int waves : {8, 16 / 2}) {
        run<0>("ds_read_b32 (256 B)", waves, 8, 256, out);
        run<1>("ds_read_b64 (512 B)", waves, 8, 512, out);
        run<2>("ds_read_b64 all lanes, two halves 16 KiB apart (512 B)", waves, 8, 512, out);
        run<3>("ds_read_b64 under a half EXEC mask (256 B), alternating", waves, 8, 256, out);
        run<4>("ds_read2st64_b32 (2 x 256 B)", waves, 4, 512, out);
        run<5>("half-mask ds_read_b64 + v_pk_add_f32 per entry", waves, 8, 256, out);
        run<6>("ds_read2st64_b32 + 2 v_add_f32 (today's pair)", waves, 4, 512, out);
        run<7>("ds_read_b128 under EXEC = 16 lanes (256 B)", waves, 8, 256, out);
        break;
    }
    return 0;
}
