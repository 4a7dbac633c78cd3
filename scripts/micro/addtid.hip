// Micro-test: ds_read_addtid_b32 on gfx950 -- (1) which bits of M0 form the address, (2) what a token's LDS read costs with it
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/addtid.hip -o /tmp/addtid && /tmp/addtid
// Form A (the product kernel): v_bfi_b32 addr + ds_read_b32.   Form B: s_and_b32 m0 + ds_read_addtid_b32 (address = M0 + offset + lane*4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// (1) semantics: LDS dword i holds i; read with M0 = m0val and the given immediate offset; out[lane] = value read
template <int OFF>
__global__ void k_sem(uint32_t m0val, uint32_t *out) {
  extern __shared__ uint32_t lds[];
  for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = i;
  __syncthreads();
  uint32_t v;
  asm volatile("s_mov_b32 m0, %1\n s_nop 4\n ds_read_addtid_b32 %0 offset:%2\n s_waitcnt lgkmcnt(0)\n" : "=v"(v) : "s"(m0val), "n"(OFF) : "memory");
  out[threadIdx.x] = v;
}

#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29", \
  "v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49", \
  "v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69", \
  "v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89", \
  "v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109", \
  "v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127", \
  "s40","s41","s42","s43","s44","s45","s46","s47","s60","s61","s62","scc","memory"

#define SETUP \
  "s_mov_b32 s60, %[it]\n s_mov_b32 s62, 0xff00\n" \
  "s_mov_b32 s40, 0x0300+3\n s_mov_b32 s41, 0x1100+50\n s_mov_b32 s42, 0x7f00+17\n s_mov_b32 s43, 0x2a00+90\n" \
  "s_mov_b32 s44, 0xf000+8\n s_mov_b32 s45, 0x6100+61\n s_mov_b32 s46, 0x0900+33\n s_mov_b32 s47, 0xc400+72\n" \
  "v_mov_b32 v110, 0xff00\n v_lshlrev_b32 v111, 2, %[lane]\n"
#define RD_A(s, v) "v_bfi_b32 " v ", v110, " s ", v111\n ds_read_b32 " v ", " v "\n"
#define RD_B(s, v) "s_and_b32 m0, " s ", s62\n s_nop 0\n ds_read_addtid_b32 " v "\n"
#define RD_C(s, v) "s_and_b32 m0, " s ", s62\n ds_read_addtid_b32 " v "\n"     /* without the wait state: is it needed? (timing only) */
#define ADD(s, v) "s_set_gpr_idx_idx " s "\n v_add_f32 v28, " v ", v28\n"
#define BATCH(RD) \
  RD("s40","v12") RD("s41","v13") RD("s42","v14") RD("s43","v15") RD("s44","v16") RD("s45","v17") RD("s46","v18") RD("s47","v19") \
  "s_waitcnt lgkmcnt(0)\n s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n" \
  ADD("s40","v12") ADD("s41","v13") ADD("s42","v14") ADD("s43","v15") ADD("s44","v16") ADD("s45","v17") ADD("s46","v18") ADD("s47","v19") \
  "s_set_gpr_idx_off\n"
#define KERNEL(NAME, RD) \
__global__ __launch_bounds__(1024) void NAME(int it, float *out) { \
  extern __shared__ uint32_t lds[]; \
  for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = 0; \
  __syncthreads(); \
  const uint32_t lane = threadIdx.x & 63; \
  asm volatile(SETUP "L_loop_%=:\n" BATCH(RD) \
    "s_sub_u32 s60, s60, 1\n s_cmp_lg_u32 s60, 0\n s_cbranch_scc1 L_loop_%=\n" \
    : : [it] "s"(it), [lane] "v"(lane) : CLOB); \
  if (it < 0) out[threadIdx.x] = 1.f; \
}
KERNEL(kA, RD_A)
KERNEL(kB, RD_B)
KERNEL(kC, RD_C)

// ---- structure variants of form A (same instructions per token, different order) ----
#define RDX(s, v) "v_bfi_b32 " v ", v110, " s ", v111\n ds_read_b32 " v ", " v "\n"
#define READS_X "" RDX("s40","v12") RDX("s41","v13") RDX("s42","v14") RDX("s43","v15") RDX("s44","v16") RDX("s45","v17") RDX("s46","v18") RDX("s47","v19")
#define READS_Y "" RDX("s40","v20") RDX("s41","v21") RDX("s42","v22") RDX("s43","v23") RDX("s44","v24") RDX("s45","v25") RDX("s46","v26") RDX("s47","v27")
#define ADDS_X "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n" ADD("s40","v12") ADD("s41","v13") ADD("s42","v14") ADD("s43","v15") ADD("s44","v16") ADD("s45","v17") ADD("s46","v18") ADD("s47","v19") "s_set_gpr_idx_off\n"
#define ADDS_Y "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n" ADD("s40","v20") ADD("s41","v21") ADD("s42","v22") ADD("s43","v23") ADD("s44","v24") ADD("s45","v25") ADD("s46","v26") ADD("s47","v27") "s_set_gpr_idx_off\n"
// P: software-pipelined as the product kernel: wait, reads of the next batch, adds of the current one (two batches per iteration)
#define BODY_P "s_waitcnt lgkmcnt(0)\n" READS_Y ADDS_X "s_waitcnt lgkmcnt(0)\n" READS_X ADDS_Y
// Q: adds start as soon as the first read is back (in-order LDS returns: lgkmcnt(7), (6), ...) -- not available to the product
//    kernel as it is (its scalar token loads share the counter and return out of order)
#define ADDW(n, s, v) "s_waitcnt lgkmcnt(" #n ")\n" ADD(s, v)
#define BODY_Q READS_X "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n" ADDW(7,"s40","v12") ADDW(6,"s41","v13") ADDW(5,"s42","v14") ADDW(4,"s43","v15") ADDW(3,"s44","v16") ADDW(2,"s45","v17") ADDW(1,"s46","v18") ADDW(0,"s47","v19") "s_set_gpr_idx_off\n" \
               READS_Y "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n" ADDW(7,"s40","v20") ADDW(6,"s41","v21") ADDW(5,"s42","v22") ADDW(4,"s43","v23") ADDW(3,"s44","v24") ADDW(2,"s45","v25") ADDW(1,"s46","v26") ADDW(0,"s47","v27") "s_set_gpr_idx_off\n"
// R: pipelined AND partial waits: reads of the next batch in flight while the current one's adds wait read by read
#define BODY_R READS_Y "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n" ADDW(15,"s40","v12") ADDW(14,"s41","v13") ADDW(13,"s42","v14") ADDW(12,"s43","v15") ADDW(11,"s44","v16") ADDW(10,"s45","v17") ADDW(9,"s46","v18") ADDW(8,"s47","v19") "s_set_gpr_idx_off\n" \
               READS_X "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n" ADDW(15,"s40","v20") ADDW(14,"s41","v21") ADDW(13,"s42","v22") ADDW(12,"s43","v23") ADDW(11,"s44","v24") ADDW(10,"s45","v25") ADDW(9,"s46","v26") ADDW(8,"s47","v27") "s_set_gpr_idx_off\n"
// S: the index mode left ON across the whole loop (the address v_bfi then runs under it too: src1 = s-register, not indexed)
#define ADDS_X_NOON ADD("s40","v12") ADD("s41","v13") ADD("s42","v14") ADD("s43","v15") ADD("s44","v16") ADD("s45","v17") ADD("s46","v18") ADD("s47","v19")
#define KERNEL2(NAME, PRO, BODY, TOKENS_PER_IT) \
__global__ __launch_bounds__(1024) void NAME(int it, float *out) { \
  extern __shared__ uint32_t lds[]; \
  for (uint32_t i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = 0; \
  __syncthreads(); \
  const uint32_t lane = threadIdx.x & 63; \
  asm volatile(SETUP PRO "L_loop_%=:\n" BODY \
    "s_sub_u32 s60, s60, 1\n s_cmp_lg_u32 s60, 0\n s_cbranch_scc1 L_loop_%=\n s_waitcnt lgkmcnt(0)\n" \
    : : [it] "s"(it), [lane] "v"(lane) : CLOB); \
  if (it < 0) out[threadIdx.x] = 1.f; \
}
KERNEL2(kP, READS_X, BODY_P, 16)
KERNEL2(kQ, "", BODY_Q, 16)
KERNEL2(kR, READS_X, BODY_R, 16)

int main() {
  uint32_t *out;
  CHECK(hipMalloc(&out, 4096));
  std::vector<uint32_t> h(64);
  CHECK(hipFuncSetAttribute((const void *)k_sem<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CHECK(hipFuncSetAttribute((const void *)k_sem<0xff01>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CHECK(hipFuncSetAttribute((const void *)k_sem<0x100>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  struct T { uint32_t m0; int off; } tests[] = {{0, 0}, {0x100, 0}, {0xff00, 0}, {0x10000, 0}, {0x1ff00, 0}, {0xabc10300, 0}, {0x3ff, 0xff01}, {0x0, 0x100}};
  for (auto &t : tests) {
    if (t.off == 0) hipLaunchKernelGGL(k_sem<0>, dim3(1), dim3(64), 131072, 0, t.m0, out);
    else if (t.off == 0xff01) hipLaunchKernelGGL(k_sem<0xff01>, dim3(1), dim3(64), 131072, 0, t.m0, out);
    else hipLaunchKernelGGL(k_sem<0x100>, dim3(1), dim3(64), 131072, 0, t.m0, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), out, 256, hipMemcpyDeviceToHost));
    printf("M0 = 0x%08x offset 0x%04x: lane0 reads dword 0x%x (byte 0x%x), lane1 0x%x, lane63 0x%x\n", t.m0, t.off, h[0], h[0] * 4, h[1], h[63]);
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int it = 20000;
  struct V { const char *name; void (*fn)(int, float *); int tok; } vs[] = {{"A v_bfi + ds_read_b32", kA, 8}, {"B s_and m0 + s_nop + ds_read_addtid", kB, 8}, {"C s_and m0 + ds_read_addtid (no nop)", kC, 8},
    {"P form A software-pipelined (as the kernel)", kP, 16}, {"Q form A, adds wait read by read", kQ, 16}, {"R pipelined + read-by-read waits", kR, 16}};
  for (auto &v : vs) {
    CHECK(hipFuncSetAttribute((const void *)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipLaunchKernelGGL(v.fn, dim3(256), dim3(1024), 131072, 0, 100, (float *)out);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(v.fn, dim3(256), dim3(1024), 131072, 0, it, (float *)out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * 2.4e9;
    printf("%-40s %8.3f ms  %6.2f CU cycles per token (16 waves, 8 tokens per batch)\n", v.name, best, cyc / ((double)it * v.tok * 16));
  }
  return 0;
}
