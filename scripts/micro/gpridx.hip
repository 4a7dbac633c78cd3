// Micro-benchmark: what does one (accumulator-index write, indexed add) pair of the LDS-staged kernel cost on gfx950?
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/gpridx.hip -o /tmp/gpridx && /tmp/gpridx
// 256 workgroups x 16 waves (4 per SIMD, 128 VGPRs each, like k_lds_spmm_*_w16); every wave runs ITER x 16 pairs.
// Reported: SIMD cycles per pair at 4 (and 2, 1) waves per SIMD, for
//   A  s_set_gpr_idx_idx + v_add_f32 (the product kernel's form)      B  v_add_f32 only (index fixed)
//   C  s_mov_b32 m0 + v_add_f32 (M0 written directly, mode bits in the value)   D  s_set_gpr_idx_idx only
//   E  A with an independent VALU instruction between the two         F  one index write, two indexed adds
//   G  v_add_f32 chain on ONE fixed register without index mode (plain dependent adds)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29", \
  "v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49", \
  "v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69", \
  "v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89", \
  "v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109", \
  "v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127", \
  "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s60","s61","scc","memory"

#define PAIR16(IDX, MID, ADD) \
  IDX("s40") MID ADD IDX("s41") MID ADD IDX("s42") MID ADD IDX("s43") MID ADD IDX("s44") MID ADD IDX("s45") MID ADD IDX("s46") MID ADD IDX("s47") MID ADD \
  IDX("s48") MID ADD IDX("s49") MID ADD IDX("s50") MID ADD IDX("s51") MID ADD IDX("s52") MID ADD IDX("s53") MID ADD IDX("s54") MID ADD IDX("s55") MID ADD

#define SETUP(orv) \
  "s_mov_b32 s60, %[it]\n" \
  "s_mov_b32 s40, " #orv "+3\n s_mov_b32 s41, " #orv "+50\n s_mov_b32 s42, " #orv "+17\n s_mov_b32 s43, " #orv "+90\n" \
  "s_mov_b32 s44, " #orv "+8\n s_mov_b32 s45, " #orv "+61\n s_mov_b32 s46, " #orv "+33\n s_mov_b32 s47, " #orv "+72\n" \
  "s_mov_b32 s48, " #orv "+1\n s_mov_b32 s49, " #orv "+44\n s_mov_b32 s50, " #orv "+25\n s_mov_b32 s51, " #orv "+83\n" \
  "s_mov_b32 s52, " #orv "+12\n s_mov_b32 s53, " #orv "+57\n s_mov_b32 s54, " #orv "+39\n s_mov_b32 s55, " #orv "+95\n" \
  "v_mov_b32 v110, 1.0\n v_mov_b32 v111, 0\n"

#define IDX_SET(s) "s_set_gpr_idx_idx " s "\n"
#define IDX_MOV(s) "s_mov_b32 m0, " s "\n"
#define IDX_NONE(s) ""
#define ADD1 "v_add_f32 v10, v110, v10\n"
#define ADD2 "v_add_f32 v10, v110, v10\n v_add_f32 v11, v110, v11\n"
#define MID0 ""
#define MIDV "v_bfi_b32 v111, v110, v111, v111\n"

#define KERNEL(NAME, ORV, IDX, MID, ADD, ON) \
__global__ __launch_bounds__(1024) void NAME(int it, float *out) { \
  asm volatile(SETUP(ORV) ON \
    "L_loop_%=:\n" PAIR16(IDX, MID, ADD) \
    "s_sub_u32 s60, s60, 1\n s_cmp_lg_u32 s60, 0\n s_cbranch_scc1 L_loop_%=\n" \
    "s_set_gpr_idx_off\n" \
    : : [it] "s"(it) : CLOB); \
  if (it < 0) out[threadIdx.x] = 1.f; \
}
KERNEL(kA, 0, IDX_SET, MID0, ADD1, "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n")
KERNEL(kB, 0, IDX_NONE, MID0, ADD1, "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n")
KERNEL(kC, 0xA000, IDX_MOV, MID0, ADD1, "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n")
KERNEL(kD, 0, IDX_SET, MID0, "", "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n")
KERNEL(kE, 0, IDX_SET, MIDV, ADD1, "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n")
KERNEL(kF, 0, IDX_SET, MID0, ADD2, "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n")
KERNEL(kG, 0, IDX_NONE, MID0, ADD1, "")
// H: the index write placed BEFORE an independent VALU instruction and the add after it (software-pipelined by one)
KERNEL(kH, 0, IDX_SET, MIDV MIDV, ADD1, "s_set_gpr_idx_on s40, gpr_idx(SRC1,DST)\n")

int main() {
  float *out;
  CHECK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int it = 20000;
  struct V { const char *name; void (*fn)(int, float *); } vs[] = {
    {"A idx+add", kA}, {"B add only (indexed, fixed)", kB}, {"C s_mov m0+add", kC}, {"D idx only", kD},
    {"E idx, bfi, add", kE}, {"F idx, 2 adds", kF}, {"G plain dependent add", kG}, {"H idx, 2 bfi, add", kH}};
  int clk_khz = 0;
  CHECK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  printf("clock %d kHz\n", clk_khz);
  for (int threads : {1024, 512, 256}) {
    for (auto &v : vs) {
      hipLaunchKernelGGL(v.fn, dim3(256), dim3(threads), 0, 0, 100, out);
      CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int r = 0; r < 3; r++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(v.fn, dim3(256), dim3(threads), 0, 0, it, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double cyc = best * 1e-3 * clk_khz * 1e3;           // cycles of the launch
      const double pairs_per_simd = (double)it * 16 * (threads / 256);  // waves per SIMD x pairs per wave
      printf("%4d threads (%d waves/SIMD)  %-30s %8.3f ms  %6.2f SIMD cycles per pair (per wave: %6.2f)\n", threads, threads / 256, v.name, best,
             cyc / pairs_per_simd, cyc / ((double)it * 16));
    }
  }
  return 0;
}
