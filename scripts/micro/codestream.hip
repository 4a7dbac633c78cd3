// Feasibility probe: can a wave run a long STRAIGHT-LINE instruction stream (the token stream of the LDS-staged product compiled
// into machine code: `ds_read_b32 x, base offset:col*256` + `v_add_f32 acc_k, x, acc_k` per stored entry, no address
// arithmetic, no index register) from device memory at a useful rate?  Every instruction line is fetched once (no reuse in the
// instruction cache), 16 waves per CU each in a region of their own.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/codestream.hip -o /tmp/codestream -lhsa-runtime64 && /tmp/codestream
// Reports CU cycles per token for (a) every wave streaming its own REGION bytes of code once, (b) all waves looping over one
// small cached region for the same number of tokens.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define HCHECK(x) do { hsa_status_t e = (x); if (e != HSA_STATUS_SUCCESS && e != HSA_STATUS_INFO_BREAK) { printf("HSA error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

static hsa_agent_t g_gpu;
static hsa_amd_memory_pool_t g_pool;
static hsa_status_t find_gpu(hsa_agent_t a, void *) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU) { g_gpu = a; return HSA_STATUS_INFO_BREAK; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t find_pool(hsa_amd_memory_pool_t p, void *) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  bool alloc = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) { g_pool = p; return HSA_STATUS_INFO_BREAK; }
  return HSA_STATUS_SUCCESS;
}

// one group of 8 tokens = 17 dwords... (8 ds_read = 16 dwords, 1 waitcnt, 8 adds = 8 dwords) = 25 dwords = 100 bytes
constexpr uint32_t SG = 64;   // a super-group: [touch or 5 x s_nop] + 64 groups
// PAIR = 0: one ds_read_b32 per token (25 dwords per group of 8); PAIR = 1: one ds_read2st64_b32 per TWO tokens (its two 8-bit offsets
// count rows of 256 bytes -- exactly the row stride of a chunk in LDS): 17 dwords per group of 8 tokens
template <int PAIR>
__global__ void k_fill(uint32_t *code, uint64_t region_dw, uint64_t groups_per_region, uint32_t nregions, int touch) {
  constexpr uint32_t GROUP_DW = PAIR ? 17 : 25, SG_DW = 5 + SG * GROUP_DW;
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups_per_region * nregions) return;
  const uint64_t r = g / groups_per_region, gi = g % groups_per_region;
  uint32_t *p = code + r * region_dw + (gi / SG) * SG_DW + 5 + (gi % SG) * GROUP_DW;
  if (gi % SG == 0) {   // the lines 8 KiB ahead of here, one per lane, into the L2 (result unused)
    uint32_t *t = p - 5;
    if (touch) {
      t[0] = 0x802CFF2Cu; t[1] = SG_DW * 4;          // s_add_u32 s44, s44, <bytes of a super-group>
      t[2] = 0x822D802Du;                            // s_addc_u32 s45, s45, 0
      t[3] = 0xDC508000u; t[4] = 0x052C0006u;        // global_load_dword v5, v6, s[44:45]
    } else {
      for (int i = 0; i < 5; i++) t[i] = 0xBF800000u;
    }
  }
  uint32_t h = (uint32_t)(g * 2654435761u);
  const uint32_t xs = (gi & 1) ? 20 : 12;            // x registers of this group; the adds below use the OTHER set (previous group)
  const uint32_t xa = (gi & 1) ? 12 : 20;
  uint32_t q = 0;
  if (PAIR) {
    for (int i = 0; i < 4; i++) {
      h = h * 1664525u + 1013904223u;
      const uint32_t r0 = (h >> 8) & 0xff, r1 = (h >> 16) & 0xff;
      p[q++] = 0xD8700000u | (r1 << 8) | r0;         // ds_read2st64_b32 v[xs+2i : xs+2i+1], v4 offset0:r0 offset1:r1
      p[q++] = ((xs + 2 * i) << 24) | 4u;
    }
    p[q++] = 0xBF8CC47Fu;                            // s_waitcnt lgkmcnt(4): the previous group's four reads are in
  } else {
    for (int i = 0; i < 8; i++) {
      h = h * 1664525u + 1013904223u;
      const uint32_t colrow = (h >> 8) & 0xff;       // row of the chunk in LDS
      p[q++] = 0xD86C0000u | (colrow << 8);          // ds_read_b32 v[xs+i], v4 offset:colrow*256
      p[q++] = ((xs + i) << 24) | 4u;
    }
    p[q++] = 0xBF8CC87Fu;                            // s_waitcnt lgkmcnt(8): the previous group's reads are in
  }
  for (int i = 0; i < 8; i++) {
    h = h * 1664525u + 1013904223u;
    const uint32_t k = 28 + (h >> 10) % 96;
    p[q++] = 0x02000000u | (k << 17) | (k << 9) | (256 + xa + i);   // v_add_f32 v[k], v[xa+i], v[k]
  }
  if (gi == groups_per_region - 1) {                 // region epilogue behind the last group: wait, return
    p[q++] = 0xBF8C0070u;                            // s_waitcnt vmcnt(0) lgkmcnt(0)
    p[q++] = 0xBE801D1Eu;                            // s_setpc_b64 s[30:31]
  }
}

#define CLOB "v4","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29", \
  "v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49", \
  "v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69", \
  "v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89", \
  "v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109", \
  "v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127", \
  "v5","v6","s30","s31","s40","s41","s42","s44","s45","scc","memory"

// every wave runs `reps` times through the region at code + region_index * region_bytes
__global__ __launch_bounds__(1024) void k_run(const uint32_t *code, uint64_t region_bytes, int own_region, int reps, float *out) {
  extern __shared__ uint32_t lds[];
  for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t idx = own_region ? (uint64_t)blockIdx.x * 16 + wave : 0;
  const uint64_t addr = (uint64_t)code + idx * region_bytes;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
  const uint32_t lane4 = (threadIdx.x & 63) * 4;
  asm volatile(
    "v_mov_b32 v4, %[l4]\n v_lshlrev_b32 v6, 5, %[l4]\n"
    "s_mov_b32 s40, %[lo]\n s_mov_b32 s41, %[hi]\n s_mov_b32 s42, %[reps]\n"
    "s_add_u32 s44, s40, 0x2000\n s_addc_u32 s45, s41, 0\n"
    "L_again_%=:\n"
    "s_swappc_b64 s[30:31], s[40:41]\n"
    "s_sub_u32 s42, s42, 1\n s_cmp_lg_u32 s42, 0\n s_cbranch_scc1 L_again_%=\n"
    : : [l4] "v"(lane4), [lo] "s"(lo), [hi] "s"(hi), [reps] "s"(reps) : CLOB);
  if (reps < 0) out[threadIdx.x] = 1.f;
}

int main(int argc, char **argv) {
  const bool use_hipmalloc = argc > 1 && atoi(argv[1]) == 1;   // 1: plain hipMalloc instead of the executable HSA pool
  CHECK(hipSetDevice(0));
  float *out;
  CHECK(hipMalloc(&out, 4096));
  const uint32_t nregions = 256 * 16;
  const uint64_t groups = 2560;                       // 20 480 tokens per wave: about half a tile-slice of the bench workload
  const uint64_t region_dw = (groups / SG) * (5 + SG * 25) + 8 + 4096, region_bytes = region_dw * 4;   // (+ slack the last touches may read)
  const size_t total = (size_t)region_bytes * nregions;
  uint32_t *code = nullptr;
  if (use_hipmalloc) {
    CHECK(hipMalloc((void **)&code, total));
  } else {
    HCHECK(hsa_init());
    HCHECK(hsa_iterate_agents(find_gpu, nullptr));
    HCHECK(hsa_amd_agent_iterate_memory_pools(g_gpu, find_pool, nullptr));
    HCHECK(hsa_amd_memory_pool_allocate(g_pool, total, HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG, (void **)&code));
    HCHECK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, code));
  }
  printf("code: %zu MiB at %p (%s), %llu bytes per wave\n", total >> 20, (void *)code, use_hipmalloc ? "hipMalloc" : "HSA executable pool",
         (unsigned long long)region_bytes);
  const uint64_t ngroups = groups * nregions;
  for (int mode = 0; mode < 4; mode++) {
  const int touch = mode & 1, pair = mode >> 1;
  printf("--- %s, in-stream touches %s\n", pair ? "ds_read2st64_b32 per TWO tokens (8.5 bytes of code per token)" : "ds_read_b32 per token (12.5 bytes of code per token)",
         touch ? "ON (64 lines from 8 KiB ahead, once per 64 groups)" : "off");
  if (pair) hipLaunchKernelGGL(k_fill<1>, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, 0, code, region_dw, groups, nregions, touch);
  else hipLaunchKernelGGL(k_fill<0>, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, 0, code, region_dw, groups, nregions, touch);
  CHECK(hipDeviceSynchronize());
  CHECK(hipFuncSetAttribute((const void *)k_run, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int own = 1; own >= 0; own--) {
    for (int r = 0; r < 3; r++) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_run, dim3(256), dim3(1024), 65536, 0, code, region_bytes, own, 1, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double tokens_per_cu = (double)groups * 8 * 16;
      printf("%s: %8.3f ms  %6.2f CU cycles per token (2.4 GHz)\n", own ? "own region per wave (streamed once)" : "all waves in region 0 (cached)      ", ms,
             ms * 1e-3 * 2.4e9 / tokens_per_cu);
    }
  }
  }
  return 0;
}
