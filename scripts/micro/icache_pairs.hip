// Round 5 probe 2: do two compute units share an instruction cache, and which two?  A straight-line stream fetched once per wave runs at
// ~5.1 bytes of code per clock and CU (codestream_b64.hip: the streamed case of every mode sits on cycles = bytes x 0.195); if the
// workgroups on two CUs that share an instruction cache run the SAME stream in step, each line is filled once for both.  Here the
// workgroups of an XCD (round-robin dispatch: b & 7 = XCD, b >> 3 = index inside it) pair up by one bit of that index and a pair streams
// one region; "pair bit 0" = every workgroup its own region.  mode 2 of codestream_b64.hip (ds_read_b64, no swap: the fetch-bound case).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/icache_pairs.hip -o /tmp/icache_pairs -lhsa-runtime64 && /tmp/icache_pairs
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


// a super-group: touch (5 dwords: 16 lines = 2 KiB, from 4 KiB ahead) + SG groups of at most 2 KiB together
// group g: G / 2 x ds_read2st64_b32 into x-set g % NS; s_waitcnt lgkmcnt((NS - 1) * G / 2); G x v_add_f32 from x-set (g + 1) % NS
__global__ void k_fill(uint32_t *code, uint64_t region_dw, uint64_t groups_per_region, uint32_t nregions, uint32_t G, uint32_t NS, uint32_t x0,
                       uint32_t acc0, uint32_t nacc, uint32_t SG, uint32_t mode) {
  const uint32_t GROUP_DW = 2 * G + 1 + ((mode == 1 || mode == 3) ? G / 2 : 0), SG_DW = 5 + SG * GROUP_DW;
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups_per_region * nregions) return;
  const uint64_t r = g / groups_per_region, gi = g % groups_per_region;
  uint32_t *p = code + r * region_dw + (gi / SG) * SG_DW + 5 + (gi % SG) * GROUP_DW;
  if (gi % SG == 0) {   // 16 lines 4 KiB ahead of here into the L2 (result unused)
    uint32_t *t = p - 5;
    t[0] = 0x802CFF2Cu; t[1] = SG_DW * 4;          // s_add_u32 s44, s44, <bytes of a super-group>
    t[2] = 0x822D802Du;                            // s_addc_u32 s45, s45, 0
    t[3] = 0xDC508000u; t[4] = 0x052C0006u;        // global_load_dword v5, v6, s[44:45]
  }
  uint32_t h = (uint32_t)(g * 2654435761u);
  const uint32_t xs = x0 + G * (uint32_t)(gi % NS), xa = x0 + G * (uint32_t)((gi + 1) % NS);
  uint32_t q = 0;
  for (uint32_t i = 0; i < G / 2; i++) {
    h = h * 1664525u + 1013904223u;
    const uint32_t r0 = (h >> 8) & 0xff, r1 = (h >> 16) & 0xff;
    if (mode == 0) {
      p[q++] = 0xD8700000u | (r1 << 8) | r0;       // ds_read2st64_b32 v[xs+2i : xs+2i+1], v4 offset0:r0 offset1:r1
      p[q++] = ((xs + 2 * i) << 24) | 4u;
    } else {
      p[q++] = 0xD8EC0000u | ((r0 & 0x7f) << 8);   // ds_read_b64 v[xs+2i : xs+2i+1], v7 offset:r0 * 256   (lanes 32-63: + 3 rows, in v7)
      p[q++] = ((xs + 2 * i) << 24) | 7u;
    }
  }
  p[q++] = 0xBF8CC07Fu | (((NS - 1) * G / 2) << 8);   // s_waitcnt lgkmcnt((NS - 1) * G / 2)
  if (mode == 1) for (uint32_t i = 0; i < G / 2; i++) p[q++] = 0x7E00B500u | ((xa + 2 * i) << 17) | (256 + xa + 2 * i + 1);   // v_permlane32_swap_b32 xa, xb
  if (mode == 3) for (uint32_t i = 0; i < G / 2; i++) p[q++] = 0x7E000300u | (8u << 17) | (256 + xa + 2 * i + 1);             // v_mov_b32 v8, xb
  for (uint32_t i = 0; i < G; i++) {
    h = h * 1664525u + 1013904223u;
    const uint32_t k = acc0 + (h >> 10) % nacc;
    p[q++] = 0x02000000u | (k << 17) | (k << 9) | (256 + xa + i);   // v_add_f32 v[k], v[xa+i], v[k]
  }
  if (gi == groups_per_region - 1) {
    p[q++] = 0xBF8C0070u;                            // s_waitcnt vmcnt(0) lgkmcnt(0)
    p[q++] = 0xBE801D1Eu;                            // s_setpc_b64 s[30:31]
  }
}

#define V8(a) "v" #a "0","v" #a "1","v" #a "2","v" #a "3","v" #a "4","v" #a "5","v" #a "6","v" #a "7","v" #a "8","v" #a "9"
#define CLOB128 "v4","v5","v6","v8","v9",V8(1),V8(2),V8(3),V8(4),V8(5),V8(6),V8(7),V8(8),V8(9),V8(10),V8(11),"v120","v121","v122","v123","v124","v125","v126","v127"
#define CLOB256 CLOB128,"v128","v129",V8(13),V8(14),V8(15),V8(16),V8(17),V8(18),V8(19),V8(20),V8(21),V8(22),V8(23),V8(24),"v250","v251","v252","v253","v254","v255"
#define CLOBS "s30","s31","s40","s41","s42","s44","s45","scc","memory"

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_run(const uint32_t *code, uint64_t region_bytes, int own_region, int reps, float *out, uint32_t pairbit) {
  extern __shared__ uint32_t lds[];
  for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  const uint32_t blk = pairbit ? (((wi & ~pairbit) << 3) | xcd) : blockIdx.x;   // (workgroups wi and wi ^ pairbit of an XCD stream one region)
  const uint64_t idx = own_region ? (uint64_t)blk * NW + wave : 0;
  const uint64_t addr = (uint64_t)code + idx * region_bytes;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
  const uint32_t lane4 = (threadIdx.x & 63) * 4;
  const uint32_t lane8 = (threadIdx.x & 31) * 8 + ((threadIdx.x >> 5) & 1) * 768;   // lanes 32-63 read the row three rows on
#define BODY \
    "v_mov_b32 v4, %[l4]\n v_and_b32 v6, 60, %[l4]\n v_lshlrev_b32 v6, 5, v6\n v_mov_b32 v7, %[l8]\n" \
    "s_mov_b32 s40, %[lo]\n s_mov_b32 s41, %[hi]\n s_mov_b32 s42, %[reps]\n" \
    "s_add_u32 s44, s40, 0x1000\n s_addc_u32 s45, s41, 0\n" \
    "L_again_%=:\n" \
    "s_swappc_b64 s[30:31], s[40:41]\n" \
    "s_sub_u32 s42, s42, 1\n s_cmp_lg_u32 s42, 0\n s_cbranch_scc1 L_again_%=\n"
  if (NW == 16) asm volatile(BODY : : [l4] "v"(lane4), [l8] "v"(lane8), [lo] "s"(lo), [hi] "s"(hi), [reps] "s"(reps) : CLOB128, "v7", CLOBS);
  else asm volatile(BODY : : [l4] "v"(lane4), [l8] "v"(lane8), [lo] "s"(lo), [hi] "s"(hi), [reps] "s"(reps) : CLOB256, "v7", CLOBS);
  if (reps < 0) out[threadIdx.x] = 1.f;
}

int main() {
  CHECK(hipSetDevice(0));
  float *out;
  CHECK(hipMalloc(&out, 4096));
  HCHECK(hsa_init());
  HCHECK(hsa_iterate_agents(find_gpu, nullptr));
  HCHECK(hsa_amd_agent_iterate_memory_pools(g_gpu, find_pool, nullptr));
  struct Cfg { int nw; uint32_t G, NS; };
  const Cfg cfgs[] = {{8, 10, 2}};
  const uint64_t entries_per_cu = 2560ull * 8 * 16;   // as codestream.hip: 327 680 entries per CU
  uint32_t *code = nullptr;
  const size_t cap = (size_t)1200 << 20;
  HCHECK(hsa_amd_memory_pool_allocate(g_pool, cap, HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG, (void **)&code));
  HCHECK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, code));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void *)k_run<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CHECK(hipFuncSetAttribute((const void *)k_run<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));   // (one workgroup per CU)
  for (const Cfg &c : cfgs) for (uint32_t mode = 2; mode < 3; mode++) for (uint32_t pairbit : {0u, 1u, 2u, 4u, 8u, 16u}) {
    const uint32_t nregions = 256 * c.nw;
    const uint32_t GDW = 2 * c.G + 1 + ((mode == 1 || mode == 3) ? c.G / 2 : 0);
    const uint32_t SG = (512 - 5) / GDW;
    const uint64_t groups = (entries_per_cu / c.nw / c.G + SG - 1) / SG * SG;
    const uint64_t region_dw = (groups / SG) * (5 + SG * GDW) + 8 + 4096, region_bytes = region_dw * 4;
    if ((size_t)region_bytes * nregions > cap) { printf("config too large\n"); continue; }
    const uint32_t x0 = 12 + (12 & 0), acc0 = x0 + c.G * c.NS, nacc = (c.nw == 16 ? 128 : 256) - acc0;
    const uint64_t ngroups = groups * nregions;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, 0, code, region_dw, groups, nregions, c.G, c.NS, x0, acc0, nacc, SG, mode);
    CHECK(hipDeviceSynchronize());
    printf("--- pair bit %2u, mode %u: %2d waves per CU, groups of %2u entries, %u x-sets, %3u accumulators, %.1f bytes of code per entry\n", pairbit, mode, c.nw, c.G, c.NS, nacc,
           (double)region_bytes / (groups * c.G));
    for (int own = 1; own >= 0; own--) {
      float best = 1e9f;
      for (int r = 0; r < 4; r++) {
        CHECK(hipEventRecord(e0));
        if (c.nw == 16) hipLaunchKernelGGL(k_run<16>, dim3(256), dim3(1024), 65536, 0, code, region_bytes, own, 1, out, pairbit);
        else hipLaunchKernelGGL(k_run<8>, dim3(256), dim3(512), 160 * 1024 - 64, 0, code, region_bytes, own, 1, out, pairbit);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
      }
      printf("%s: %8.3f ms  %6.2f CU cycles per entry (2.4 GHz)\n", own ? "own region per wave (streamed once)" : "all waves in region 0 (cached)      ", best,
             best * 1e-3 * 2.4e9 / ((double)groups * c.G * c.nw));
      fflush(stdout);
    }
  }
  return 0;
}
