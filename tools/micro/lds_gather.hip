// lds_gather.hip — what the scan's LDS look-ups cost under other table layouts ("attack the scan's LDS
// floor with a gated experiment"): dependent chains of ds_read_u16, two chains per lane, 16 waves per CU, as scan2 runs them.
//   hipcc --offload-arch=gfx950 -O3 -o lds_gather lds_gather.hip && ./lds_gather
// Next states follow the automaton's shape, not a uniform draw: a step lands on one of HOT shallow rows (trie depth <= 4: 230
// of config 2's 1 769) with probability P_HOT (0.86 for random text against 146 keyword paths), else on a deep row.
// Layouts (cycles of the LDS per wave-look-up decide; a ds_read_b32-class access serves 2 x 32 lanes over 32 banks):
//   0  the shipped one: 32-byte rows (16 entries of two bases), every row once                         [1 769 rows: 56.6 KB]
//   1  three bases per look-up: 128-byte rows (64 entries)                                              [  990 rows: 127 KB]
//   2  the hot rows four times, a lane reads copy (lane & 3): banks 8 c .. 8 c + 7 of a 128-byte line; deep rows once; the
//      address needs a compare and a select more                                                        [230 x 128 + 1 539 x 32 = 78.7 KB]
//   3  every row twice (64-byte lines), copy (lane & 1): no compare                                     [113 KB]
//   4  the 85 rows of depth <= 3 thirty-two times, lane l reads bank l % 32 of its copy: no two lanes of a group meet there;
//      the rest once                                                                                    [85 x 1 KB + 1 684 x 32 = 139 KB]
//   5  the split look-up: a shallow state (trie depth <= 4) is a function of the last bases and needs no
//      row; a "goes deep" bitmap over the 6-mer, 32 copies with lane l at bank l % 32 (ds_read_b32, no two lanes of a group meet),
//      is probed on every step, and only the lanes that are deep (14 %) or go deep (12 %) read a 16-bit entry — the deep rows'
//      table or a 6-mer -> state table, one masked ds_read_u16 with the address picked per lane                [56.6 + 16 + 8 KB]
// Prints cycles (s_memtime) and ns per wave-look-up per CU, and what a 10 M x 150-nt step would need of the LDS at that cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 2000;
constexpr uint32_t ROWS = 1769, HOT = 230, HOT3 = 85;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16; return x; }

template <int LAYOUT>
__global__ __launch_bounds__(1024) void gather_kernel(uint32_t *out, unsigned long long *cyc, unsigned long long *rt, uint32_t seed, uint32_t p_hot_1024) {
  extern __shared__ uint16_t tab[];
  constexpr uint32_t NSYM = LAYOUT == 1 ? 64u : 16u;
  constexpr uint32_t NROWS = LAYOUT == 1 ? 990u : ROWS;
  constexpr uint32_t NHOT = LAYOUT == 4 ? HOT3 : HOT;
  constexpr uint32_t SHALLOW = 0xFFFFu, BITMAP_AT = ROWS * 16u, SIX_AT = BITMAP_AT + 8192u;      // (uint16 units: the bitmap's 4 096 dwords, then the 6-mer table)
  // a row's first cell (32-byte units) in the layout
  auto cell = [](uint32_t row) -> uint32_t {
    if (LAYOUT == 1) return row * 4u;
    if (LAYOUT == 2) return row < HOT ? row * 4u : HOT * 4u + (row - HOT);
    if (LAYOUT == 3) return row * 2u;
    if (LAYOUT == 4) return row < HOT3 ? row * 32u : HOT3 * 32u + (row - HOT3);
    return row;
  };
  // entries: the next row's first cell (the hot rows take the share p_hot of the steps)
  for (uint32_t i = threadIdx.x; i < NROWS * NSYM; i += blockDim.x) {
    const uint32_t row = i / NSYM, sym = i % NSYM;
    const uint32_t x = mix(i * 2654435761u + seed);
    const uint32_t nxt = (x & 1023u) < p_hot_1024 ? (x >> 10) % NHOT : NHOT + (x >> 10) % (NROWS - NHOT);
    uint16_t ent = (uint16_t)cell(nxt);
    if (LAYOUT == 5) ent = (x & 1023u) < 154u ? (uint16_t)(NHOT + (x >> 10) % (NROWS - NHOT)) : (uint16_t)SHALLOW;      // a deep state stays deep on 15 % of its steps
    if (LAYOUT == 2 && row < HOT) { for (uint32_t c = 0; c < 4; c++) tab[(cell(row) + c) * 16u + sym] = ent; }
    else if (LAYOUT == 3) { tab[cell(row) * 16u + sym] = ent; tab[(cell(row) + 1u) * 16u + sym] = ent; }
    else if (LAYOUT == 4 && row < HOT3) { for (uint32_t c = 0; c < 32; c++) tab[(cell(row) * 16u) + (sym >> 1) * 64u + c * 2u + (sym & 1u)] = ent; }   // dword (sym / 2) of copy c at bank c
    else if (LAYOUT == 1) tab[cell(row) * 16u + sym] = ent;
    else tab[cell(row) * 16u + sym] = ent;
  }
  if (LAYOUT == 5) {
    uint32_t *bm = reinterpret_cast<uint32_t *>(tab + BITMAP_AT);
    for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) {      // dword (i / 32) of copy (i % 32): 13.8 % of the 6-mers go deep
      uint32_t v = 0;
      for (uint32_t b = 0; b < 32; b++) v |= (mix((i >> 5) * 32u + b + seed * 7919u) % 1000u < 138u ? 1u : 0u) << b;
      bm[i] = v;
    }
    for (uint32_t i = threadIdx.x; i < 4096u; i += blockDim.x) tab[SIX_AT + i] = (uint16_t)(NHOT + mix(i ^ seed) % (NROWS - NHOT));
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t e[2], w[2];
  for (int c = 0; c < 2; c++) { e[c] = cell((threadIdx.x * 7u + c * 13u) % NHOT); w[c] = mix(threadIdx.x * 2654435761u ^ (c * 40503u) ^ seed); }
  if (LAYOUT == 5) e[0] = e[1] = SHALLOW;
  typedef __attribute__((address_space(3))) uint16_t lds_u16;
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
      for (int c = 0; c < 2; c++) {
        uint32_t off;
        if (LAYOUT == 5) {
          const uint32_t k6 = (w[c] >> (4 * s)) & 0xFFFu;
          const uint32_t probe = *reinterpret_cast<const lds_u32 *>(static_cast<uintptr_t>(2u * BITMAP_AT + (((k6 >> 5) << 5) + (lane & 31u)) * 4u));
          const bool deep = e[c] != SHALLOW, go = ((probe >> (k6 & 31u)) & 1u) != 0u;
          const uint32_t at = deep ? ((e[c] << 5) | ((k6 & 0xFu) << 1)) : 2u * (SIX_AT + k6);
          uint32_t nx = SHALLOW;
          if (deep || go) nx = *reinterpret_cast<const lds_u16 *>(static_cast<uintptr_t>(at));
          e[c] = nx;
          continue;
        }
        if (LAYOUT == 0) off = (e[c] << 5) | (((w[c] >> (4 * s)) & 0xFu) << 1);
        else if (LAYOUT == 1) off = (e[c] << 5) | (((w[c] >> (6 * s)) & 0x3Fu) << 1);
        else if (LAYOUT == 2) off = (e[c] << 5) | (((w[c] >> (4 * s)) & 0xFu) << 1) | (e[c] < HOT * 4u ? (lane & 3u) << 5 : 0u);
        else if (LAYOUT == 3) off = (e[c] << 5) | (((w[c] >> (4 * s)) & 0xFu) << 1) | ((lane & 1u) << 5);
        else { const uint32_t sym = (w[c] >> (4 * s)) & 0xFu; off = e[c] < HOT3 * 32u ? (e[c] << 5) + (sym >> 1) * 128u + (lane & 31u) * 4u + (sym & 1u) * 2u : (e[c] << 5) | (sym << 1); }
        e[c] = *reinterpret_cast<const lds_u16 *>(static_cast<uintptr_t>(off));
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) w[c] = w[c] * 1664525u + 1013904223u;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = e[0] ^ e[1];
  if (lane == 0) { cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; rt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int LAYOUT>
static int run(const char *what, uint32_t lds_bytes, int cus, uint32_t *d_out, unsigned long long *d_cyc, unsigned long long *d_rt, uint32_t p_hot, double lookups_per_read) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gather_kernel<LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0));
    gather_kernel<LAYOUT><<<cus, 1024, lds_bytes>>>(d_out, d_cyc, d_rt, 1u + rep, p_hot);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  CHECK(hipGetLastError());
  std::vector<unsigned long long> hc(cus * 16), hr(cus * 16);
  CHECK(hipMemcpy(hc.data(), d_cyc, hc.size() * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hr.data(), d_rt, hr.size() * 8, hipMemcpyDeviceToHost));
  double ticks = 0, rticks = 0; for (size_t i = 0; i < hc.size(); i++) { ticks += hc[i]; rticks += hr[i]; }
  ticks /= hc.size(); rticks /= hr.size();
  const double lookups = (double)ITER * 4 * 2 * 16;      // wave-look-ups per CU
  const double ns = rticks * 10.0 / lookups;
  // a 10 M-read step: reads x look-ups per read / 64 lanes / CUs wave-look-ups per CU
  const double step_us = 1e7 * lookups_per_read / 64.0 / cus * ns / 1e3;
  printf("%-78s %5.1f KB  %.2f cycles  %.2f ns per wave-look-up per CU  -> %.0f us of LDS time per 10 M-read step (%.0f look-ups per read)\n", what, lds_bytes / 1024.0,
         ticks / lookups, ns, step_us, lookups_per_read);
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs; dependent ds_read_u16 chains, 2 per lane, 16 waves per CU\n", p.name, cus);
  uint32_t *d_out; unsigned long long *d_cyc, *d_rt;
  CHECK(hipMalloc(&d_out, (size_t)cus * 1024 * 4)); CHECK(hipMalloc(&d_cyc, (size_t)cus * 16 * 8)); CHECK(hipMalloc(&d_rt, (size_t)cus * 16 * 8));
  for (uint32_t p_hot : {880u, 584u}) {      // 0.86: depth <= 4 (layouts 0-3); 0.57: depth <= 3 (layout 4's private copies)
    printf("-- share of the steps that land on a hot row: %.2f\n", p_hot / 1024.0);
    if (run<0>("0 shipped: 32-byte rows, once", ROWS * 32, cus, d_out, d_cyc, d_rt, p_hot, 75)) return 1;
    if (run<1>("1 three bases per look-up: 128-byte rows (halves-only automaton, 990 rows)", 990 * 128, cus, d_out, d_cyc, d_rt, p_hot, 50)) return 1;
    if (run<2>("2 hot rows x 4 (copy = lane & 3: 8 banks each), + compare and select", HOT * 128 + (ROWS - HOT) * 32, cus, d_out, d_cyc, d_rt, p_hot, 75)) return 1;
    if (run<3>("3 every row x 2 (copy = lane & 1)", ROWS * 64, cus, d_out, d_cyc, d_rt, p_hot, 75)) return 1;
    if (run<4>("4 rows of depth <= 3 x 32 (a bank per lane), the rest once", HOT3 * 1024 + (ROWS - HOT3) * 32, cus, d_out, d_cyc, d_rt, p_hot, 75)) return 1;
  }
  printf("-- the split look-up (14 %% of the lanes deep, 12 %% going deep: its own draw)\n");
  if (run<5>("5 split: a bitmap probe per step (a bank per lane), a masked 16-bit read for the deep lanes", ROWS * 32 + 16384 + 8192, cus, d_out, d_cyc, d_rt, 880u, 75)) return 1;
  return 0;
}
