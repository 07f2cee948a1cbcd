// valu_issue.hip — issue cost of wave64 VALU instructions on gfx950 at 1, 2, 4, 8 waves per SIMD,
// and of a random-bank ds_read_u16 gather alone and beside VALU work.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
// Every wave runs ITER iterations of a block of 64 instructions over 8 independent registers
// (dependency distance 8).  cycles per wave-instruction per SIMD = wave lifetime (s_memtime)
// / (waves per SIMD x instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 2000;

#define R8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

#define ADD(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b));
#define AND(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(b));
#define ALN(r) asm volatile("v_alignbit_b32 %0, %0, %1, 4" : "+v"(r) : "v"(b));
#define ANDOR(r) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r) : "v"(b));
#define BFE(r) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(r));
#define LSHL64(r) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d##r));
#define SDWA(r) asm volatile("v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r) : "v"(b));

template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(uint32_t *out, unsigned long long *cyc, uint32_t seed, unsigned long long *rt) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
  uint64_t da0 = a0, da1 = a1, da2 = a2, da3 = a3, da4 = a4, da5 = a5, da6 = a6, da7 = a7;
  uint32_t b = seed | 0x10101u;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; i++) {
    if (KIND == 0) { R64(ADD) }
    if (KIND == 1) { R64(AND) }
    if (KIND == 2) { R64(ALN) }
    if (KIND == 3) { R64(ANDOR) }
    if (KIND == 4) { R64(BFE) }
    if (KIND == 5) { R64(LSHL64) }
    if (KIND == 6) { R64(SDWA) }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(da0 ^ da1 ^ da2 ^ da3 ^ da4 ^ da5 ^ da6 ^ da7);
  if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; rt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// A dependent chain of ds_read_u16 at pseudo-random rows of a 32-byte-per-row table (the scan's access pattern),
// CH chains per lane, with NV independent VALU instructions per look-up step beside them.
template <int CH, int NV>
__global__ __launch_bounds__(1024) void lds_kernel(uint32_t *out, unsigned long long *cyc, uint32_t seed, uint32_t rows, unsigned long long *rt) {
  extern __shared__ uint16_t tab[];
  for (uint32_t i = threadIdx.x; i < rows * 16; i += blockDim.x) {
    uint32_t x = (i * 2654435761u + seed) ^ (i >> 3);
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    tab[i] = (uint16_t)((x % rows) << 5);          // entry = next row's byte offset (narrow form)
  }
  __syncthreads();
  uint32_t e[CH], w[CH];
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
  uint32_t b = seed | 0x10101u;
#pragma unroll
  for (int c = 0; c < CH; c++) { e[c] = ((threadIdx.x * 7 + c * 13) % rows) << 5; w[c] = (threadIdx.x * 2654435761u) ^ (c * 40503u) ^ seed; }
  typedef __attribute__((address_space(3))) uint16_t lds_u16;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int s = 0; s < 8; s++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const uint32_t off = (e[c] & 0xFFE0u) | ((w[c] >> (4 * s)) & 0xFu) << 1;
        e[c] = *reinterpret_cast<const lds_u16 *>(static_cast<uintptr_t>(off + 0));
      }
      if (NV >= 1) { ADD(a0) } if (NV >= 2) { ADD(a1) } if (NV >= 3) { ADD(a2) } if (NV >= 4) { ADD(a3) }
      if (NV >= 5) { ADD(a4) } if (NV >= 6) { ADD(a5) } if (NV >= 7) { ADD(a6) } if (NV >= 8) { ADD(a7) }
    }
#pragma unroll
    for (int c = 0; c < CH; c++) w[c] = w[c] * 1664525u + 1013904223u;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
#pragma unroll
  for (int c = 0; c < CH; c++) acc ^= e[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; rt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = __builtin_amdgcn_s_memrealtime() - r0; }
}

static double avg_cycles(unsigned long long *d_cyc, int waves) {
  std::vector<unsigned long long> h(waves);
  CHECK(hipMemcpy(h.data(), d_cyc, waves * 8, hipMemcpyDeviceToHost));
  double s = 0; for (auto x : h) s += (double)x;
  return s / waves;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
  uint32_t *d_out; unsigned long long *d_cyc, *d_rt;
  CHECK(hipMalloc(&d_out, (size_t)cus * 2048 * 4)); CHECK(hipMalloc(&d_cyc, (size_t)cus * 32 * 8)); CHECK(hipMalloc(&d_rt, (size_t)cus * 32 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char *names[] = {"v_add_u32", "v_and_b32", "v_alignbit_b32", "v_and_or_b32", "v_bfe_u32", "v_lshlrev_b64", "v_and_b32_sdwa"};
  printf("== VALU issue: cycles (s_memtime ticks) per wave-instruction per SIMD; ticks may run at a fixed 100 MHz: see ns column ==\n");
  for (int kind = 0; kind < 7; kind++) {
    for (int wps : {1, 2, 4, 8}) {
      const int waves_per_cu = 4 * wps, blocks = cus * waves_per_cu / 4;     // 256-thread blocks = 4 waves, one per SIMD
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0));
        switch (kind) {
          case 0: valu_kernel<0><<<blocks, 256>>>(d_out, d_cyc, 1, d_rt); break;
          case 1: valu_kernel<1><<<blocks, 256>>>(d_out, d_cyc, 1, d_rt); break;
          case 2: valu_kernel<2><<<blocks, 256>>>(d_out, d_cyc, 1, d_rt); break;
          case 3: valu_kernel<3><<<blocks, 256>>>(d_out, d_cyc, 1, d_rt); break;
          case 4: valu_kernel<4><<<blocks, 256>>>(d_out, d_cyc, 1, d_rt); break;
          case 5: valu_kernel<5><<<blocks, 256>>>(d_out, d_cyc, 1, d_rt); break;
          case 6: valu_kernel<6><<<blocks, 256>>>(d_out, d_cyc, 1, d_rt); break;
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double ticks = avg_cycles(d_cyc, blocks * 4), rticks = avg_cycles(d_rt, blocks * 4);
      const double insts = (double)ITER * 64;
      printf("%-16s waves/SIMD %d: wave lifetime %.0f shader ticks = %.1f us (100 MHz counter): clock %.2f GHz; kernel %.3f ms -> %.3f cycles and %.3f ns per wave-instruction per SIMD\n",
             names[kind], wps, ticks, rticks / 100.0, ticks / (rticks * 10.0), ms, ticks / (insts * wps), rticks * 10.0 / (insts * wps));
    }
  }
  printf("== LDS gather (ds_read_u16, pseudo-random rows of 32 B, 1769 rows) beside NV v_add_u32 per look-up; 16 waves per CU ==\n");
#define LDSRUN(CH, NV) do { \
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(lds_kernel<CH, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    float ms = 0; \
    for (int rep = 0; rep < 2; rep++) { \
      CHECK(hipEventRecord(e0)); \
      lds_kernel<CH, NV><<<cus, 1024, 1769 * 32>>>(d_out, d_cyc, 1, 1769, d_rt); \
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); } \
    const double ticks = avg_cycles(d_cyc, cus * 16), rticks = avg_cycles(d_rt, cus * 16); \
    const double lookups = (double)ITER * 8 * CH * 16; /* wave-lookups per CU */ \
    printf("chains %d, extra v_add_u32 per step of CH look-ups %d (plus ~3 address VALU per look-up): kernel %.3f ms, %.2f ns per wave-look-up per CU, %.2f cycles (wave lifetime %.0f cycles, clock %.2f GHz, %.2f ns per wave-look-up by the wave's own clock)\n", CH, NV, ms, ms * 1e6 / lookups, ticks / lookups, ticks, ticks / (rticks * 10.0), rticks * 10.0 / lookups); \
  } while (0)
  LDSRUN(1, 0); LDSRUN(2, 0); LDSRUN(4, 0);
  LDSRUN(2, 2); LDSRUN(2, 4); LDSRUN(2, 6); LDSRUN(2, 8);
  LDSRUN(4, 4); LDSRUN(4, 8);
  return 0;
}
