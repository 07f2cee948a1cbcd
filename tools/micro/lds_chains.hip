// lds_chains.hip — is the scan's 6.7 cycles per wave-look-up the LDS arrays' throughput (random rows: ~3.3 lanes on the busiest
// bank, twice per 64-lane access) or the latency of two dependent chains per lane on sixteen waves?  The shipped layout
// (32-byte rows, 1 769 of them, next states drawn as the automaton draws them) with 1, 2, 3, 4, 6 and 8 chains per lane.
//   hipcc --offload-arch=gfx950 -O3 -o lds_chains lds_chains.hip && ./lds_chains
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 1000;
constexpr uint32_t ROWS = 1769, HOT = 230;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16; return x; }

template <int CH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void chains_kernel(uint32_t *out, unsigned long long *rt, uint32_t seed) {
  extern __shared__ uint16_t tab[];
  for (uint32_t i = threadIdx.x; i < ROWS * 16u; i += blockDim.x) {
    const uint32_t x = mix(i * 2654435761u + seed);
    const uint32_t nxt = (x & 1023u) < 880u ? (x >> 10) % HOT : HOT + (x >> 10) % (ROWS - HOT);
    tab[i] = (uint16_t)nxt;
  }
  __syncthreads();
  uint32_t e[CH], w[CH];
  for (int c = 0; c < CH; c++) { e[c] = (threadIdx.x * 7u + c * 13u) % HOT; w[c] = mix(threadIdx.x * 2654435761u ^ (c * 40503u) ^ seed); }
  typedef __attribute__((address_space(3))) uint16_t lds_u16;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < ITER; i++) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        const uint32_t off = (e[c] << 5) | (((w[c] >> (4 * s)) & 0xFu) << 1);
        e[c] = *reinterpret_cast<const lds_u16 *>(static_cast<uintptr_t>(off));
      }
    }
#pragma unroll
    for (int c = 0; c < CH; c++) w[c] = w[c] * 1664525u + 1013904223u;
  }
  uint32_t acc = 0;
  for (int c = 0; c < CH; c++) acc ^= e[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) rt[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = __builtin_amdgcn_s_memrealtime() - r0;
}

template <int CH, int WAVES>
static int run(int cus, uint32_t *d_out, unsigned long long *d_rt) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chains_kernel<CH, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int rep = 0; rep < 3; rep++) chains_kernel<CH, WAVES><<<cus, 64 * WAVES, ROWS * 32>>>(d_out, d_rt, 1u + rep);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> hr(cus * WAVES);
  CHECK(hipMemcpy(hr.data(), d_rt, hr.size() * 8, hipMemcpyDeviceToHost));
  double ticks = 0; for (auto v : hr) ticks += v; ticks /= hr.size();
  const double lookups = (double)ITER * 4 * CH * WAVES;      // wave-look-ups per CU
  const double ns = ticks * 10.0 / lookups;
  printf("%d chains per lane, %2d waves per CU: %.2f ns per wave-look-up per CU (%.2f cycles at 2.4 GHz) -> %3.0f us of a 10 M-read step's 75 look-ups per read\n",
         CH, WAVES, ns, ns * 2.4, 1e7 * 75 / 64.0 / cus * ns / 1e3);
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  uint32_t *d_out; unsigned long long *d_rt;
  CHECK(hipMalloc(&d_out, (size_t)cus * 1024 * 4)); CHECK(hipMalloc(&d_rt, (size_t)cus * 16 * 8));
  if (run<1, 16>(cus, d_out, d_rt) || run<2, 16>(cus, d_out, d_rt) || run<3, 16>(cus, d_out, d_rt) || run<4, 16>(cus, d_out, d_rt) ||
      run<6, 16>(cus, d_out, d_rt) || run<8, 16>(cus, d_out, d_rt)) return 1;
  if (run<2, 12>(cus, d_out, d_rt) || run<2, 8>(cus, d_out, d_rt) || run<4, 8>(cus, d_out, d_rt) || run<4, 12>(cus, d_out, d_rt)) return 1;
  return 0;
}
