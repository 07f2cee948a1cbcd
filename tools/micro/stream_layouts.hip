// stream_layouts.hip — what does the scan kernel's way of fetching its reads cost by itself?  (round 6, verdict item 1 (a))
// Persistent 1 024-thread blocks, one per CU, each owning a contiguous range of 10 M reads x 40 B; a wave takes items of 128 reads
// (two per lane) in turn, the next item's loads in flight while this one is "consumed" (an XOR fold), exactly the shape of
// scan2_kernel's v2_load_item — and the same bytes fetched in other ways:
//   0  shipped: per lane and read five 8-byte loads at a 40-byte lane stride (array of structs)
//   1  array of structs, per read 16 + 16 + 8 bytes (8-byte aligned dwordx4)
//   2  tiles of 64 reads, 16-byte chunks chunk-major (two dwordx4 + one dwordx2 per read, every wave instruction whole lines)
//   3  tiles of 64 reads, word-major (ten dword loads per read, 256 contiguous bytes per wave instruction)
//   4  the item's 5 120 bytes as five lane-contiguous dwordx4 (NOT per-read data in a lane: the ceiling of this loop shape)
//   5..7  as 0 / 2 / 4 with two items of prefetch
// and a plain grid-stride dwordx4 stream (no persistence) as the chip's reference.
//   hipcc --offload-arch=gfx950 -O3 -o stream_layouts stream_layouts.hip && ./stream_layouts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr uint64_t N_READS = 10000000ull;
constexpr int STRIDE = 40, NW = 10, RPL = 2;

struct Item { uint32_t w[RPL][NW]; };

template <int MODE>
__device__ __forceinline__ void load_item(const uint8_t *__restrict__ base, const uint64_t first, const uint64_t hi, const int lane, Item &it) {
  if (MODE == 4) {      // lane-contiguous: 20 dwords per lane = five dwordx4 of the item's 5 120 bytes
    const uint4 *p = reinterpret_cast<const uint4 *>(base + first * STRIDE);
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const uint64_t idx = (uint64_t)k * 64 + lane;
      uint4 t = make_uint4(0, 0, 0, 0);
      if (first + (idx * 16) / STRIDE < hi) t = p[idx];
      uint32_t *d = &it.w[0][0] + 4 * k;
      d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < RPL; q++) {
    const uint64_t r = first + (uint64_t)q * 64 + lane;
    const bool live = r < hi;
    if (MODE == 0) {
      const uint2 *p = reinterpret_cast<const uint2 *>(base + (live ? r : 0) * STRIDE);
#pragma unroll
      for (int k = 0; k < NW / 2; k++) { uint2 t = make_uint2(0, 0); if (live) t = p[k]; it.w[q][2 * k] = t.x; it.w[q][2 * k + 1] = t.y; }
    } else if (MODE == 1) {
      const uint8_t *p = base + (live ? r : 0) * STRIDE;
      uint4 a = make_uint4(0, 0, 0, 0), b = a; uint2 c = make_uint2(0, 0);
      if (live) {
        // (8-byte aligned 16-byte loads: the hardware takes them; the compiler must not assume 16)
        typedef uint4 __attribute__((aligned(8))) uint4_a8;
        a = *reinterpret_cast<const uint4_a8 *>(p); b = *reinterpret_cast<const uint4_a8 *>(p + 16); c = *reinterpret_cast<const uint2 *>(p + 32);
      }
      it.w[q][0] = a.x; it.w[q][1] = a.y; it.w[q][2] = a.z; it.w[q][3] = a.w; it.w[q][4] = b.x; it.w[q][5] = b.y; it.w[q][6] = b.z; it.w[q][7] = b.w;
      it.w[q][8] = c.x; it.w[q][9] = c.y;
    } else if (MODE == 2) {      // tile of 64 reads = 2 560 bytes: [chunk 0: 64 x 16][chunk 1: 64 x 16][chunk 2: 64 x 8]
      const uint8_t *t = base + (first + (uint64_t)q * 64) * STRIDE;
      uint4 a = make_uint4(0, 0, 0, 0), b = a; uint2 c = make_uint2(0, 0);
      if (live) { a = reinterpret_cast<const uint4 *>(t)[lane]; b = reinterpret_cast<const uint4 *>(t + 1024)[lane]; c = reinterpret_cast<const uint2 *>(t + 2048)[lane]; }
      it.w[q][0] = a.x; it.w[q][1] = a.y; it.w[q][2] = a.z; it.w[q][3] = a.w; it.w[q][4] = b.x; it.w[q][5] = b.y; it.w[q][6] = b.z; it.w[q][7] = b.w;
      it.w[q][8] = c.x; it.w[q][9] = c.y;
    } else {                     // MODE 3: word-major tile
      const uint32_t *t = reinterpret_cast<const uint32_t *>(base + (first + (uint64_t)q * 64) * STRIDE);
#pragma unroll
      for (int k = 0; k < NW; k++) { uint32_t v = 0; if (live) v = t[k * 64 + lane]; it.w[q][k] = v; }
    }
  }
}

__device__ __forceinline__ uint32_t fold(const Item &it) {
  uint32_t a = 0;
#pragma unroll
  for (int q = 0; q < RPL; q++)
#pragma unroll
    for (int k = 0; k < NW; k++) a ^= it.w[q][k] + (uint32_t)k;
  return a;
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(const uint8_t *__restrict__ base, uint32_t *__restrict__ out, uint64_t per_block, uint64_t n) {
  __shared__ uint32_t next;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) next = 0;
  __syncthreads();
  const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
  const uint32_t n_items = lo < hi ? (uint32_t)((hi - lo + 127) / 128) : 0u;
  auto draw = [&]() -> uint32_t { uint32_t i = 0; if (lane == 0) i = atomicAdd(&next, 1u); return (uint32_t)__builtin_amdgcn_readfirstlane((int)i); };
  uint32_t acc = 0;
  Item cur, nx, nx2;
  uint32_t i0 = draw(), i1 = 0xFFFFFFFFu, i2 = 0xFFFFFFFFu;
  if (i0 < n_items) load_item<MODE>(base, lo + (uint64_t)i0 * 128, hi, lane, cur);
  if (DEPTH >= 2) { i1 = draw(); if (i1 < n_items) load_item<MODE>(base, lo + (uint64_t)i1 * 128, hi, lane, nx); }
  while (i0 < n_items) {
    if (DEPTH == 1) {
      i1 = draw();
      if (i1 < n_items) load_item<MODE>(base, lo + (uint64_t)i1 * 128, hi, lane, nx);
      acc ^= fold(cur);
      cur = nx; i0 = i1;
    } else {
      i2 = draw();
      if (i2 < n_items) load_item<MODE>(base, lo + (uint64_t)i2 * 128, hi, lane, nx2);
      acc ^= fold(cur);
      cur = nx; nx = nx2; i0 = i1; i1 = i2;
    }
  }
  out[(size_t)blockIdx.x * 1024 + tid] = acc;
}

__global__ __launch_bounds__(256) void plain_kernel(const uint4 *__restrict__ p, uint32_t *__restrict__ out, uint64_t n16) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) { const uint4 t = p[i]; acc ^= t.x ^ t.y ^ t.z ^ t.w; }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE, int DEPTH>
static int run(const char *what, int cus, const uint8_t *d, uint32_t *d_out) {
  const uint64_t per_block = (((N_READS + cus - 1) / cus + 511) / 512) * 512;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int rep = 0; rep < 12; rep++) {
    CHECK(hipEventRecord(a));
    stream_kernel<MODE, DEPTH><<<cus, 1024>>>(d, d_out, per_block, N_READS);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float t; CHECK(hipEventElapsedTime(&t, a, b)); if (rep >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2];
  printf("%-72s median %.1f us (min %.1f) -> %.2f TB/s\n", what, med * 1e3, ms[0] * 1e3, N_READS * 40.0 / (med * 1e-3) / 1e12);
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  uint8_t *d; uint32_t *d_out;
  const size_t bytes = N_READS * STRIDE + 65536;
  CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&d_out, (size_t)4096 * 1024 * 4));
  CHECK(hipMemset(d, 0x5A, bytes));
  for (int warm = 0; warm < 200; warm++) plain_kernel<<<cus * 8, 256>>>(reinterpret_cast<const uint4 *>(d), d_out, N_READS * 40 / 16);      // the clocks up
  CHECK(hipDeviceSynchronize());
  {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int g : {cus * 4, cus * 8, cus * 16}) {
      std::vector<float> ms;
      for (int rep = 0; rep < 12; rep++) {
        CHECK(hipEventRecord(a)); plain_kernel<<<g, 256>>>(reinterpret_cast<const uint4 *>(d), d_out, N_READS * 40 / 16); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float t; CHECK(hipEventElapsedTime(&t, a, b)); if (rep >= 2) ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      printf("plain grid-stride dwordx4 stream, %5d blocks of 256: median %.1f us -> %.2f TB/s\n", g, ms[ms.size() / 2] * 1e3, N_READS * 40.0 / (ms[ms.size() / 2] * 1e-3) / 1e12);
    }
  }
  if (run<0, 1>("0 shipped: 5 x 8 B per read at a 40-B lane stride, one item ahead", cus, d, d_out)) return 1;
  if (run<1, 1>("1 array of structs, 16 + 16 + 8 B per read, one item ahead", cus, d, d_out)) return 1;
  if (run<2, 1>("2 tiles of 64 reads, 16-B chunks chunk-major, one item ahead", cus, d, d_out)) return 1;
  if (run<3, 1>("3 tiles of 64 reads, word-major (10 dword loads), one item ahead", cus, d, d_out)) return 1;
  if (run<4, 1>("4 lane-contiguous dwordx4 (ceiling of the loop shape), one item ahead", cus, d, d_out)) return 1;
  if (run<0, 2>("5 shipped loads, two items ahead", cus, d, d_out)) return 1;
  if (run<2, 2>("6 chunk-major tiles, two items ahead", cus, d, d_out)) return 1;
  if (run<4, 2>("7 lane-contiguous dwordx4, two items ahead", cus, d, d_out)) return 1;
  return 0;
}
