// dcrx_synth.hip — device side of the synthetic-read generator
// (include/dcrx_synth.h): one thread per read, same function as the host.
#include <hip/hip_runtime.h>

#include "dcrx_synth_core.h"

namespace dcrx {

__global__ __launch_bounds__(256) void synth_kernel(GeneDevPtrs V, GeneDevPtrs J, SynthParams P, uint64_t first,
                                                    uint64_t n, uint32_t stride, uint8_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) synth_read(V, J, P, first + i, reinterpret_cast<uint32_t *>(out + i * (uint64_t)stride), stride / 4);
}

hipError_t launch_synth(const DevTables &T, const SynthParams &P, uint64_t first, uint64_t n, uint32_t stride,
                        uint8_t *d_packed, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const uint64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, T.g[0], T.g[1], P, first, n, stride,
                     d_packed);
  return hipGetLastError();
}

}  // namespace dcrx
