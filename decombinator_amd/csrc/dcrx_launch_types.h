// dcrx_launch_types.h — plain structs shared by host launch code, kernels and
// the test-only host emulation (no HIP headers needed).
#pragma once

#include <cstdint>

#include "dcrx_debug_flags.h"

// words of one read kept in registers by the fast scan: reads up to 16*20 = 320 nt
#define DCRX_NWMAX 20
// words of the longest reads the v2 kernels hold in registers (128-byte stride: 511 nt), one read per lane
#define DCRX_V2_NWLONG 32
#define DCRX_FAST_READ_LEN (16 * DCRX_NWMAX)
// reads of up to 511 nt (strides of up to 128 bytes) run on the register shapes / the list kernel, whose hit positions have
// nine bits; longer ones — to 65 535 nt: the 16-bit lengths, exception positions and record offsets of the ABI — in batches of
// their own through decombine_long_kernel, which keeps every position in a plain integer
#define DCRX_FAST_MAX_STRIDE 128
#define DCRX_FAST_MAX_READ_LEN 511
#define DCRX_MAX_STRIDE 16384
#define DCRX_MAX_READ_LEN 65535
#define DCRX_BLOCK 512   /* fast kernel, one base per step */
#define DCRX_BLOCK16 1024 /* fast kernel, two bases per step (one block per CU: the table takes most of the LDS) */
#define DCRX_QBLOCK 512  /* list kernel */
#define DCRX_QUEUE_HEADER 8 /* work counters in front of the queues: rescue count, general count, -, ticket, blocks done */
#define DCRX_EXC_LDS 4    /* exception entries of a read the list kernel keeps in LDS */
/* extra dwords of a general-kernel lane slot after the hit lists: exception positions (u16) + bytes (u8) */
#define DCRX_GSLOT_EXTRA (DCRX_EXC_LDS / 2 + DCRX_EXC_LDS / 4)

namespace dcrx {

struct BatchDev {
  const uint8_t *packed;
  uint32_t stride;
  uint32_t read_len;
  const uint16_t *lens;
  uint64_t n_reads;
  uint64_t n_exc;
  const uint32_t *exc_read;
  const uint16_t *exc_pos;
  const uint8_t *exc_chr;
  const uint32_t *exc_flag;  // workspace bitmap: read owns >= 1 exception
};

struct CfgDev {
  int32_t orientation, allow_ns, lenthreshold;
  uint32_t flags;
};

}  // namespace dcrx
