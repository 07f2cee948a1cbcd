/*
 * dcrx_codes.h — counter indices and per-read status codes shared by the C-ABI
 * (include/dcrx.h), the HIP kernels and the CPU oracle.
 *
 * Every counter is one key of the reference's `counts = coll.Counter()`
 * (reference src/decombinator/decombine.py:598); the line where the reference
 * increments it is cited next to each index.  Every status code is one exit
 * path of the reference's dcr() (decombine.py:534-585) for the LAST attempted
 * frame of a read.
 */
#ifndef DCRX_CODES_H
#define DCRX_CODES_H

#define DCRX_N_COUNTERS 32

enum dcrx_counter {
  DCRX_C_MULTIPLE_V_MATCHES = 0,        /* decombine.py:279 */
  DCRX_C_VERR2 = 1,                     /* :318  (half1 found, error in 2nd half) */
  DCRX_C_FOUNDV1NOTV2 = 2,              /* :334 */
  DCRX_C_VERR1 = 3,                     /* :370 */
  DCRX_C_FOUNDV2NOTV1 = 4,              /* :389 and, by the reference's own slip, :526 */
  DCRX_C_NO_VTAGS_FOUND = 5,            /* :393 */
  DCRX_C_MULTIPLE_J_MATCHES = 6,        /* :403 */
  DCRX_C_JERR2 = 7,                     /* :445 */
  DCRX_C_FOUNDJ1NOTJ2 = 8,              /* :469 */
  DCRX_C_JERR1 = 9,                     /* :504 */
  DCRX_C_NO_J_ASSIGNED = 10,            /* :530 */
  DCRX_C_DCRFILTER_INTERTAGN = 11,      /* :556 */
  DCRX_C_DCRFILTER_TOOLONG_INTERTAG = 12, /* :560 */
  DCRX_C_DCRFILTER_IMPOSS_DELETION = 13,  /* :565 */
  DCRX_C_DCRFILTER_TAG_OVERLAP = 14,    /* :569 */
  DCRX_C_VJ_ASSIGNMENT_FAILED = 15,     /* :584 */
  DCRX_C_V_DEL_FAILED_TAG_AT_END = 16,  /* :761 */
  DCRX_C_V_DEL_FAILED = 17,             /* :784 */
  DCRX_C_J_DEL_FAILED = 18,             /* :816 */
  DCRX_C_VJ_COUNT = 19,                 /* :1013 */
  DCRX_C_READ_COUNT = 20,               /* :991 */
  DCRX_C_FOUNDJ2NOTJ1 = 21,             /* never incremented by the reference (:526 bumps foundv2notv1); stays 0 */
  DCRX_C_FRAME_FORWARD = 22,            /* not a reference counter: reads decombined in the forward frame */
  DCRX_C_DEVICE_ERRORS = 31             /* not a reference counter: waves that gave up waiting for another (the fused scan's ring:
                                           a bound no run has reached); non-zero = the call's records are NOT complete */
};

enum dcrx_status {
  DCRX_S_OK = 0,
  DCRX_S_V_MULTI = 1,            /* :278-280 */
  DCRX_S_V_WALK_FAIL_AT_END = 2, /* full-tag V hit, walk refused at :760-762 */
  DCRX_S_V_WALK_FAIL = 3,        /* full-tag V hit, walk exhausted :783-785 */
  DCRX_S_V_HALF1_EXHAUSTED = 4,  /* :334-335 */
  DCRX_S_V_HALF2_EXHAUSTED = 5,  /* :389-390 */
  DCRX_S_V_NONE = 6,             /* :393-394 */
  DCRX_S_J_MULTI = 7,            /* :402-404 */
  DCRX_S_J_WALK_FAIL = 8,        /* full-tag J hit, walk exhausted :815-817 */
  DCRX_S_J_HALF1_EXHAUSTED = 9,  /* :469-470 */
  DCRX_S_J_HALF2_EXHAUSTED = 10, /* :526-527 */
  DCRX_S_J_NONE = 11,            /* :530-531 */
  DCRX_S_F_INTERTAG_N = 12,      /* :553-556 */
  DCRX_S_F_TOOLONG = 13,         /* :557-560 */
  DCRX_S_F_IMPOSS_DEL = 14,      /* :561-565 */
  DCRX_S_F_OVERLAP = 15,         /* :566-569 */
  DCRX_N_STATUS = 16
};

enum dcrx_orientation {
  DCRX_ORIENT_REVERSE = 0, /* decombine.py:999-1001 (default) */
  DCRX_ORIENT_FORWARD = 1, /* :1002-1004 */
  DCRX_ORIENT_BOTH = 2     /* :1005-1010 */
};

#endif /* DCRX_CODES_H */
