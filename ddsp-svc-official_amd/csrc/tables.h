#pragma once
#include "common.h"

enum {
    TAB_IRDFT_RE = 1,       // key n0 = n_mag
    TAB_IRDFT_RE_HANN = 2,  // key n0 = n_mag
    TAB_IRDFT_CPLX = 3,     // key n0 = n_mag
    TAB_RDFT_FWD_W = 4,     // key n0 = N
    TAB_RDFT_INV_W = 5,     // key n0 = N
};

// leading dimensions are padded to a multiple of 4 floats so that every row start is 16-byte aligned
static inline int ddsp_pad4(int n) { return (n + 3) & ~3; }
