"""gr_dvbt_amd -- host-side Python binding of libdvbt_hip.so (the MI355X DVB-T RX hot path).

Thin ctypes layer over the C ABI declared in include/dvbt_hip.h.  There is no CPU
fallback: importing works anywhere, but every compute entry point raises unless the HIP
library is built in-tree (gr_dvbt_amd/lib/libdvbt_hip.so) and a GPU is visible.
"""
from .binding import (  # noqa: F401
    lib, build, DvbtError, RxParams, RxReport, Rx, RxStream, get_dims, device_count, Block, Tag, Sideband,
    TAG_SYNC_START, TAG_SUPERFRAME_START, TAG_SYMBOL_INDEX,
    QPSK, QAM16, QAM64, NH, C1_2, C2_3, C3_4, C5_6, C7_8, T2k, T8k, G1_32, G1_16, G1_8, G1_4,
    TAP_ACQ, TAP_FFT, TAP_EQ, TAP_DEMAP, TAP_SYMDEINT, TAP_BITDEINT, TAP_VITERBI, TAP_DEINT,
    TAP_RS, TAP_TS, TAP_CP_START, TAP_SYMBOL_INDEX, TAP_FREQ_OFFSET, TAP_BITDEINT_LP, TAP_SOFT, TAP_CSI, TAP_BITDEINT_LOG, ALPHA1, ALPHA2, ALPHA4, AUTO,
)
