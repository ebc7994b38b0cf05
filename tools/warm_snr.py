"""BASELINE config 5 (8k QPSK 7/8 + AWGN, ofdm_sym_acquisition's snr = the channel's) at 9 / 8 / 7 dB with the default warm-up of the chunk decoders and with 288 windows
(dvbt_rx_params.viterbi_warm_windows): bytes of the Viterbi tap that differ from the oracle's, next to the bits of the decoder's INPUT that differ (hard decisions within
float rounding of a boundary) -- which of the two explains the differing bytes of profiles/r03_snr_sweep.jsonl where the RS decoder has given up.  GPU box: python tools/warm_snr.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gr_dvbt_amd as g
from oracle import pyoracle as po

nsf = int(sys.argv[1]) if len(sys.argv) > 1 else 3
c = po.cfg(po.QPSK, po.C7_8, po.T8k)
clean = po.stream_slice(c, nsf, 21)
for snr in (9.0, 8.0, 7.0):
    iq = po.channel(clean, c.N, snr_db=snr, seed=5)
    o = po.rx(c, iq, snr_db=snr, want=("bitdeint", "vit"))
    row = {"snr_db": snr, "superframes": nsf, "oracle_viterbi_bytes": int(len(o["vit"])), "lock_periods": len(o["lock_periods"])}
    for warm in (0, 288):
        rx = g.Rx(po.QPSK, po.C7_8, po.T8k, max_samples=len(iq), snr_db=snr, taps=True, viterbi_warm_windows=warm)
        rx.run(iq)
        v, bd = rx.tap(g.TAP_VITERBI), rx.tap(g.TAP_BITDEINT)
        same_len = len(v) == len(o["vit"])
        nb = int(np.unpackbits(bd.reshape(-1) ^ o["bitdeint"].reshape(-1)).sum()) if bd.size == o["bitdeint"].size else None
        row["warm_up_%d" % (warm or 72)] = {"viterbi_bytes_that_differ": int((v != o["vit"]).sum()) if same_len else None, "same_length": bool(same_len),
                                            "decoder_input_bits_that_differ": nb}
        rx.close()
    print(json.dumps(row), flush=True)
