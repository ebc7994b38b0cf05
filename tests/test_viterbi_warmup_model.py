"""How early a chunk decoder of the Viterbi stage must start (dvbt_rx_params.viterbi_warm_windows), on the CPU with the oracle's decoder
(tools/hier_warmup.py holds the full series): a decoder started at a block boundary from all-zero metrics against the streaming decoder from W windows behind its start.
The statements the default of 72 windows rests on, at a size that runs in seconds:
  * at a pre-Viterbi bit error rate of 2 % (rate 7/8: the worst puncturing) no start of this sample differs at 72 windows (tools/warm_proof_series.py: two of 83,000 do, none at 1 %);
  * on a collapsed channel (8 %) some starts DO differ at 72 windows -- the equality is a statement about the input -- and none at 288;
  * on the HP stream of a hierarchical transmission (two thirds of the decoder's input bits are constant zeros) some starts differ at 72 windows on a CLEAN signal, none at 288."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _decoder_input(po, hier, nsf=6):
    c = po.cfg(po.QAM64, po.C7_8, po.T2k, hierarchy=hier)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.tx(c, po.make_ts((272 * ibits * nsf) // (204 * 8), 5), lead_in=500, tail=3 * c.N)
    r = po.rx(c, iq, want=("bitdeint",))
    return c, r["bitdeint"].reshape(-1).copy()


def _with_bit_errors(vin, m, ber, seed):
    rng = np.random.RandomState(seed)
    flips = np.zeros(len(vin), np.uint8)
    for b in range(m):
        flips |= (rng.rand(len(vin)) < ber).astype(np.uint8) << b
    return vin ^ flips


def test_warm_up_of_the_chunk_decoders(po):
    import hier_warmup as hw
    c, vin = _decoder_input(po, 0)
    ok = hw.experiment(c, _with_bit_errors(vin, c.m, 0.02, 3), (72, 288), 700, 7)["by_warm_up_windows"]
    assert ok["72"]["starts_that_differ"] == 0 and ok["288"]["starts_that_differ"] == 0
    bad = hw.experiment(c, _with_bit_errors(vin, c.m, 0.08, 3), (72, 288), 700, 7)["by_warm_up_windows"]
    assert 0 < bad["72"]["starts_that_differ"] < 60 and bad["72"]["last_differing_byte_behind_the_start"] < 200
    assert bad["288"]["starts_that_differ"] == 0
    ch, vh = _decoder_input(po, 2)
    hp = hw.experiment(ch, vh, (72, 288), 700, 7)["by_warm_up_windows"]
    assert 0 < hp["72"]["starts_that_differ"] < 60 and hp["288"]["starts_that_differ"] == 0
