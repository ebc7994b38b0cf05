"""The receive flowgraph of apps/dvbt_rx_demo*.grc driven block by block through the per-block C ABI -- the drop-in path.

GNU Radio's scheduler is replaced by the smallest thing that keeps its contract: every block is called with a window of
its input and room for `call_symbols` OFDM symbols' worth of output (forecast() decides whether the window suffices),
consumes what it reports (consume_each) and attaches tags at absolute output offsets; the stock vector_to_stream between
the bit de-interleaver and the Viterbi decoder is the factor `payload` on item counts and tag offsets.

mode "host":   dvbt_<blk>_work, host buffers (numpy) -- what a GNU Radio block shell does: every item crosses PCIe twice per block.
mode "device": dvbt_<blk>_work_device, device buffers (torch CUDA tensors), one HIP stream -- adjacent HIP blocks hand
               items over in HBM; the only host traffic is the input, the TS and the tags.

Used by tests/test_gpu_flowgraph.py (TS identical to the segment API and the oracle) and by bench.py --mode blocks.
"""
import numpy as np

from . import binding as b


class _Stage:
    def __init__(self, block, in_item, out_item, in_cap_items, out_cap_items, mode, torch=None, min_extra=0):
        self.blk, self.in_item, self.out_item, self.mode = block, in_item, out_item, mode
        self.tags = []            # (absolute input item offset, key, value), ascending
        self.r = 0                # items consumed
        self.w = 0                # items available (written by upstream)
        self.min_extra = min_extra
        self.out_cap = out_cap_items
        if mode == "device":
            self.out = torch.empty(out_cap_items * out_item + 64, dtype=torch.uint8, device="cuda")
        else:
            self.out = np.empty(out_cap_items * out_item + 64, dtype=np.uint8)
        self.produced = 0
        self.calls = 0


class RxFlowgraph:
    """ofdm_sym_acquisition -> fft -> demod_reference_signals -> dvbt_demap -> symbol_inner_interleaver(0) ->
    bit_inner_deinterleaver -> [vector_to_stream] -> viterbi_decoder -> convolutional_deinterleaver -> reed_solomon_dec ->
    energy_descramble, with the parameters of the demo flowgraphs."""

    def __init__(self, constellation, code_rate, mode_t, n_samples, mode="device", call_symbols=4, snr_db=30.0, bsize=768):
        self.mode = mode
        self.torch = None
        if mode == "device":
            import torch
            self.torch = torch
            self.stream = torch.cuda.Stream()
        d = self.dims = b.get_dims(constellation, code_rate, mode_t)
        N, cp, P = d.fft_length, d.cp_length, d.payload_length
        self.call_symbols = call_symbols
        nsym = n_samples // (N + cp) + 2
        self.vit_out_mult = bsize * d.cr_k // 8
        vit_in_block = bsize * d.cr_n // d.m
        nbytes = nsym * P * d.m * d.cr_k // (8 * d.cr_n) + 4096
        mk = lambda name, *a: b.Block(name, *a)
        S = lambda blk, i, o, ci, co, **k: _Stage(blk, i, o, ci, co, mode, self.torch, **k)
        self.stages = [
            S(mk("ofdm_sym_acquisition", 1, N, d.Kmax + 1, cp, snr_db), 8, N * 8, n_samples, nsym),
            S(mk("fft", N, 1, 1), N * 8, N * 8, nsym, nsym),
            S(mk("demod_reference_signals", 8, N, P, constellation, b.NH, code_rate, code_rate, b.G1_32, mode_t, 0, 0), N * 8, P * 8, nsym, nsym),
            S(mk("demap", P, constellation, b.NH, mode_t, 1.0), P * 8, P, nsym, nsym),
            S(mk("symbol_inner_interleaver", P, mode_t, 0), P, P, nsym, nsym),
            S(mk("bit_inner_deinterleaver", P, constellation, b.NH, mode_t), P, P, nsym, nsym),
            S(mk("viterbi_decoder", constellation, b.NH, code_rate, bsize, 0, -1), 1, 1, nsym * P, nbytes),
            S(mk("convolutional_deinterleaver", 136, 12, 17), 1, 1632, nbytes, nbytes // 1632 + 2),
            S(mk("reed_solomon_dec", 2, 8, 0x11d, 255, 239, 8, 51, 8, 0), 1632, 1504, nbytes // 1632 + 2, nbytes // 1632 + 2),
            S(mk("energy_descramble", 8), 1504, 1, nbytes // 1632 + 2, nbytes),
        ]
        self.vit_in_block = vit_in_block
        self.input = None

    # ---- buffers
    def _ptr(self, buf, byte_off):
        if self.mode == "device":
            return buf.data_ptr() + byte_off
        return buf[byte_off:]

    def _call(self, k, nout, nin, in_buf, in_off_bytes, tags_rel):
        st = self.stages[k]
        out_off = st.produced * st.out_item
        if self.mode == "device":
            return st.blk.work_device(nout, nin, in_buf.data_ptr() + in_off_bytes, st.out.data_ptr() + out_off, tags_rel, self.stream.cuda_stream)
        return st.blk.work(nout, nin, in_buf[in_off_bytes:], st.out[out_off:], tags_rel)

    def run(self, iq):
        """iq: complex64 samples (numpy).  Returns the TS bytes (numpy uint8)."""
        d = self.dims
        N, cp, P, cs = d.fft_length, d.cp_length, d.payload_length, self.call_symbols
        if self.mode == "device":
            src = self.torch.from_numpy(np.ascontiguousarray(iq).view(np.uint8)).cuda()
        else:
            src = np.ascontiguousarray(iq).view(np.uint8)
        S = self.stages
        S[0].w = len(iq)
        # items per call, per stage (output side): `cs` OFDM symbols' worth
        vit_blocks = max(1, cs * P // self.vit_in_block)
        per_call = [cs, cs, cs, cs, cs, cs, vit_blocks * self.vit_out_mult, max(2, (cs * d.info_bits_per_symbol // 8 // 1632) & ~1),
                    max(2, cs * d.info_bits_per_symbol // 8 // 1632), max(1, cs * d.info_bits_per_symbol // 8 // (4 * 1504)) * 4 * 1504]
        progress = True
        while progress:
            progress = False
            for k, st in enumerate(S):
                in_buf = src if k == 0 else S[k - 1].out
                while True:
                    avail = st.w - st.r
                    nout = min(per_call[k], st.out_cap - st.produced)
                    if nout <= 0 or avail <= 0:
                        break
                    need = st.blk.forecast(nout)
                    # a scheduler offers what it has when the forecast cannot be met at the stream's end; the block decides
                    nin = avail if k != 0 else min(avail, need + 0)
                    if k == 0 and avail < 2 * N + cp + 16:
                        break
                    if k == 0:
                        nin = min(avail, (nout - 1) * (N + cp) + 2 * N + cp + 16)
                    elif k == 6:
                        nin = min(avail, need)
                        if nin < self.vit_in_block:
                            break
                    elif k == 7:
                        nin = min(avail, need)
                        if nin < 2 * 1632:
                            break
                    else:
                        nin = min(avail, need)
                    tags_rel = [(o - st.r, key, v) for (o, key, v) in st.tags if st.r <= o < st.r + nin]
                    produced, consumed, tout = self._call(k, nout, nin, in_buf, st.r * st.in_item, tags_rel)
                    st.calls += 1
                    if consumed == 0 and produced == 0:
                        break
                    # tags travel to the next stage at absolute offsets (x payload through vector_to_stream).  Blocks with one
                    # output item per input item pass the tags of their input on, as GNU Radio's default propagation policy does
                    if k + 1 < len(S):
                        scale = P if k == 5 else 1
                        if k in (1, 3, 4, 5, 8):
                            tout = tout + [t for t in tags_rel if t[0] < consumed]
                        for (o, key, v) in sorted(tout):
                            S[k + 1].tags.append(((st.produced + o) * scale, key, v))
                    st.r += consumed
                    st.tags = [t for t in st.tags if t[0] >= st.r]
                    st.produced += produced
                    if k + 1 < len(S):
                        S[k + 1].w = st.produced * (P if k == 5 else 1)
                    progress = True
        last = S[-1]
        if self.mode == "device":
            self.stream.synchronize()
            return last.out[:last.produced].cpu().numpy()
        return last.out[:last.produced].copy()

    def close(self):
        for st in self.stages:
            st.blk.close()
