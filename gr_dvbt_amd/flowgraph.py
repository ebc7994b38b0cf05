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

    def __init__(self, constellation, code_rate, mode_t, n_samples, mode="device", call_symbols=4, snr_db=30.0, bsize=768, register_buffers=False):
        """register_buffers (host mode): page-lock the source and every block's output buffer once (dvbt_host_register), as a GNU Radio shell would its
        flowgraph buffers: the host-pointer entries then DMA straight from / to them instead of staging every item through pinned memory of the handle"""
        self.mode = mode
        self.registered = []
        self.register_buffers = register_buffers and mode == "host"
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
        if self.register_buffers:
            for st in self.stages:
                self._register(st.out)

    def _register(self, arr):
        import ctypes as C
        L = b.lib()
        L.dvbt_host_register.argtypes = [C.c_void_p, C.c_size_t]
        L.dvbt_host_unregister.argtypes = [C.c_void_p]
        if L.dvbt_host_register(arr.ctypes.data, arr.nbytes) == 0:
            self.registered.append(arr.ctypes.data)

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

    def _setup(self, iq):
        d = self.dims
        N, cp, P, cs = d.fft_length, d.cp_length, d.payload_length, self.call_symbols
        if self.mode == "device":
            self.src = self.torch.from_numpy(np.ascontiguousarray(iq).view(np.uint8)).cuda()
        else:
            self.src = np.ascontiguousarray(iq).view(np.uint8)
            if self.register_buffers:
                self._register(self.src)
        self.stages[0].w = len(iq)
        # items per call, per stage (output side): `cs` OFDM symbols' worth
        vit_blocks = max(1, cs * P // self.vit_in_block)
        self.out_mult = [1, 1, 1, 1, 1, 1, self.vit_out_mult, 2, 1, 4 * 1504]          # set_output_multiple of the reference blocks
        self.per_call = [cs, cs, cs, cs, cs, cs, vit_blocks * self.vit_out_mult, max(2, (cs * d.info_bits_per_symbol // 8 // 1632) & ~1),
                         max(2, cs * d.info_bits_per_symbol // 8 // 1632), max(1, cs * d.info_bits_per_symbol // 8 // (4 * 1504)) * 4 * 1504]

    def _step(self, k, lock=None, drain=False):
        """one general_work call of stage k if its input allows one; returns True when something was consumed or produced.  lock: a threading.Lock that
        guards what neighbouring stages share (items available, the tag list) in the threaded driver."""
        d = self.dims
        N, cp, P = d.fft_length, d.cp_length, d.payload_length
        S, st = self.stages, self.stages[k]
        in_buf = self.src if k == 0 else S[k - 1].out
        if lock:
            lock.acquire()
        avail = st.w - st.r
        tags_all = list(st.tags)
        if lock:
            lock.release()
        nout = min(self.per_call[k], st.out_cap - st.produced)
        if nout <= 0 or avail <= 0:
            return False
        need = st.blk.forecast(nout)
        # GNU Radio's executor halves the output request (down to the block's output multiple) while the forecast cannot be met
        # (gnuradio-runtime/lib/block_executor.cc "try_again"); only then is the block blocked on input.  Here that is done once the upstream block has
        # finished (drain): while the stream flows a block is simply offered what has arrived (its buffer would hold more than one call's worth in a
        # flowgraph), so that the steady-state call size stays `call_symbols`
        mult = self.out_mult[k]
        while drain and need > avail and nout // 2 >= mult:
            nout = (nout // 2) // mult * mult
            need = st.blk.forecast(nout)
        if k == 0:
            if avail < 2 * N + cp + 16:
                return False
            nin = min(avail, (nout - 1) * (N + cp) + 2 * N + cp + 16)
        elif k == 6:
            nin = min(avail, need)
            if nin < self.vit_in_block:
                return False
        elif k == 7:
            nin = min(avail, need)
            if nin < 2 * 1632:
                return False
        else:
            nin = min(avail, need)
        tags_rel = [(o - st.r, key, v) for (o, key, v) in tags_all if st.r <= o < st.r + nin]
        produced, consumed, tout = self._call(k, nout, nin, in_buf, st.r * st.in_item, tags_rel)
        st.calls += 1
        if consumed == 0 and produced == 0:
            return False
        # tags travel to the next stage at absolute offsets (x payload through vector_to_stream).  Blocks with one
        # output item per input item pass the tags of their input on, as GNU Radio's default propagation policy does
        if lock:
            lock.acquire()
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
        if lock:
            lock.release()
        return True

    def _result(self):
        last = self.stages[-1]
        if self.mode == "device":
            self.stream.synchronize()
            return last.out[:last.produced].cpu().numpy()
        return last.out[:last.produced].copy()

    def run(self, iq):
        """iq: complex64 samples (numpy).  Returns the TS bytes (numpy uint8).  ONE thread calls the ten blocks in turn."""
        self._setup(iq)
        for drain in (False, True):                      # the stream flows; then every block finishes on what is left
            progress = True
            while progress:
                progress = False
                for k in range(len(self.stages)):
                    while self._step(k, None, drain):
                        progress = True
        return self._result()

    def run_threaded(self, iq):
        """The same flowgraph under a thread-per-block scheduler, which is what GNU Radio's is (gr::thread_body_wrapper / tpb_thread_body: every block's
        general_work runs in its own thread, woken when a neighbour has produced or consumed).  The ten blocks' calls -- host copies, launches, the
        synchronisation at the end of a host-pointer call -- overlap; the throughput is the slowest block's, not the sum.  (ctypes releases the GIL for the
        duration of a call; the output buffers hold the whole stream, so no block ever waits for room.)  host mode only: device-mode blocks share one HIP stream."""
        import threading
        self._setup(iq)
        n = len(self.stages)
        lock = threading.Lock()
        cv = threading.Condition(lock)
        done = [False] * n
        errors = []

        def body(k):
            try:
                while True:
                    if self._step(k, lock):
                        with cv:
                            cv.notify_all()
                        continue
                    with cv:
                        # nothing to do: finished if upstream is (and a last look found nothing), else wait for upstream to produce
                        if k == 0 or done[k - 1]:
                            upstream_final = True
                        else:
                            upstream_final = False
                            cv.wait(0.002)
                    if upstream_final:
                        if not self._step(k, lock, True):
                            break
                        with cv:
                            cv.notify_all()
            except Exception as e:        # pragma: no cover
                errors.append(e)
            finally:
                with cv:
                    done[k] = True
                    cv.notify_all()

        ths = [threading.Thread(target=body, args=(k,)) for k in range(n)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errors:
            raise errors[0]
        return self._result()

    def close(self):
        for p in self.registered:
            b.lib().dvbt_host_unregister(p)
        self.registered = []
        for st in self.stages:
            st.blk.close()
