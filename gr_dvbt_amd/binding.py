import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_HERE, "lib", "libdvbt_hip.so")
_SRC = os.path.join(_HERE, "csrc", "dvbt_hip.hip")

QPSK, QAM16, QAM64 = 0, 1, 2
AUTO = -1          # RxStream: constellation / hierarchy / code_rate from the stream's TPS word
NH = 0
C1_2, C2_3, C3_4, C5_6, C7_8 = 0, 1, 2, 3, 4
T2k, T8k = 0, 1
G1_32, G1_16, G1_8, G1_4 = 0, 1, 2, 3
(TAP_ACQ, TAP_FFT, TAP_EQ, TAP_DEMAP, TAP_SYMDEINT, TAP_BITDEINT, TAP_VITERBI, TAP_DEINT, TAP_RS,
 TAP_TS, TAP_CP_START, TAP_SYMBOL_INDEX, TAP_FREQ_OFFSET, TAP_BITDEINT_LP, TAP_SOFT, TAP_CSI, TAP_BITDEINT_LOG) = range(17)
ALPHA1, ALPHA2, ALPHA4 = 1, 2, 3


class DvbtError(RuntimeError):
    pass


def hipcc():
    """the ROCm compiler driver (no environment variable is consulted: the package reads none)"""
    return "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(_ROOT, "include", "dvbt_hip.h"))
    if not force and os.path.exists(_SO) and all(os.path.getmtime(s) <= os.path.getmtime(_SO) for s in srcs):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-o", _SO, _SRC]
    subprocess.check_call(cmd)
    return _SO


class RxParams(C.Structure):
    _fields_ = [("constellation", C.c_int), ("hierarchy", C.c_int), ("code_rate", C.c_int),
                ("guard_interval", C.c_int), ("transmission_mode", C.c_int), ("include_cell_id", C.c_int),
                ("cell_id", C.c_int), ("snr_db", C.c_float), ("viterbi_bsize", C.c_int),
                ("rs_oracle_compat", C.c_int), ("descramble", C.c_int), ("max_samples", C.c_size_t),
                ("device", C.c_int), ("viterbi_chunk_bytes", C.c_int), ("resample_interp", C.c_int), ("resample_decim", C.c_int),
                ("front_scale", C.c_float), ("soft_decision", C.c_int), ("hier_stream", C.c_int), ("launch_graph", C.c_int), ("front_priority", C.c_int),
                ("viterbi_warm_windows", C.c_int), ("viterbi_verify", C.c_int)]


class RxReport(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_symbols", C.c_int32), ("first_out_symbol", C.c_int32),
                ("n_out_symbols", C.c_int32), ("cp_start0", C.c_int32), ("first_call", C.c_int32),
                ("n_viterbi_bytes", C.c_int64), ("n_rs_items", C.c_int64), ("n_rs_bytes", C.c_int64),
                ("n_ts_bytes", C.c_int64), ("rs_fail_words", C.c_int32), ("rs_corrected_symbols", C.c_int32),
                ("resume_sample", C.c_int64), ("segment_offset", C.c_int64), ("stream_symbol_offset", C.c_int64),
                ("ts_first_packet", C.c_int64), ("stream_rs_items", C.c_int64), ("tps_bits", C.c_uint64), ("tps_valid", C.c_int32),
                ("tps_length_indicator", C.c_int32), ("tps_constellation", C.c_int32), ("tps_hierarchy", C.c_int32),
                ("tps_code_rate_hp", C.c_int32), ("tps_code_rate_lp", C.c_int32), ("tps_guard_interval", C.c_int32),
                ("tps_transmission_mode", C.c_int32), ("tps_cell_id", C.c_int32), ("tps_mismatch", C.c_int32),
                ("n_lock_periods", C.c_int32), ("total_symbols", C.c_int32)]


class LockPeriod(C.Structure):
    _fields_ = [("offset", C.c_int64), ("first_call", C.c_int32), ("cp_start0", C.c_int32), ("n_symbols", C.c_int32),
                ("first_out_symbol", C.c_int32)]


class ViterbiProof(C.Structure):
    _fields_ = [("chunks", C.c_int64), ("decoded_again", C.c_int64), ("sequential", C.c_int64), ("not_proven", C.c_int64)]


class RxCut(C.Structure):
    _fields_ = [("stream_symbol_offset", C.c_int64), ("start_delay_symbols", C.c_int32), ("descr_call_phase", C.c_int32)]


class Dims(C.Structure):
    _fields_ = [("fft_length", C.c_int), ("cp_length", C.c_int), ("Kmax", C.c_int), ("payload_length", C.c_int),
                ("zeros_on_left", C.c_int), ("m", C.c_int), ("cr_k", C.c_int), ("cr_n", C.c_int),
                ("norm", C.c_float), ("ntraceback", C.c_int), ("info_bits_per_symbol", C.c_int)]


_lib = None


def lib():
    """Load libdvbt_hip.so. Fails loudly when the extension is missing: there is no fallback.

    Note for processes that also use PyTorch: import torch BEFORE the first call of this function, so that
    both share torch's bundled HIP runtime (two HIP runtimes in one process cannot both open the GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise DvbtError(f"{_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(the product path has no CPU fallback)")
        L = C.CDLL(_SO)
        L.dvbt_last_error.restype = C.c_char_p
        L.dvbt_version.restype = C.c_char_p
        L.dvbt_rx_create.argtypes = [C.POINTER(RxParams), C.POINTER(C.c_void_p)]
        L.dvbt_rx_segment_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(RxReport)]
        L.dvbt_rx_segment_run_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(RxReport)]
        L.dvbt_rx_segment_enqueue_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dvbt_rx_segment_finish.argtypes = [C.c_void_p, C.POINTER(RxReport)]
        L.dvbt_rx_read_tap.restype = C.c_int64
        L.dvbt_rx_read_tap.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.dvbt_rx_tap_device_ptr.restype = C.c_void_p
        L.dvbt_rx_tap_device_ptr.argtypes = [C.c_void_p, C.c_int]
        L.dvbt_rx_stage_ms.restype = C.c_double
        L.dvbt_rx_stage_ms.argtypes = [C.c_void_p, C.c_char_p]
        L.dvbt_rx_enable_timing.argtypes = [C.c_void_p, C.c_int]
        L.dvbt_rx_enable_taps.argtypes = [C.c_void_p, C.c_int]
        L.dvbt_rx_set_cut.argtypes = [C.c_void_p, C.POINTER(RxCut)]
        L.dvbt_rx_destroy.argtypes = [C.c_void_p]
        L.dvbt_rx_lock_periods.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.dvbt_get_dims.argtypes = [C.c_int] * 5 + [C.POINTER(Dims)]
        _lib = L
    return _lib


def _chk(r):
    if r < 0:
        raise DvbtError(f"libdvbt_hip error {r}: {lib().dvbt_last_error().decode()}")
    return r


def device_count():
    return lib().dvbt_device_count()


def get_dims(constellation, code_rate, mode, guard=G1_32, hierarchy=NH):
    d = Dims()
    _chk(lib().dvbt_get_dims(constellation, hierarchy, code_rate, guard, mode, C.byref(d)))
    return d


class Rx:
    """Device-resident DVB-T receive chain (segment API of include/dvbt_hip.h)."""

    _TAP_DTYPE = {TAP_ACQ: np.complex64, TAP_FFT: np.complex64, TAP_EQ: np.complex64, TAP_CP_START: np.int32,
                  TAP_SYMBOL_INDEX: np.int32, TAP_FREQ_OFFSET: np.int32, TAP_SOFT: np.int8, TAP_CSI: np.float32}

    def __init__(self, constellation, code_rate, mode, max_samples, guard=G1_32, hierarchy=NH, snr_db=30.0,
                 viterbi_bsize=768, rs_oracle_compat=0, descramble=1, device=0, viterbi_chunk_bytes=0, taps=False,
                 resample=(0, 0), front_scale=0.0, soft_decision=0, hier_stream=0, launch_graph=0, front_priority=0, viterbi_warm_windows=0, viterbi_verify=0):
        self.L = lib()
        self.p = RxParams(constellation, hierarchy, code_rate, guard, mode, 0, 0, snr_db, viterbi_bsize,
                          rs_oracle_compat, descramble, max_samples, device, viterbi_chunk_bytes, resample[0], resample[1], front_scale, soft_decision, hier_stream, launch_graph, front_priority,
                          viterbi_warm_windows, viterbi_verify)
        self.h = C.c_void_p()
        _chk(self.L.dvbt_rx_create(C.byref(self.p), C.byref(self.h)))
        self.dims = get_dims(constellation, code_rate, mode, guard, hierarchy)
        if taps:
            _chk(self.L.dvbt_rx_enable_taps(self.h, int(taps)))          # True / 1: the debug taps; 2: plus the per-lock-period log of the decoder's input
        self.report = None

    def run(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        rep = RxReport()
        _chk(self.L.dvbt_rx_segment_run(self.h, iq.ctypes.data_as(C.c_void_p), len(iq), C.byref(rep)))
        self.report = rep
        return rep

    def set_cut(self, stream_symbol_offset):
        """Declare the following segments as the continuation of a cut stream (include/dvbt_hip.h: dvbt_rx_set_cut)."""
        cut = RxCut(int(stream_symbol_offset), 0, 0)
        _chk(self.L.dvbt_rx_set_cut(self.h, C.byref(cut)))

    def run_device(self, dptr, nsamples, stream=None):
        """dvbt_rx_segment_run_device: synchronous, follows every loss of the CP lock inside the segment"""
        rep = RxReport()
        _chk(self.L.dvbt_rx_segment_run_device(self.h, C.c_void_p(dptr), nsamples, C.c_void_p(stream) if stream else None, C.byref(rep)))
        self.report = rep
        return rep

    def enqueue_device(self, dptr, nsamples, stream=None):
        _chk(self.L.dvbt_rx_segment_enqueue_device(self.h, C.c_void_p(dptr), nsamples,
                                                   C.c_void_p(stream) if stream else None))

    def finish(self):
        rep = RxReport()
        _chk(self.L.dvbt_rx_segment_finish(self.h, C.byref(rep)))
        self.report = rep
        return rep

    def tap(self, tap):
        r = self.report
        d = self.dims
        sizes = {TAP_ACQ: r.n_symbols * d.fft_length * 8, TAP_FFT: r.n_symbols * d.fft_length * 8,
                 TAP_EQ: r.n_out_symbols * d.payload_length * 8, TAP_DEMAP: r.n_out_symbols * d.payload_length,
                 TAP_SYMDEINT: r.n_out_symbols * d.payload_length, TAP_BITDEINT: r.n_out_symbols * d.payload_length,
                 TAP_VITERBI: r.n_viterbi_bytes, TAP_DEINT: r.n_rs_bytes // 188 * 204, TAP_RS: r.n_rs_bytes,
                 TAP_TS: r.n_ts_bytes, TAP_CP_START: r.n_symbols * 4, TAP_SYMBOL_INDEX: max(r.n_symbols - 1, 0) * 4,
                 TAP_FREQ_OFFSET: max(r.n_symbols - 1, 0) * 4, TAP_BITDEINT_LP: r.n_out_symbols * d.payload_length,
                 TAP_SOFT: r.n_out_symbols * d.payload_length * d.m, TAP_CSI: r.n_out_symbols * d.payload_length * 4}
        nbytes = max(int(sizes[tap]), 0)
        buf = np.zeros(nbytes, np.uint8)
        if nbytes:
            n = _chk(self.L.dvbt_rx_read_tap(self.h, tap, buf.ctypes.data_as(C.c_void_p), nbytes))
            buf = buf[:n]
        out = buf.view(self._TAP_DTYPE.get(tap, np.uint8))
        if tap in (TAP_ACQ, TAP_FFT):
            out = out.reshape(-1, d.fft_length)
        elif tap in (TAP_EQ, TAP_DEMAP, TAP_SYMDEINT, TAP_BITDEINT, TAP_BITDEINT_LP, TAP_CSI):
            out = out.reshape(-1, d.payload_length)
        elif tap == TAP_SOFT:
            out = out.reshape(-1, d.payload_length * d.m)
        return out

    def lock_periods(self):
        """[(offset, first_call, cp_start0, n_symbols, first_out_symbol)] of the last run() / run_device()"""
        n = _chk(self.L.dvbt_rx_lock_periods(self.h, None, 0))
        buf = (LockPeriod * max(n, 1))()
        _chk(self.L.dvbt_rx_lock_periods(self.h, buf, n))
        return [(b.offset, b.first_call, b.cp_start0, b.n_symbols, b.first_out_symbol) for b in buf[:n]]

    def walk_stats(self):
        """(passes through the one-launch tracker, passes through the general kernels, calls per chunk of the one-launch tracker, its longest window)"""
        class W(C.Structure):
            _fields_ = [("small_passes", C.c_int64), ("general_passes", C.c_int64), ("small_chunk_calls", C.c_int32), ("small_max_calls", C.c_int32)]
        w = W()
        self.L.dvbt_rx_walk_stats.argtypes = [C.c_void_p, C.POINTER(W)]
        _chk(self.L.dvbt_rx_walk_stats(self.h, C.byref(w)))
        return w.small_passes, w.general_passes, w.small_chunk_calls, w.small_max_calls

    def period_taps(self):
        """taps=2: [(decoder input of the lock period (uint8), offset of its decoded bytes in the VITERBI tap, their count)] for every lock period of the
        last run() / run_device() that reached the Viterbi decoder"""
        class P(C.Structure):
            _fields_ = [("bitdeint_offset", C.c_int64), ("bitdeint_bytes", C.c_int64), ("viterbi_offset", C.c_int64), ("viterbi_bytes", C.c_int64)]
        self.L.dvbt_rx_period_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        n = _chk(self.L.dvbt_rx_period_taps(self.h, None, 0))
        buf = (P * max(n, 1))()
        _chk(self.L.dvbt_rx_period_taps(self.h, buf, n))
        total = max([b.bitdeint_offset + b.bitdeint_bytes for b in buf[:n]] + [0])
        log = np.zeros(total, np.uint8)
        if total:
            _chk(self.L.dvbt_rx_read_tap(self.h, TAP_BITDEINT_LOG, log.ctypes.data_as(C.c_void_p), total))
        return [(log[b.bitdeint_offset:b.bitdeint_offset + b.bitdeint_bytes], int(b.viterbi_offset), int(b.viterbi_bytes)) for b in buf[:n]]

    def viterbi_proof(self):
        """what the proof + repair passes of the last launch of the Viterbi decoder did: dict(chunks, decoded_again, sequential, not_proven) -- not_proven is the
        final check's count (viterbi_verify >= 1), -1 when it did not run"""
        p = ViterbiProof()
        self.L.dvbt_rx_viterbi_proof.argtypes = [C.c_void_p, C.POINTER(ViterbiProof)]
        _chk(self.L.dvbt_rx_viterbi_proof(self.h, C.byref(p)))
        return {"chunks": int(p.chunks), "decoded_again": int(p.decoded_again), "sequential": int(p.sequential), "not_proven": int(p.not_proven)}

    def viterbi_proof_total(self):
        """the passes' counters summed over every launch of the handle's decoder since it was created: dict(chunks, decoded_again, sequential, not_proven = -1)"""
        p = ViterbiProof()
        self.L.dvbt_rx_viterbi_proof_total.argtypes = [C.c_void_p, C.POINTER(ViterbiProof)]
        _chk(self.L.dvbt_rx_viterbi_proof_total(self.h, C.byref(p)))
        return {"chunks": int(p.chunks), "decoded_again": int(p.decoded_again), "sequential": int(p.sequential), "not_proven": int(p.not_proven)}

    def viterbi_check(self):
        """viterbi_verify >= 1: (chunks of the last launch of the Viterbi decoder, chunks NOT proven equal to the streaming decoder when the launch ends)"""
        a, b = C.c_int64(), C.c_int64()
        self.L.dvbt_rx_viterbi_check.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _chk(self.L.dvbt_rx_viterbi_check(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def tap_device_ptr(self, tap):
        return self.L.dvbt_rx_tap_device_ptr(self.h, tap)

    def enable_timing(self, on=True):
        """True / 1: HIP events around every stage; 2: around the decoder only (stage_ms("viterbi")); False / 0: none"""
        _chk(self.L.dvbt_rx_enable_timing(self.h, int(on)))

    def stage_ms(self, name):
        return self.L.dvbt_rx_stage_ms(self.h, name.encode())

    def close(self):
        if self.h:
            self.L.dvbt_rx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StreamParams(C.Structure):
    _fields_ = [("rx", RxParams), ("segment_superframes", C.c_int), ("rank", C.c_int), ("world", C.c_int), ("ts_ring_bytes", C.c_int64), ("borrow_device_pushes", C.c_int), ("chains", C.c_int)]


class StreamInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("pieces_in_flight", C.c_int32), ("finished", C.c_int32),
                ("samples_pushed", C.c_int64), ("ts_bytes_decoded", C.c_int64), ("ts_bytes_ready", C.c_int64), ("ts_bytes_pulled", C.c_int64),
                ("first_superframe_call", C.c_int64), ("first_ts_packet", C.c_int64),
                ("constellation", C.c_int32), ("hierarchy", C.c_int32), ("code_rate", C.c_int32), ("auto_configured", C.c_int32), ("samples_released", C.c_int64), ("in_walk", C.c_int32)]


class RxStream:
    """dvbt_rx_stream_*: push samples in calls of any size, pull the TS in order; the bytes are those of one chain over the whole stream."""

    def __init__(self, constellation, code_rate, mode, segment_superframes=0, guard=G1_32, hierarchy=NH, snr_db=30.0, viterbi_bsize=768,
                 rs_oracle_compat=0, device=0, rank=0, world=0, soft_decision=0, ts_ring_bytes=0, borrow=0, viterbi_warm_windows=0, viterbi_verify=0, chains=0):
        self.L = lib()
        for fn in ("create", "push", "push_device", "finish", "status"):
            getattr(self.L, f"dvbt_rx_stream_{fn}").restype = C.c_int
        self.L.dvbt_rx_stream_create.argtypes = [C.POINTER(StreamParams), C.POINTER(C.c_void_p)]
        self.L.dvbt_rx_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.dvbt_rx_stream_push_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        self.L.dvbt_rx_stream_pull.restype = C.c_int64
        self.L.dvbt_rx_stream_pull.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.dvbt_rx_stream_finish.argtypes = [C.c_void_p]
        self.L.dvbt_rx_stream_status.argtypes = [C.c_void_p, C.POINTER(StreamInfo)]
        self.L.dvbt_rx_stream_destroy.argtypes = [C.c_void_p]
        rx = RxParams(constellation, hierarchy, code_rate, guard, mode, 0, 0, snr_db, viterbi_bsize, rs_oracle_compat, 1, 0, device, 0, 0, 0, 0.0, soft_decision)
        rx.viterbi_warm_windows = viterbi_warm_windows
        rx.viterbi_verify = viterbi_verify
        self.p = StreamParams(rx, segment_superframes, rank, world, ts_ring_bytes, borrow, chains)
        self.L.dvbt_rx_stream_pull_chunk.restype = C.c_int64
        self.L.dvbt_rx_stream_pull_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64)]
        self.h = C.c_void_p()
        _chk(self.L.dvbt_rx_stream_create(C.byref(self.p), C.byref(self.h)))
        self._out = np.empty(1 << 22, np.uint8)

    def push(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        _chk(self.L.dvbt_rx_stream_push(self.h, iq.ctypes.data_as(C.c_void_p), len(iq)))

    def push_device(self, dptr, nsamples, stream=None):
        _chk(self.L.dvbt_rx_stream_push_device(self.h, C.c_void_p(dptr), nsamples, C.c_void_p(stream) if stream else None))

    def pull(self, max_bytes=None):
        """the TS bytes that are ready (all of them unless max_bytes is given), as a numpy array"""
        chunks = []
        left = max_bytes
        while left is None or left > 0:
            cap = len(self._out) if left is None else min(len(self._out), left)
            n = _chk(self.L.dvbt_rx_stream_pull(self.h, self._out.ctypes.data_as(C.c_void_p), cap))
            if n <= 0:
                break
            chunks.append(self._out[:n].copy())
            if left is not None:
                left -= n
        return np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)

    def pull_chunks(self):
        """[(first packet index in the stream's TS, uint8 array)] of what is ready: this rank's packets of a sharded stream"""
        out = []
        while True:
            fp = C.c_int64()
            n = _chk(self.L.dvbt_rx_stream_pull_chunk(self.h, self._out.ctypes.data_as(C.c_void_p), len(self._out) // 188 * 188, C.byref(fp)))
            if n <= 0:
                break
            out.append((fp.value, self._out[:n].copy()))
        return out

    def finish(self):
        _chk(self.L.dvbt_rx_stream_finish(self.h))

    def trace(self):
        """dvbt_rx_stream_trace: what the stream did (walk windows, epochs, pieces that left their epoch), as text"""
        self.L.dvbt_rx_stream_trace.restype = C.c_int64
        self.L.dvbt_rx_stream_trace.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        buf = C.create_string_buffer(1 << 17)
        self.L.dvbt_rx_stream_trace(self.h, buf, len(buf))
        return buf.value.decode()

    def viterbi_proof(self):
        """the Viterbi stage's proof + repair passes summed over the stream's launches so far: dict(chunks, decoded_again, sequential, not_proven = -1)"""
        p = ViterbiProof()
        self.L.dvbt_rx_stream_viterbi_proof.argtypes = [C.c_void_p, C.POINTER(ViterbiProof)]
        _chk(self.L.dvbt_rx_stream_viterbi_proof(self.h, C.byref(p)))
        return {"chunks": int(p.chunks), "decoded_again": int(p.decoded_again), "sequential": int(p.sequential), "not_proven": int(p.not_proven)}

    def info(self):
        i = StreamInfo()
        _chk(self.L.dvbt_rx_stream_status(self.h, C.byref(i)))
        return i

    def close(self):
        if self.h:
            self.L.dvbt_rx_stream_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ per-block C ABI (one triple per reference block)
TAG_SYNC_START, TAG_SUPERFRAME_START, TAG_SYMBOL_INDEX = 1, 2, 3


class Tag(C.Structure):
    _fields_ = [("rel_offset", C.c_int64), ("key", C.c_int32), ("value", C.c_int32)]


class Sideband(C.Structure):
    _fields_ = [("in_tags", C.POINTER(Tag)), ("n_in_tags", C.c_int), ("out_tags", C.POINTER(Tag)), ("out_cap", C.c_int),
                ("n_out_tags", C.c_int), ("n_consumed", C.c_int)]


def _params(fields):
    return type("P", (C.Structure,), {"_fields_": fields})


_I = C.c_int
BLOCK_PARAMS = {
    "ofdm_sym_acquisition": _params([("blocks", _I), ("fft_length", _I), ("occupied_tones", _I), ("cp_length", _I), ("snr", C.c_float)]),
    "fft": _params([("fft_size", _I), ("forward", _I), ("shift", _I)]),
    "demod_reference_signals": _params([(n, _I) for n in ("itemsize", "ninput", "noutput", "constellation", "hierarchy", "code_rate_hp",
                                                          "code_rate_lp", "guard_interval", "transmission_mode", "include_cell_id", "cell_id")]),
    "demap": _params([("nsize", _I), ("constellation", _I), ("hierarchy", _I), ("transmission_mode", _I), ("gain", C.c_float)]),
    "symbol_inner_interleaver": _params([("nsize", _I), ("transmission_mode", _I), ("direction", _I)]),
    "bit_inner_deinterleaver": _params([("nsize", _I), ("constellation", _I), ("hierarchy", _I), ("transmission_mode", _I)]),
    "viterbi_decoder": _params([("constellation", _I), ("hierarchy", _I), ("code_rate", _I), ("bsize", _I), ("S0", _I), ("SK", _I)]),
    "convolutional_deinterleaver": _params([("blocks", _I), ("I", _I), ("M", _I)]),
    "reed_solomon_dec": _params([(n, _I) for n in ("p", "m", "gfpoly", "n", "k", "t", "s", "blocks", "oracle_compat")]),
    "energy_descramble": _params([("nblocks", _I)]),
    "resampler": _params([("interpolation", _I), ("decimation", _I), ("scale", C.c_float)]),
}


class Block:
    """Generic driver of dvbt_<name>_{create,forecast,work,destroy}; mirrors gr::dvbt::<name>::make(...)."""

    def __init__(self, name, *args):
        self.L = lib()
        self.name = name
        self.params = BLOCK_PARAMS[name](*args)
        self.h = C.c_void_p()
        _chk(getattr(self.L, f"dvbt_{name}_create")(C.byref(self.params), C.byref(self.h)))
        self._work = getattr(self.L, f"dvbt_{name}_work")
        self._work.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Sideband)]
        self._work_device = getattr(self.L, f"dvbt_{name}_work_device", None)
        if self._work_device is not None:
            self._work_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Sideband), C.c_void_p]
        self._tout = (Tag * 4096)()

    def forecast(self, noutput_items):
        n = C.c_int()
        _chk(getattr(self.L, f"dvbt_{self.name}_forecast")(self.h, noutput_items, C.byref(n)))
        return n.value

    def work(self, noutput_items, ninput_items, inp, out, tags=()):
        """returns (items produced, items consumed, [(rel_offset, key, value)] attached to the output)"""
        tin = (Tag * max(len(tags), 1))(*[Tag(*t) for t in tags])
        tout = (Tag * 4096)()
        sb = Sideband(tin, len(tags), tout, 4096, 0, 0)
        r = _chk(self._work(self.h, noutput_items, ninput_items, inp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(sb)))
        return r, sb.n_consumed, [(tout[i].rel_offset, tout[i].key, tout[i].value) for i in range(min(sb.n_out_tags, 4096))]

    def work_device(self, noutput_items, ninput_items, in_ptr, out_ptr, tags=(), stream=None):
        """dvbt_<name>_work_device: device pointers (ints) in and out, the caller's HIP stream; same return as work()"""
        tin = (Tag * max(len(tags), 1))(*[Tag(*t) for t in tags])
        tout = self._tout
        sb = Sideband(tin, len(tags), tout, 4096, 0, 0)
        r = _chk(self._work_device(self.h, noutput_items, ninput_items, C.c_void_p(in_ptr), C.c_void_p(out_ptr), C.byref(sb),
                                   C.c_void_p(stream) if stream else None))
        return r, sb.n_consumed, [(tout[i].rel_offset, tout[i].key, tout[i].value) for i in range(min(sb.n_out_tags, 4096))]

    def viterbi_proof(self):
        """viterbi_decoder only: the proof + repair passes' counters summed over the calls since the last reset"""
        p = ViterbiProof()
        self.L.dvbt_viterbi_decoder_proof.argtypes = [C.c_void_p, C.POINTER(ViterbiProof)]
        _chk(self.L.dvbt_viterbi_decoder_proof(self.h, C.byref(p)))
        return {"chunks": int(p.chunks), "decoded_again": int(p.decoded_again), "sequential": int(p.sequential), "not_proven": int(p.not_proven)}

    def close(self):
        if self.h:
            fn = getattr(self.L, f"dvbt_{self.name}_destroy")
            fn.argtypes = [C.c_void_p]
            fn(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
