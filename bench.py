#!/usr/bin/env python3
"""bench.py -- RX Msamples/s of the DVB-T receive hot path on MI355X (BASELINE.json metric).

One step = one pass of the whole chain (ofdm_sym_acquisition -> FFT -> demod_reference_signals ->
dvbt_demap -> symbol/bit de-interleave -> viterbi_decoder -> convolutional_deinterleaver ->
reed_solomon_dec -> energy_descramble) over one batch of synthetic loopback baseband that is already
resident in HBM.

The batch is ONE stream (1 lead-in superframe + N x --superframes payload superframes).  It is cut at
superframe boundaries into N x --segments pieces (gr_dvbt_amd/multi.py::plan_cuts); every
rank (one process per GPU, torch.distributed / RCCL) generates and decodes only ITS pieces, each on its own
handle and HIP stream, and the decoded packets travel to rank 0 in the design's single collective per step
(multi.gather_pieces).  --pipeline steps are in flight per piece (default 4): the piece's handles take its steps in turn, each on
its own HIP stream, so that the next steps' latency-bound front-end kernels run while the Viterbi decoder of the current step holds the machine
(one step in flight: --pipeline 1; stage_ms_per_piece_solo has those kernel times).  After the timed loop rank 0 stitches the pieces of the last step and compares the TS
with the packets that were transmitted: the bench fails (exit 1) when a single byte differs.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# independent pieces run on separate HIP streams; give the runtime enough hardware queues that two
# streams of one process do not share one (the ROCm default maps them onto 4 queues round-robin)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
REALTIME_MSPS = 64.0 / 7.0       # OFDM elementary rate at the input of ofdm_sym_acquisition
SEED = 20240607
EXPERIMENT_BUILD = False         # set by tools/ab_bench.py alone (attribution builds of the library whose output is wrong on purpose): no verification

WORKLOADS = {"8k_qam64_7_8": ("QAM64", "C7_8", "T8k"), "2k_qam16_1_2": ("QAM16", "C1_2", "T2k"), "8k_qpsk_7_8": ("QPSK", "C7_8", "T8k")}


class _DevView:
    """Zero-copy torch view of a library-owned device buffer."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def workload_cfg(name):
    from oracle import pyoracle as po
    const, cr, mode = (getattr(po, x) for x in WORKLOADS[name])
    return (const, cr, mode), po.cfg(const, cr, mode)


def add_awgn(iq, snr_db, seed, ref_power):
    rng = np.random.RandomState(seed)
    sig = np.sqrt(ref_power / (10 ** (snr_db / 10)) / 2)
    n = (rng.randn(len(iq)) + 1j * rng.randn(len(iq))).astype(np.complex64)
    return (iq + np.float32(sig) * n).astype(np.complex64)


def add_cfo(iq, cfo, N, first_sample):
    """carrier offset of `cfo` subcarrier spacings; the phase is that of the stream's sample index, so pieces cut from one stream stay coherent"""
    n = np.arange(first_sample, first_sample + len(iq), dtype=np.float64)
    return (iq * np.exp(2j * np.pi * cfo / N * n)).astype(np.complex64)


def cpu_baseline(cfg_name, n_superframes=3, all_cores=True):
    """The oracle port (oracle/o_chain.c, -O3 -funroll-loops -msse2) timed on a bounded sample of the same workload on this
    box's host cores: (i) one thread end to end = the reported value; (ii) what a thread-per-block scheduler
    like GNU Radio's could reach = sample / slowest stage; (iii) segment-parallel on all cores (every thread decodes its
    own copy of the sample; ctypes releases the GIL)."""
    import threading
    from oracle import pyoracle as po
    _, c = workload_cfg(cfg_name)
    iq = po.stream_slice(c, n_superframes + 1, 99)
    t0 = time.time()
    r = po.rx(c, iq, want=("ts",))
    dt = time.time() - t0
    stage = [round(x, 3) for x in r["t_stage"]]
    out = {"value": round(len(iq) / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": f"{n_superframes + 1} superframes of {cfg_name} ({len(iq)} samples, {dt:.1f} s), oracle/o_chain.c end to end",
           "stage_seconds": stage,
           "pipeline_parallel_bound": {"value": round(len(iq) / max(r["t_stage"]) / 1e6, 3), "cores": sum(1 for x in r["t_stage"] if x > 0.0005),
                                       "note": "sample / slowest stage (thread-per-block scheduler)"}}
    ncores = os.cpu_count() or 1
    if all_cores and ncores > 1:
        nthr = min(ncores, 16)                                  # bounded: every thread holds its own working set
        ths = [threading.Thread(target=lambda: po.rx(c, iq, want=("ts",))) for _ in range(nthr)]
        t0 = time.time()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dta = time.time() - t0
        out["all_cores"] = {"value": round(nthr * len(iq) / dta / 1e6, 3), "cores": nthr, "note": "segment-parallel, one copy of the sample per thread"}
    return out


def reference_sse2_viterbi():
    """The reference's own SSE2 Viterbi kernels (oracle/_ref, built from /root/reference in the authoring container; the prebuilt .so
    travels to the GPU box) timed natively beside the port on one host core (oracle/o_refbench.c): decoded Mbit/s of
    d_viterbi_butterfly2_sse2 + d_viterbi_get_output_sse2 in the block's calling pattern."""
    import ctypes as C
    from oracle import pyoracle as po
    L = po.lib()
    L.o_ref_viterbi_mbps.restype = C.c_double
    L.o_ref_viterbi_mbps.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
    v = L.o_ref_viterbi_mbps(po._REF.encode(), 200 * 1000 * 1000, 24)
    if v <= 0:
        return None
    return {"value": round(v, 2), "unit": "Mbit/s decoded", "cores": 1, "kind": "reference",
            "sample": "100 M trellis steps through lib/d_viterbi.c (SSE2) as compiled into oracle/_ref/libdviterbi_ref.so"}


def hbm_copy_gbs(torch, device):
    """achievable HBM bandwidth of this box: device-to-device copy of 1 GiB (read + write), best of 5"""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        best = max(best, 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return round(best, 1)


def profiled_traffic(workload, nsf, alg_bytes):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/rNN_pmc_summary_<workload>_<nsf>sf.json, written by
    tools/save_profiles.py from `rocprofv3 --pmc` runs of this command; counters cannot be collected from inside the run).  None when no
    profile of this workload and size is committed or its launch is not the one measured here (algorithmic bytes differ by > 1 %)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_summary_{workload}_{nsf}sf.json")), reverse=True):
        try:
            d = json.load(open(f))
            k = d["kernels"]["viterbi3_kernel"]
            if abs(d.get("viterbi_algorithmic_bytes", 0) - alg_bytes) <= 0.01 * alg_bytes and "hbm_bytes_corrected" in k:
                return int(k["hbm_bytes_corrected"]), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def valu_issue(workload, nsf, alg_bytes, solo_ms, clock_hz, simds=1024, cycles_per_inst=4.0):
    """What actually bounds the dominant kernel: the SIMDs' VALU issue port.  Every wave64 instruction of its mix (packed 16-bit, three-operand, DPP)
    occupies a SIMD for 4 cycles (profiles/r02_ubench_valu.json), so the launch cannot take less than SQ_INSTS_VALU x 4 / 1024 SIMD-cycles.  `frac_pmc` = that
    against the launch's own cycle count in the same PMC passes (GRBM_GUI_ACTIVE is summed over the 8 XCDs); `frac_live` = that against this run's
    solo_launch_ms at the device's clock.  None when no PMC pass of this launch is committed.  Pure arithmetic on a committed file: never raises."""
    import glob
    try:
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_summary_{workload}_{nsf}sf.json")), reverse=True):
            d = json.load(open(f))
            k = d["kernels"]["viterbi3_kernel"]
            if abs(d.get("viterbi_algorithmic_bytes", 0) - alg_bytes) > 0.01 * alg_bytes or "SQ_INSTS_VALU" not in k:
                continue
            insts = float(k["SQ_INSTS_VALU"])
            floor_cycles = insts * cycles_per_inst / simds                  # SIMD-cycles the issue port needs
            out = {"bound": "valu_issue", "wave_instructions_per_launch": int(insts), "cycles_per_instruction": cycles_per_inst, "simds": simds,
                   "source": os.path.relpath(f, ROOT)}
            if k.get("GRBM_GUI_ACTIVE"):
                out["frac_pmc"] = round(floor_cycles / (float(k["GRBM_GUI_ACTIVE"]) / 8.0), 4)
            if solo_ms and solo_ms > 0 and clock_hz and clock_hz > 0:
                out["clock_mhz"] = round(clock_hz / 1e6, 1)
                out["frac_live"] = round(floor_cycles / (solo_ms * 1e-3 * clock_hz), 4)
            return out
    except Exception:
        pass
    return None


def second_kernel(workload, nsf, vit_alg_bytes, reps, d, c, solo, nseg):
    """symbol8k_kernel / symbol2k_kernel (A1 tail + A2 + A3 + A4 in one kernel), the kernel furthest below the roof that binds it: per OFDM symbol it reads the N + cp samples
    (8 B each) and writes the payload's label bytes and the TPS carriers' values (SURVEY 8d rows A1..A4 fused: 67,584 + 6,048 + 544 B at 8k QAM64).  Duration: the `fft` stage of
    the handle's HIP events with ONE step in flight (the drift model's small launches in front of the symbol kernel are inside that interval: ~15 us).  The PMC figures come from
    the same committed passes as the dominant kernel's.  Never raises."""
    import glob
    try:
        name = "symbol8k_kernel" if c.N == 8192 else "symbol2k_kernel"
        nsym = sum(int(r.n_symbols) for r in reps) / nseg
        alg = nsym * ((c.N + c.cp) * 8 + d.payload_length + 8 * 68)
        ms = solo.get("fft") or 0.0
        out = {"kernel": name, "bound": "hbm", "algorithmic_bytes_per_launch": int(alg), "launch_ms": ms, "unit": "GB/s", "peak": HBM_PEAK_GBS}
        if ms > 0:
            out["achieved"] = round(alg / (ms * 1e-3) / 1e9, 1); out["frac"] = round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_summary_{workload}_{nsf}sf.json")), reverse=True):
            k = json.load(open(f))
            if abs(k.get("viterbi_algorithmic_bytes", 0) - vit_alg_bytes) > 0.01 * vit_alg_bytes or name not in k["kernels"]:
                continue
            k = k["kernels"][name]
            if "hbm_bytes_corrected" in k:
                out["traffic"] = int(k["hbm_bytes_corrected"])
            if k.get("SQ_INSTS_VALU") and k.get("GRBM_GUI_ACTIVE"):
                out["valu_issue_frac_pmc"] = round(float(k["SQ_INSTS_VALU"]) * 4.0 / 1024 / (float(k["GRBM_GUI_ACTIVE"]) / 8.0), 4)
            out["source"] = os.path.relpath(f, ROOT)
            break
        return out
    except Exception as e:                                      # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:200]}


NOMINAL_CLOCK_HZ = 2.4e9         # MI355X peak engine clock (MI355X_MICROARCH.md); the microbenchmarks in tools/ read 2400 MHz from hipDeviceAttributeClockRate


def device_clock_hz(torch, local):
    """the device's peak shader clock: what the runtime reports, else the nominal figure (torch 2.10 on ROCm reports none)"""
    try:
        hz = float(getattr(torch.cuda.get_device_properties(local), "clock_rate", 0) or 0) * 1e3      # kHz
    except Exception:
        hz = 0.0
    return hz if hz > 0 else NOMINAL_CLOCK_HZ


def viterbi_check_of(rx):
    """the proof + repair passes of the handle's last launch of the Viterbi decoder (dvbt_rx_viterbi_proof): chunks, chunks the warm-up alone did not prove (decoded again), chunks the
    sequential pass took, chunks not proven when the launch ended (the final check; None when it did not run)"""
    p = rx.viterbi_proof()
    return {"chunks": p["chunks"], "decoded_again": p["decoded_again"], "sequential": p["sequential"], "not_proven_after_repair": p["not_proven"] if p["not_proven"] >= 0 else None}


STAGES = ("acq", "fft", "demod", "inner", "viterbi", "rs", "total")
_STREAMS = []                    # HIP streams of the process, reused by every Job: streams map onto a few hardware queues in creation order, and a
                                 # later Job's fresh streams may land on ONE queue (its steps in flight would then run one after the other)


def _stream(torch, k):
    while len(_STREAMS) <= k:
        _STREAMS.append(torch.cuda.Stream())
    return _STREAMS[k]


class Job:
    """One stream cut into world x segments pieces; this rank's pieces resident in HBM, one handle + stream each."""

    def __init__(self, a, torch, g, dist, rank, world, local, workload, superframes, snr=None, chunk=0, from_file_rate=False, soft=False, cfo=0.0):
        from oracle import pyoracle as po
        from gr_dvbt_amd import multi
        self.torch, self.g, self.dist, self.rank, self.world, self.local, self.multi, self.po = torch, g, dist, rank, world, local, multi, po
        self.args = a
        # where the collectives' tensors live: device memory under RCCL (backend nccl), host memory under gloo (the gather is then staged through pinned host buffers)
        self.host_staged = bool(dist) and dist.get_backend() == "gloo"
        self.cdev = "cpu" if self.host_staged else f"cuda:{local}"
        (const, cr, mode), c = workload_cfg(workload)
        self.c, self.workload, self.snr = c, workload, snr
        self.dims = d = g.get_dims(const, cr, mode)
        self.nsf = 1 + world * superframes                      # lead-in superframe + the payload
        self.seed = SEED
        self.n_total = po.stream_len(c, self.nsf)
        L = c.N + c.cp
        nseg = max(1, a.segments)
        snr_db = 30.0 if snr is None else snr                   # ofdm_sym_acquisition's snr parameter (30 in the demo flowgraphs)
        self.ref_power = None
        # ---- pre-scan (rank 0): where does the reference's chain start decoding?  Two numbers, broadcast once.
        plan = torch.zeros(2, dtype=torch.int64, device=self.cdev if dist else "cpu")
        if rank == 0:
            head = po.stream_slice(c, self.nsf, self.seed, 0, po.STREAM_LEAD_IN + 360 * L)
            if cfo:
                head = add_cfo(head, cfo, c.N, 0)
            if snr is not None:
                self.ref_power = float(np.mean(np.abs(head[po.STREAM_LEAD_IN:po.STREAM_LEAD_IN + 100000]) ** 2))
                head = add_awgn(head, snr, 5, self.ref_power)
            if EXPERIMENT_BUILD:                                 # tools/ab_bench.py only (builds with wrong output): where a correct chain starts is known
                plan[0], plan[1] = 0, 204 if (const == po.QAM64 and mode == po.T8k) else 272
            else:
                pre = g.Rx(const, cr, mode, max_samples=len(head), device=local, snr_db=snr_db)
                rep = pre.run(head)
                pre.close()
                if rep.status != 0:
                    raise SystemExit(f"pre-scan failed: status {rep.status}")
                plan[0], plan[1] = int(rep.segment_offset), int(rep.first_call + rep.first_out_symbol)
        if dist:
            dist.broadcast(plan, src=0)
            if snr is not None:
                pw = torch.tensor([self.ref_power or 0.0], dtype=torch.float64, device=self.cdev)
                dist.broadcast(pw, src=0)
                self.ref_power = float(pw.item())
        self.grid0, self.sf_call = int(plan[0]), int(plan[1])
        cuts = multi.plan_cuts(d, self.n_total, self.grid0, self.sf_call, world * nseg)
        if len(cuts) != world * nseg:
            raise SystemExit(f"stream of {self.nsf} superframes cannot be cut into {world * nseg} pieces")
        self.cuts = cuts
        mine = cuts[rank * nseg:(rank + 1) * nseg]
        self.pieces = []
        for i, cu in enumerate(mine):
            iq = po.stream_slice(c, self.nsf, self.seed, cu["begin"], cu["end"])       # only this rank's part of THE stream
            if cfo:
                iq = add_cfo(iq, cfo, c.N, cu["begin"])
            if snr is not None:
                iq = add_awgn(iq, snr, 1000 + rank * nseg + i, self.ref_power)
            kw = {"soft_decision": 1} if soft else {}
            if getattr(a, "viterbi_warm", 0):
                kw["viterbi_warm_windows"] = a.viterbi_warm
            if getattr(a, "viterbi_verify", 0) and not soft:
                kw["viterbi_verify"] = a.viterbi_verify
            if from_file_rate:
                rx_const = 0.0022097087 if mode == po.T2k else 0.00055242272           # blocks_multiply_const_vxx_0 of the RX flowgraph
                iq = po.resample(iq / np.float32(rx_const), 70, 64, 1.0)                # what dvbt_tx_demo writes: the 10 Msps stream
                kw.update({"resample": (64, 70), "front_scale": rx_const})
            d_iq = torch.from_numpy(iq.view(np.float32)).cuda()
            cap = int(len(iq) * 0.45) + 4096
            depth = max(1, getattr(a, "pipeline", 1))            # handles (each with its own HIP stream) that take this piece's steps in turn
            rxs, streams, views = [], [], []
            for _ in range(depth):
                rx = g.Rx(const, cr, mode, max_samples=len(iq), device=local, viterbi_chunk_bytes=chunk, snr_db=snr_db, launch_graph=1 if getattr(a, "graph", False) else 0,
                           front_priority=1 if getattr(a, "front_priority", False) else 0, **kw)
                rx.set_cut(cu["sym_off"])
                rxs.append(rx); streams.append(_stream(torch, i * depth + len(streams)))
                views.append(torch.as_tensor(_DevView(rx.tap_device_ptr(g.TAP_TS), cap), device=f"cuda:{local}"))
            self.pieces.append({"iq": d_iq, "n": len(iq), "rxs": rxs, "streams": streams, "views": views, "rx": rxs[0], "stream": streams[0],
                                "cut": cu, "cap": cap, "ts_view": views[0]})
        # samples of the stream this rank is responsible for (overlaps between pieces are overhead, not throughput)
        self.samples_owned = (cuts[min((rank + 1) * nseg, len(cuts)) - 1]["end"] if rank == world - 1 else cuts[(rank + 1) * nseg]["begin"]) - mine[0]["begin"]
        self.samples_decoded = sum(p["n"] for p in self.pieces)
        torch.cuda.synchronize()                       # uploads done before any piece's stream reads them
        # the single exchange step: packets + their counts -> rank 0 over xGMI, ONE gather per step.  Double
        # buffered and asynchronous: the gather of step k travels while step k+1 decodes; every gather is complete before
        # the closing barrier of the timed region.
        self.slot = (multi.HEADER_BYTES + max(p["cap"] for p in self.pieces) + 63) // 64 * 64
        if dist:
            mx = torch.tensor([self.slot], dtype=torch.int64, device=self.cdev)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            self.slot = (int(mx.item()) + 63) // 64 * 64
        self.send = [torch.zeros(self.slot * nseg, dtype=torch.uint8, device=f"cuda:{local}") for _ in range(2)] if dist else None
        self.recv = [[torch.empty(self.slot * nseg, dtype=torch.uint8, device=self.cdev) for _ in range(world)] for _ in range(2)] if (dist and rank == 0) else [None, None]
        self.send_host = [torch.zeros(self.slot * nseg, dtype=torch.uint8).pin_memory() for _ in range(2)] if self.host_staged else None
        self.pending = [None, None]
        self.nstep = 0
        self.ngather = 0
        self.inflight = []
        self.reps = None
        self.done_times = []
        self.step_ms = None
        self.depth = max(1, getattr(a, "pipeline", 1))

    def step(self):
        """Enqueue this step on the next handle of every piece, then complete the OLDEST step in flight (depth 1: this one).  With depth 2 the
        front end of step k+1 (acquisition, symbol kernel, TPS, de-interleavers) runs on its own stream while the Viterbi decoder of step k
        still holds the machine; drain() completes what is left."""
        depth = len(self.pieces[0]["rxs"])
        k = self.nstep % depth
        for p in self.pieces:
            p["rxs"][k].enqueue_device(p["iq"].data_ptr(), p["n"], p["streams"][k].cuda_stream)
        self.inflight.append(k)
        if len(self.inflight) >= depth:
            self._complete(self.inflight.pop(0))
        self.nstep += 1

    def _complete(self, k):
        torch, multi, dist = self.torch, self.multi, self.dist
        self.reps = [p["rxs"][k].finish() for p in self.pieces]
        self.done_times.append(time.perf_counter())
        for p in self.pieces:
            p["rx"], p["stream"], p["ts_view"] = p["rxs"][k], p["streams"][k], p["views"][k]
        if dist:
            b = self.ngather & 1
            if self.pending[b] is not None:
                self.pending[b].wait()                 # the buffer pair of two steps ago is free again
            for i, (p, r) in enumerate(zip(self.pieces, self.reps)):
                multi.pack_piece(self.send[b][i * self.slot:(i + 1) * self.slot], multi.piece_meta(r), p["ts_view"])
            ev = torch.cuda.Event(); ev.record()       # the next decode may overwrite the TS buffers only after these copies
            for p in self.pieces:
                p["stream"].wait_event(ev)
            if self.host_staged:                       # gloo: the packed pieces leave through a pinned host buffer
                self.send_host[b].copy_(self.send[b])    # (synchronous D2H on the current stream, behind the packing copies)
                self.pending[b] = multi.gather_pieces(self.send_host[b], self.recv[b], dst=0, async_op=True)
            else:
                self.pending[b] = multi.gather_pieces(self.send[b], self.recv[b], dst=0, async_op=True)
            self.last_buf = b
            self.ngather += 1

    def drain(self):
        while self.inflight:
            self._complete(self.inflight.pop(0))
        if self.dist:
            for b in range(2):
                if self.pending[b] is not None:
                    self.pending[b].wait(); self.pending[b] = None
        self.torch.cuda.synchronize()

    def collect(self):
        """rank 0: the pieces of the last step in stream order, [(meta, uint8 device tensor)]"""
        multi = self.multi
        if self.dist:
            out = []
            for buf in self.recv[self.last_buf]:
                for i in range(len(self.pieces)):
                    out.append(multi.unpack_piece(buf[i * self.slot:(i + 1) * self.slot]))
            return out
        return [(multi.piece_meta(r), p["ts_view"][:int(r.n_ts_bytes)]) for p, r in zip(self.pieces, self.reps)]

    def verify(self):
        """rank 0: stitch the last step's pieces and compare with what was transmitted.  Returns a dict for the JSON line."""
        po, multi, c = self.po, self.multi, self.c
        pieces = self.collect()
        ts = multi.stitch_ts(pieces, self.dims).cpu().numpy()
        # the TS tap of the stream starts ts_first_packet RS words after the first superframe start; RS word w of the receiver
        # is the packet sent 11 words earlier (the Forney interleaver pair delays by 11 x 204 bytes end to end)
        ibits = multi.info_bits_per_symbol(self.dims)
        first_out = self.sf_call                                  # symbols of the stream before the superframe start (lead-in is < 1 symbol: call 0 holds symbol 0)
        p0 = first_out * ibits // 8 // 204 + pieces[0][0]["ts_first_packet"] - 11
        npk = len(ts) // 188
        pps = po.packets_per_superframe(c)
        bad_bytes, bad_packets = 0, 0
        for j in range(p0 // pps, (p0 + npk - 1) // pps + 1):    # superframe by superframe: the whole transmitted TS never sits in memory
            sent = po.stream_ts(c, j, 1, self.seed).reshape(-1, 188)
            a, b = max(p0, j * pps), min(p0 + npk, (j + 1) * pps)
            got = ts[(a - p0) * 188:(b - p0) * 188].reshape(-1, 188)
            diff = got != sent[a - j * pps:b - j * pps]
            bad_bytes += int(diff.sum()); bad_packets += int(diff.any(axis=1).sum())
        res = {"verified": bool(bad_bytes == 0 and npk > 0), "ts_bytes": int(len(ts)), "ts_packets": npk, "first_packet_of_stream": int(p0),
               "packets_transmitted_after_first": int(self.nsf * 272 * ibits // 8 // 204 - p0), "wrong_bytes": bad_bytes, "wrong_packets": bad_packets}
        if self.snr is not None:
            res["verified"] = bool(npk > 0)                      # with noise the criterion is the error rate below, not identity
            res["post_rs_byte_error_rate"] = bad_bytes / max(len(ts), 1)
            res["packet_error_rate"] = bad_packets / max(npk, 1)
        return res

    def stage_avg(self, name):
        """HIP-event average over the timed steps (every handle averages its own launches since enable_timing, on the stream they ran on)"""
        v = [rx.stage_ms(name) for p in self.pieces for rx in p["rxs"]]
        v = [x for x in v if x >= 0]
        return sum(v) / max(len(v), 1)

    def solo_stage_ms(self, n=5):
        """stage times with ONE step in flight (no overlap with the next step's front end): the kernels' own durations"""
        for p in self.pieces:
            p["rxs"][0].enable_timing(True)            # a new measurement window on the first handle
        for _ in range(n):
            for p in self.pieces:
                p["rxs"][0].enqueue_device(p["iq"].data_ptr(), p["n"], p["streams"][0].cuda_stream)
            for p in self.pieces:
                p["rxs"][0].finish()
        return {k: round(sum(p["rxs"][0].stage_ms(k) for p in self.pieces) / len(self.pieces), 4) for k in STAGES}

    def close(self):
        for p in self.pieces:
            for rx in p["rxs"]:
                rx.close()


def extra_workloads(a, torch, g, local):
    """Throughput lines for the other single-GPU BASELINE configs (2: 2k QAM16 1/2 clean; 5: 8k QPSK 7/8 + AWGN at 14 dB), same
    machinery, shorter streams; each is verified like the main line (config 5: post-RS error rate against the transmitted packets)."""
    class A:
        segments = 1
        pipeline = a.pipeline
    res = {}
    # last line: the opt-in soft-decision mode (k_soft.hpp, k_soft4.hpp; no reference counterpart, no parity claim) on the headline configuration
    # and the headline configuration on what a capture looks like: a carrier offset of 0.2 subcarrier spacings (the float-accumulator path of k_drift.hpp,
    # the DRIFT instantiation of the symbol kernel) and 25 dB of noise
    for name, wl, snr, nsf, steps, soft, cfo in (("config2_2k_qam16_1_2", "2k_qam16_1_2", None, 64, 400, False, 0.0), ("config5_8k_qpsk_7_8_awgn14", "8k_qpsk_7_8", 14.0, 16, 400, False, 0.0),
                                                 ("soft_decision_8k_qam64_7_8", "8k_qam64_7_8", None, 32, 100, True, 0.0),
                                                 ("8k_qam64_7_8_cfo_0.2_awgn25", "8k_qam64_7_8", 25.0, 32, 100, False, 0.2)):
        job = Job(A, torch, g, None, 0, 1, local, wl, nsf, snr=snr, soft=soft, cfo=cfo)
        dt = timed_run(job, steps, 3)
        chk = job.verify()
        res[name] = {"value": round(job.n_total * steps / dt / 1e6, 2), "unit": "Msamples/s", "x_realtime": round(job.n_total * steps / dt / 1e6 / REALTIME_MSPS, 1),
                     "ms_per_step": round(dt / steps * 1e3, 3), "stream_superframes": job.nsf, "stream_samples": job.n_total, "steps": steps,
                     "steps_in_flight": a.pipeline, "rs_fail_words": [int(r.rs_fail_words) for r in job.reps], "rs_corrected_symbols": [int(r.rs_corrected_symbols) for r in job.reps], **chk}
        job.close()
    return res


def hierarchical_line(torch, g, nsf=4, reps=20):
    """A hierarchical transmission (2k 64-QAM, alpha = 2, rate 2/3; rows A0 / A4 / A6 of the scope table) through the segment API, samples resident: the demapper on the
    shifted grid, the bit de-interleaver's two outputs, and the reference's Viterbi decoder, which knows no priority streams (it unpacks d_m bits of every byte): two thirds of
    its input are constant, and a chunk decoder's default warm-up does not always hold on it (about one chunk start in 250 is not proven: config.viterbi_check.decoded_again).
    Until round 5 these modes therefore ran ONE decoder on one wavefront (1.17x real time); now the chunk decoders run with default parameters and the launch's proof +
    repair passes make them the streaming decoder.  `viterbi_bytes_equal_the_streaming_decoder`: the HIP Viterbi tap against oracle/o_viterbi.c over the chain's own decoder input."""
    import ctypes as C
    from oracle import pyoracle as po
    c = po.cfg(po.QAM64, po.C2_3, po.T2k, hierarchy=g.ALPHA2)
    ibits = c.payload * c.m * c.k // c.n
    iq = po.tx(c, po.make_ts((272 * ibits * nsf) // (204 * 8), 5), lead_in=500, tail=3 * c.N)
    dev = torch.from_numpy(iq.view(np.float32)).cuda()
    torch.cuda.synchronize()
    rx = g.Rx(po.QAM64, po.C2_3, po.T2k, max_samples=len(iq), hierarchy=g.ALPHA2, viterbi_verify=1)
    rx.enable_timing(True)
    rx.enqueue_device(dev.data_ptr(), len(iq)); rep = rx.finish()
    rx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        rx.enqueue_device(dev.data_ptr(), len(iq)); rep = rx.finish()
    dt = (time.perf_counter() - t0) / reps
    stages = {k: round(rx.stage_ms(k), 3) for k in STAGES}
    vit = rx.tap(g.TAP_VITERBI).copy()
    bd = np.ascontiguousarray(rx.tap(g.TAP_BITDEINT).reshape(-1))
    proof = viterbi_check_of(rx)
    rx.close()
    ref = np.zeros(bd.size * c.m * c.k // (8 * c.n) + 64, np.uint8)
    po.lib().o_viterbi_decode.restype = C.c_size_t
    n = po.lib().o_viterbi_decode(C.byref(c), 768, bd.ctypes.data_as(C.c_void_p), C.c_size_t(bd.size), ref.ctypes.data_as(C.c_void_p))
    return {"workload": "2k QAM64 alpha 2 rate 2/3, %d superframes" % nsf, "value": round(len(iq) / dt / 1e6, 2), "unit": "Msamples/s",
            "x_realtime": round(len(iq) / dt / 1e6 / REALTIME_MSPS, 2), "ms_per_run": round(dt * 1e3, 3), "stage_ms": stages, "samples": int(len(iq)),
            "viterbi_bytes": int(rep.n_viterbi_bytes), "symbols": int(rep.n_symbols), "viterbi_check": proof,
            "viterbi_bytes_equal_the_streaming_decoder": bool(n == len(vit) > 0 and (vit == ref[:n]).all())}


def config5_noise(torch, g, nsf=16, snrs=(9.0, 8.0), reps=5):
    """BASELINE config 5 at the noise level SURVEY 8d prescribes (pre-Viterbi BER ~ 1e-2 = 8 dB for 8k QPSK 7/8; 9 dB = where the RS decoder still
    holds): the reference's CP tracker loses the lock again and again there, so the stream goes through dvbt_rx_segment_run_device (synchronous: every
    lock period is followed inside the library, with a host round trip per period), samples resident in HBM.  ofdm_sym_acquisition's snr parameter = the
    channel's.  Packet error rate against the transmitted packets; the HIP-vs-oracle comparison of the same sweep is tests/test_gpu_config5.py."""
    from oracle import pyoracle as po
    (const, cr, mode), c = workload_cfg("8k_qpsk_7_8")
    clean = po.stream_slice(c, nsf, SEED)
    sent = {bytes(p) for p in po.stream_ts(c, 0, nsf, SEED).reshape(-1, 188)}
    out = {}
    for snr in snrs:
        iq = po.channel(clean, c.N, snr_db=snr, seed=5)
        dev = torch.from_numpy(iq.view(np.float32)).cuda()
        torch.cuda.synchronize()
        rx = g.Rx(const, cr, mode, max_samples=len(iq), snr_db=snr)
        rep = rx.run_device(dev.data_ptr(), len(iq))
        t0 = time.perf_counter()
        for _ in range(reps):
            rep = rx.run_device(dev.data_ptr(), len(iq))
        dt = (time.perf_counter() - t0) / reps
        ts = rx.tap(g.TAP_TS).reshape(-1, 188)
        good = sum(1 for p in ts if bytes(p) in sent)
        rx.close()
        out[f"awgn_{snr:g}_dB"] = {"value": round(len(iq) / dt / 1e6, 1), "unit": "Msamples/s", "x_realtime": round(len(iq) / dt / 1e6 / REALTIME_MSPS, 1),
                                   "ms_per_run": round(dt * 1e3, 3), "stream_superframes": nsf, "stream_samples": int(len(iq)), "entry": "dvbt_rx_segment_run_device",
                                   "lock_periods": int(rep.n_lock_periods), "symbols_acquired": int(rep.total_symbols), "rs_fail_words": int(rep.rs_fail_words),
                                   "rs_corrected_symbols": int(rep.rs_corrected_symbols), "ts_packets": int(len(ts)), "packet_error_rate": round(1.0 - good / max(len(ts), 1), 5)}
    return out


def stream_abi(g):
    """The streaming entry of the C ABI (dvbt_rx_stream_push / pull: what the one-block GNU Radio shell gr::dvbt::rx_hip calls) at scheduler-like call sizes"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stream_bench
    from oracle import pyoracle as po
    rows = [stream_bench.run(po, g, symbols=sym, device=dev) for sym, dev in ((4, False), (64, False), (64, True))]
    rows.append(stream_bench.run(po, g, symbols=64, device=True, borrow=1, verify=True))   # the same from device memory with the samples lent to the stream (decoded in place)
    # BASELINE config 5 at the prescribed noise through the streaming entry: the reference's tracker drops the lock every few dozen symbols, no lock period is ever
    # established, the stream is walked window by window (csrc/dvbt_stream.inc); the TS is compared with the single chain's (dvbt_rx_segment_run on all samples)
    rows.append(stream_bench.run(po, g, workload=("QPSK", "C7_8", "T8k"), nsf=16, seg_sf=4, symbols=64, awgn_db=9.0, verify=True, reps=1))
    return rows


def host_pointer_ceiling(g, c):
    """Why the ten blocks behind HOST pointers cannot reach the segment API's rate whatever the kernels do: gr::block's contract puts every intermediate item
    through host memory, i.e. across PCIe twice per block.  Bytes per OFDM symbol in and out of the ten blocks (item sizes of the flowgraph), the pageable
    copy rates of this box measured with the library's own hipMemcpy path at a 64-symbol call's size, and the rate at which copies alone would run
    (no kernel, no launch, no Python: an upper bound for dvbt_<blk>_work driven by one thread, as here; GNU Radio runs a thread per block)."""
    import ctypes as C
    N, cp, P = c.N, c.cp, c.payload
    vit = P * c.m * c.k // (8 * c.n)
    words = vit // 204
    h2d = (N + cp) * 8 + N * 8 + N * 8 + P * 8 + P + P + P + vit + words * 204 + words * 188       # acq, fft, demod, demap, symdeint, bitdeint, viterbi, deint, rs, descramble
    d2h = N * 8 + N * 8 + P * 8 + P + P + P + vit + words * 204 + words * 188 + words * 188
    L = g.lib()
    L.dvbt_device_malloc.restype = C.c_void_p; L.dvbt_device_malloc.argtypes = [C.c_size_t]
    L.dvbt_device_free.argtypes = [C.c_void_p]
    L.dvbt_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]; L.dvbt_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    nb = 64 * N * 8
    host = np.ones(nb, np.uint8)
    dev = L.dvbt_device_malloc(nb)
    rates = []
    for fn, a, b in ((L.dvbt_copy_to_device, dev, host.ctypes.data), (L.dvbt_copy_to_host, host.ctypes.data, dev)):
        fn(a, b, nb)
        t0 = time.perf_counter()
        for _ in range(20):
            fn(a, b, nb)
        rates.append(20 * nb / (time.perf_counter() - t0) / 1e9)
    L.dvbt_device_free(dev)
    per_sym_s = h2d / (rates[0] * 1e9) + d2h / (rates[1] * 1e9)
    # the regime dvbt_host_register creates: page-locked buffers, both DMA directions busy at once (a thread per block: one block's download runs beside another's
    # upload).  The bound is then the slower direction's time, not the sum
    import torch
    hp_up, hp_dn = torch.ones(nb, dtype=torch.uint8).pin_memory(), torch.empty(nb, dtype=torch.uint8).pin_memory()
    d_up, d_dn = torch.empty(nb, dtype=torch.uint8, device="cuda"), torch.ones(nb, dtype=torch.uint8, device="cuda")
    s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    reps = 40
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(2)]
    with torch.cuda.stream(s_up):
        ev[0][0].record()
        for _ in range(reps):
            d_up.copy_(hp_up, non_blocking=True)
        ev[0][1].record()
    with torch.cuda.stream(s_dn):
        ev[1][0].record()
        for _ in range(reps):
            hp_dn.copy_(d_dn, non_blocking=True)
        ev[1][1].record()
    torch.cuda.synchronize()
    both = [reps * nb / (ev[k][0].elapsed_time(ev[k][1]) * 1e-3) / 1e9 for k in range(2)]
    per_sym_reg = max(h2d / (both[0] * 1e9), d2h / (both[1] * 1e9))
    return {"bytes_per_symbol_host_to_device": int(h2d), "bytes_per_symbol_device_to_host": int(d2h), "bytes_per_sample_over_pcie": round((h2d + d2h) / (N + cp), 1),
            "pageable_copy_gbs": {"host_to_device": round(rates[0], 1), "device_to_host": round(rates[1], 1), "bytes_per_copy": nb},
            "copies_alone_msamples_per_s": round((N + cp) / per_sym_s / 1e6, 1),
            "page_locked_copy_gbs_both_directions_busy": {"host_to_device": round(both[0], 1), "device_to_host": round(both[1], 1), "bytes_per_copy": nb},
            "page_locked_copies_alone_msamples_per_s": round((N + cp) / per_sym_reg / 1e6, 1),
            "note": "with registered buffers the copies bound the ten blocks at page_locked_copies_alone_msamples_per_s, far above what is measured: what bounds a thread-per-block "
                    "run is the busiest block's time inside work() (cpp_driver.*.busiest_block_bound): its ~25 dependent small launches and its one synchronisation per call"}


def per_block_abi(g, workload, nsf=4):
    """The drop-in path timed: config `workload` pushed through the ten per-block ABI calls (gr_dvbt_amd/flowgraph.py) at GNU Radio-like
    call sizes, host-pointer entry (dvbt_<blk>_work: H2D + kernels + D2H + synchronise per block, what a gr::block shell does) and
    device-pointer entry (dvbt_<blk>_work_device: the blocks hand items over in HBM).  A bounded sample; the first pass of every variant
    warms up, the second is timed.  The Python driver's own per-call cost (ctypes, ~10 blocks x calls) is inside these numbers."""
    from oracle import pyoracle as po
    from gr_dvbt_amd.flowgraph import RxFlowgraph
    (const, cr, mode), c = workload_cfg(workload)
    iq = po.stream_slice(c, nsf + 1, 77)
    seg = g.Rx(const, cr, mode, max_samples=len(iq))
    seg.run(iq)
    want = seg.tap(g.TAP_TS)
    seg.close()
    out = {"sample": f"{nsf + 1} superframes of {workload} ({len(iq)} samples)", "unit": "Msamples/s", "variants": {}}
    out["host_pointer_ceiling"] = host_pointer_ceiling(g, c)
    # "threads": a thread per block, as under GNU Radio's scheduler (the ten blocks' calls overlap: throughput = the slowest block's);
    # the others: one thread calls the blocks in turn (throughput = the sum of all calls)
    for mode_name, cs, threaded in (("host", 4, False), ("host", 64, False), ("host", 64, True), ("host", 256, True), ("device", 4, False), ("device", 64, False)):
        best = None
        for rep in range(2):
            fg = RxFlowgraph(const, cr, mode, len(iq), mode=mode_name, call_symbols=cs)
            t0 = time.perf_counter()
            ts = fg.run_threaded(iq) if threaded else fg.run(iq)
            dt = time.perf_counter() - t0
            calls = sum(st.calls for st in fg.stages)
            fg.close()
            best = dt if best is None or dt < best else best
        n = min(len(ts), len(want))
        out["variants"][f"{mode_name}_pointers_{cs}_symbols_per_call" + ("_thread_per_block" if threaded else "")] = {
            "value": round(len(iq) / best / 1e6, 2), "x_realtime": round(len(iq) / best / 1e6 / REALTIME_MSPS, 1), "block_calls": calls,
            "ts_identical_to_segment_api": bool(n > 0 and (ts[:n] == want[:n]).all()), "ts_bytes": int(len(ts))}
    out["cpp_driver"] = per_block_cpp(po.stream_slice(c, 17, 77) if workload == "8k_qam64_7_8" else iq, workload)
    return out


def per_block_cpp(iq, workload):
    """the same ten blocks driven from C++ (gr_dvbt_amd/host/rx_blocks_bench.cpp: host pointers, 64 symbols per call): one thread calling them in turn, and a thread per
    block as under GNU Radio's scheduler -- no interpreter between the calls.  17 superframes (a handle's buffers are allocated in its first calls: a 5-superframe run
    is a third start-up); the baseband goes through a file in /dev/shm (read before the clock starts)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "gr_dvbt_amd", "host", "rx_blocks_bench")
    if not os.path.exists(exe) or workload != "8k_qam64_7_8":
        return None
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    fin, fout = os.path.join(tmp, "bb.cf32"), os.path.join(tmp, "out.ts")
    res = {}
    try:
        iq.tofile(fin)
        for cs, thr, reg in ((64, 0, 0), (64, 1, 0), (64, 0, 1), (64, 1, 1), (256, 1, 1)):
            best = None
            for _ in range(2):
                r = subprocess.run([exe, "8k", "qam64", "7/8", fin, fout, str(cs), str(thr), str(reg)], capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    return {"error": r.stderr[-300:]}
                d = json.loads(r.stdout.strip().splitlines()[-1])
                best = d if best is None or d["msamples_per_s"] > best["msamples_per_s"] else best
            res[f"host_pointers_{cs}_symbols_per_call" + ("_thread_per_block" if thr else "") + ("_registered_buffers" if reg else "")] = {"value": best["msamples_per_s"], "x_realtime": round(best["msamples_per_s"] / REALTIME_MSPS, 1),
                                                                                              "block_calls": best["block_calls"], "ts_bytes": best["ts_bytes"],
                                                                                              "busiest_block": best.get("busiest_block"), "busiest_block_bound": best.get("busiest_block_bound_msamples_per_s"),
                                                                                              "ms_per_call_of_the_busiest_block": best.get("ms_per_call_busiest")}
    finally:
        for f in (fin, fout):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(tmp)
    return res


def cpp_multi_host(workload, loops=160, seg_sf=64):
    """BASELINE config 4's host in C++ (gr_dvbt_amd/host/rx_multi_example.cpp, one rank: one GPU per box) on the bench line's workload: the baseband resident in
    device memory (uploaded before the clock starts), pushed through dvbt_rx_stream_push_device in calls of eight superframes, pieces of 64 superframes, ONE
    asynchronous double-buffered RCCL step per piece's worth of pushes (dvbt_rx_stream_gather_enqueue_ex with DVBT_GATHER_DEVICE / _wait: the runs stay in rank 0's
    device memory -- north_star's clock ends at "last TS byte resident on rank 0", as the Python line's does; `with_page_locked_mirror` is the same host with the root's
    download of exactly the bytes the headers declare, dvbt_rx_stream_gather_enqueue).  The stretch of 64 superframes behind the first one is pushed `loops` times (a seamless
    stream but for the encoder's memory at the seam); a warm-up stream runs first.  The samples are LENT to the stream (dvbt_rx_stream_params.borrow_device_pushes: the
    stretch lies four times back to back in device memory and the pushes walk through that ring, so three pieces of four are decoded where they lie and the
    fourth, across the ring's wrap, is gathered); `every_push_copies` is the same host under the default push contract (the samples are COPIED into the
    library: 1.0 ms of blit kernels per piece, rocprofv3)."""
    import subprocess
    import tempfile
    from oracle import pyoracle as po
    exe = os.path.join(ROOT, "gr_dvbt_amd", "host", "rx_multi_example")
    if not os.path.exists(exe) or workload != "8k_qam64_7_8":
        return None
    c = po.cfg(po.QAM64, po.C7_8, po.T8k)
    sf = 272 * (c.N + c.cp)
    iq = po.stream_slice(c, 66, 77)
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    fin, idf = os.path.join(tmp, "bb.cf32"), os.path.join(tmp, "nccl.id")
    try:
        iq.tofile(fin)
        res = {}
        for mode in ("lent", "two_chains", "mirror", "copy"):
            best = None
            for _ in range(2):
                if os.path.exists(idf):
                    os.remove(idf)
                r = subprocess.run([exe, "0", "1", idf, "8k", "qam64", "7/8", fin, os.path.join(tmp, "none.ts"), str(seg_sf), "0", "bench", str(loops),
                                    str(po.STREAM_LEAD_IN + sf), str(64 * sf), str(8 * sf), "8", "400000"] + (["copy"] if mode == "copy" else ["mirror"] if mode == "mirror" else []) + ([] if mode == "two_chains" else ["chains=4"]),
                                   capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
                if r.returncode != 0:
                    return {"error": (r.stdout[-200:] + r.stderr[-300:])}
                d = json.loads(r.stdout.strip().splitlines()[-1])
                best = d if best is None or d["msamples_per_s"] > best["msamples_per_s"] else best
            res[mode] = best
        best = res["lent"]
        return {"value": best["msamples_per_s"], "unit": "Msamples/s", "x_realtime": round(best["msamples_per_s"] / REALTIME_MSPS, 1), "world": 1, "samples": best["samples"],
                "seconds": best["seconds"], "exchange_steps": best["exchange_steps"], "ts_bytes": best["ts_bytes"], "status": best["status"],
                "entry": "dvbt_rx_stream_push_device (samples lent: borrow_device_pushes) + dvbt_rx_stream_gather_enqueue_ex(DVBT_GATHER_DEVICE) / _wait (RCCL, one rank; the runs stay in rank 0's device memory)",
                "segment_superframes": seg_sf, "superframes_per_push": 8, "pushes_per_exchange_step": 8,
                "chains": 4, "chains_note": "dvbt_rx_stream_params.chains = 4: three pieces decode while the fourth fills (the Python line's four steps in flight); with_two_chains = the default stream object",
                "with_two_chains": {"value": res["two_chains"]["msamples_per_s"], "seconds": res["two_chains"]["seconds"], "status": res["two_chains"]["status"]},
                "with_page_locked_mirror": {"value": res["mirror"]["msamples_per_s"], "seconds": res["mirror"]["seconds"], "status": res["mirror"]["status"]},
                "every_push_copies": {"value": res["copy"]["msamples_per_s"], "seconds": res["copy"]["seconds"], "status": res["copy"]["status"]}}
    finally:
        for f in (fin, idf):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(tmp)


def timed_run(job, steps, warmup):
    torch, dist = job.torch, job.dist
    for _ in range(warmup):
        job.step()
    job.drain()
    if not getattr(job.args, "graph", False):
        for p in job.pieces:
            for rx in p["rxs"]:
                rx.enable_timing(2)                 # HIP events around the dominant kernel only (the in-flight launch interval): every further event record
                                                    # holds an idle stream for ~6 us, which shows in a step that runs alone (--pipeline 1).  With the step replayed as a
                                                    # HIP graph (the default) the timed region carries no events at all: the kernel's own duration (roofline.solo_launch_ms)
                                                    # is measured with events right behind it, launch by launch
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    job.drain()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # dispersion: host-side intervals between the completions of consecutive steps (job._complete returns when a step's report is back)
    done = job.done_times[-steps:]
    iv = np.diff(np.array([t0] + done)) * 1e3 if len(done) == steps else np.zeros(0)
    if len(iv) > job.depth:                                   # the first `depth` completions include the pipeline's fill
        iv = iv[job.depth:]
    job.step_ms = {"min": round(float(iv.min()), 3), "median": round(float(np.median(iv)), 3), "max": round(float(iv.max()), 3),
                   "p95": round(float(np.percentile(iv, 95)), 3), "n": int(len(iv))} if len(iv) else None
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=job.cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt


def launch_ranks(a, argv):
    """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start the N ranks here, one process per GPU, each with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in its environment (what `python -m torch.distributed.run --nproc-per-node N` sets), and wait for them.
    Rank 0 inherits this process's stdout (its ONE JSON line is the bench line); the other ranks' stdout goes to stderr.  Refuses when fewer than N
    devices are visible (unless --ranks-share-gpu, the one-GPU test box's mode)."""
    import socket
    import subprocess
    import gr_dvbt_amd as g
    ndev = g.device_count()
    if ndev <= 0:
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    if ndev < a.gpus and not a.ranks_share_gpu:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} HIP device(s) visible; one process per GPU needs {a.gpus} "
                         "(--ranks-share-gpu puts every rank on device 0: a functional test, not a measurement)")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                try:
                    p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.remove(p)
                if p.returncode != 0:
                    rc = rc or p.returncode
                    for q in pending:                      # a dead rank leaves the others in a collective: stop exactly the processes started here
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="8k_qam64_7_8", choices=sorted(WORKLOADS))
    ap.add_argument("--snr", type=float, default=None, help="add AWGN at this SNR (dB); BASELINE config 5 = --workload 8k_qpsk_7_8 --snr 14")
    ap.add_argument("--superframes", type=int, default=64, help="payload superframes per GPU per step (SURVEY 8d: >= 64 for throughput runs)")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--segments", type=int, default=1, help="pieces per GPU: the rank's part of the stream is cut again, one handle + HIP stream per piece")
    ap.add_argument("--graph", action="store_true", help="replay every step as ONE HIP graph launch (dvbt_rx_params.launch_graph) instead of enqueueing it launch by launch; measured: no "
                    "difference (4.673 vs 4.670 ms with one step in flight, 4.379 vs 4.369 with three: the step is a chain of DEPENDENT kernels, not of launch calls), so it is not the default")
    ap.add_argument("--front-priority", action="store_true", help="dvbt_rx_params.front_priority: a step's front end on a high-priority stream of the handle's own")
    ap.add_argument("--viterbi-warm", type=int, default=0, help="dvbt_rx_params.viterbi_warm_windows: warm-up of the Viterbi stage's chunk decoders in windows (0 = the default 72; a multiple of 24): "
                    "what equality with the streaming decoder on collapsed channels costs (DESIGN.md 2)")
    ap.add_argument("--viterbi-verify", type=int, default=1, help="dvbt_rx_params.viterbi_verify: 1 (the bench's default) = the library's default path -- every launch of the decoder proves chunk by chunk "
                    "that it is the streaming decoder and decodes the unproven chunks again -- plus the final check whose count is config.viterbi_check.not_proven_after_repair; 0 = the same "
                    "without the final check's two small launches (what a handle gets by default); -1 = the plain chunk decoders (no proof: the round-5 default, for A/B)")
    ap.add_argument("--pipeline", type=int, default=4, help="steps in flight per piece: handles (own HIP stream each) that take the piece's steps in turn (round 6 on the final code: 4.52 / 4.30 / 4.42 / 4.39 ms per "
                    "step with 3 / 4 / 5 / 6 in flight, profiles/r06_pipeline_depth.jsonl: four is the default)")
    ap.add_argument("--from-file-rate", action="store_true",
                    help="feed the 10 Msps file format: rational_resampler 64/70 + multiply_const run on the device in front of the chain "
                         "(SURVEY 8f row 2; samples are then counted at the 10 Msps input; single piece only)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) and run the gather path even with one rank")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend of the exchange step: nccl = RCCL over xGMI on device buffers (the design); gloo = the same gather staged "
                         "through host tensors (functional tests where the ranks share one GPU, which RCCL refuses)")
    ap.add_argument("--ranks-share-gpu", action="store_true", help="every rank decodes on device 0 (tests on a one-GPU box; needs --backend gloo when --gpus > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra workload lines (BASELINE configs 1, 2, 5) and the per-block ABI timing")
    ap.add_argument("--cpu-superframes", type=int, default=23)
    a = ap.parse_args()
    if a.from_file_rate and (a.gpus > 1 or a.segments > 1):
        raise SystemExit("--from-file-rate runs one piece on one GPU")
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a, sys.argv[1:])                        # does not return
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was launched with WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's rank count and --gpus must agree")
    if a.ranks_share_gpu and a.gpus > 1 and a.backend == "nccl":
        raise SystemExit("--ranks-share-gpu with --gpus > 1 needs --backend gloo (RCCL refuses two ranks on one device)")

    # stdout carries exactly ONE line (the JSON of rank 0): libraries that print banners to the C-level stdout (RCCL does at the
    # first collective) are sent to stderr by swapping file descriptor 1; the JSON goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import gr_dvbt_amd as g

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.ranks_share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    job = Job(a, torch, g, dist, rank, world, local, a.workload, a.superframes, snr=a.snr, chunk=a.chunk, from_file_rate=a.from_file_rate)
    dt = timed_run(job, a.steps, a.warmup)
    nseg = len(job.pieces)
    ok = True
    if rank == 0:
        n_stream = job.n_total if not a.from_file_rate else sum(p["n"] for p in job.pieces)
        msps = n_stream * a.steps / dt / 1e6
        check = job.verify() if not (a.from_file_rate or EXPERIMENT_BUILD) else {"verified": None}
        ok = check["verified"] is not False
        reps = job.reps
        # dominant kernel = viterbi3_kernel: per launch, algorithmic bytes = bytes in (one per m coded bits) + decoded
        # bytes out (6048 + 3969 B per 8k QAM64 7/8 OFDM symbol); launch duration from HIP events
        # recorded on the piece's own stream; averages over this rank's pieces
        d = job.dims
        depth = len(job.pieces[0]["rxs"])
        vit_ms = job.stage_avg("viterbi")                         # average over the timed steps' launches
        stage_avg = {"viterbi": round(vit_ms, 4)} if vit_ms > 0 else None   # (the timed region carries events around the decoder only; none with --graph)
        solo = job.solo_stage_ms()                                # every stage, one step in flight, right after the timed region
        alg_bytes = sum(r.n_out_symbols * d.payload_length + r.n_viterbi_bytes for r in reps) / nseg
        solo_ms = solo["viterbi"] if (depth > 1 or vit_ms <= 0) else vit_ms       # the kernel's own duration: one step in flight
        achieved = alg_bytes / (solo_ms * 1e-3) / 1e9 if solo_ms > 0 else 0.0
        traffic, traffic_src = profiled_traffic(a.workload, job.nsf, int(alg_bytes))
        n_ts = check.get("ts_bytes") or sum(int(r.n_ts_bytes) for r in reps)
        out = {
            "metric": "RX Msamples/s (baseband in -> TS out)", "value": round(msps, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 (Viterbi/RS) + f32 (front end)",
            "data": "synthetic", "x_realtime": round(msps / REALTIME_MSPS / world, 1), "timed_region_s": round(dt, 3),
            "config": {"workload": f"{a.workload} GI 1/32 RX chain, " + (f"AWGN {a.snr} dB" if a.snr is not None else "clean TX->RX loopback")
                                   + (", input at the 10 Msps file rate (resampler 64/70 + scale on the device)" if a.from_file_rate else ""),
                       "stream_superframes": job.nsf, "superframes_per_gpu": a.superframes, "pieces_per_gpu": nseg, "steps_in_flight": depth,
                       "viterbi_warm_windows": a.viterbi_warm or 72,
                       "viterbi_verify": a.viterbi_verify,
                       **({"viterbi_check": [viterbi_check_of(p["rx"]) for p in job.pieces]} if a.viterbi_verify >= 0 else {}),
                       "stream_samples": n_stream, "samples_decoded_per_gpu_per_step": job.samples_decoded,
                       "parallelism": f"one stream cut into {world * nseg} pieces at superframe boundaries, {nseg} per GPU"
                                      + ((" + one RCCL gather of TS per step" if a.backend == "nccl" else " + one gloo gather of TS per step (host staged)") if dist else ""),
                       "ranks": int(dist.get_world_size()) if dist else 1, "backend": (dist.get_backend() if dist else None),
                       "devices": ("all ranks on device 0 (--ranks-share-gpu: functional run, not a measurement)" if a.ranks_share_gpu and world > 1 else "one process per GPU"),
                       "ts_bytes_per_step": n_ts, "status": [int(r.status) for r in reps],
                       "status_note": "dvbt_rx_report.status of each piece; bit 1 (value 2) = the CP lock ended where the finite synthetic stream's signal ends (the zeros behind its last symbol), not a loss inside it",
                       "rs_fail_words": [int(r.rs_fail_words) for r in reps],
                       **check},
            "roofline": {"bound": "valu", "kernel": "viterbi3_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "note": "the dominant kernel is bound by VALU issue, not by HBM (DESIGN.md 5: every wave64 instruction of its mix occupies the SIMD for 4 cycles); "
                                 "achieved / peak / frac are the HBM figures the contract asks for: algorithmic bytes per launch / solo_launch_ms (HIP events on the "
                                 "launch's own stream with ONE step in flight, measured right after the timed region; agrees with the rocprofv3 kernel-trace average "
                                 "in profiles/); in_flight_* = the same over the timed steps, where the launch interval also contains the time the stream waited while "
                                 "config.steps_in_flight steps shared the machine; traffic = PMC bytes per launch (FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE), "
                                 "taken by separate rocprofv3 --pmc passes of this command: " + str(traffic_src),
                         "algorithmic_bytes_per_launch": int(alg_bytes), "solo_launch_ms": round(solo_ms, 4),
                         "in_flight_launch_ms": round(vit_ms, 4) if vit_ms > 0 else None, "in_flight_achieved": round(alg_bytes / (vit_ms * 1e-3) / 1e9, 2) if vit_ms > 0 else None,
                         "chain_frac": round(msps / world * 1e6 * (8 + n_ts / n_stream) / 1e9 / HBM_PEAK_GBS, 6),
                         "hbm_copy_gbs": hbm_copy_gbs(torch, f"cuda:{local}"),
                         "second_kernel": second_kernel(a.workload, job.nsf, int(alg_bytes), reps, d, job.c, solo, nseg)},
            # what actually bounds the dominant kernel (top level: the driver's parser keeps `roofline`'s contract keys only)
            "valu_issue": valu_issue(a.workload, job.nsf, int(alg_bytes), solo_ms, device_clock_hz(torch, local)),
            "ms_per_step_dispersion": job.step_ms,
            "stage_ms_per_piece": stage_avg,
            "stage_ms_per_piece_solo": solo,
        }
    job.close()
    if rank == 0 and world == 1 and not a.no_extras and not a.from_file_rate:
        # the lines beside the headline: one that fails is recorded as such and does not take the bench line with it (the headline above is measured and verified by then)
        def extra(fn, *args):
            try:
                return fn(*args)
            except Exception as e:                              # noqa: BLE001
                import traceback
                sys.stderr.write(f"bench.py: extra line {fn.__name__} failed:\n{traceback.format_exc()}\n")
                return {"error": f"{type(e).__name__}: {e}"[:300]}
        out["extra_workloads"] = extra(extra_workloads, a, torch, g, local)
        out["extra_workloads"]["config5_8k_qpsk_7_8_at_the_prescribed_noise"] = extra(config5_noise, torch, g)
        out["extra_workloads"]["hierarchical_2k_qam64_alpha2_2_3"] = extra(hierarchical_line, torch, g)
        out["stream_abi"] = extra(stream_abi, g)
        out["per_block_abi"] = extra(per_block_abi, g, a.workload)
        out["cpp_multi_host"] = extra(cpp_multi_host, a.workload)
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:             # the contract: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_superframes)
            ref = reference_sse2_viterbi()
            if ref:
                out["cpu_baseline"]["reference_sse2_viterbi"] = ref
            if not a.no_extras and a.workload != "2k_qam16_1_2":
                # BASELINE config 1 (apps/dvbt_rx_demo.grc on CPU: 2k QAM16 1/2): the same port, one core
                out["cpu_baseline"]["config1_2k_qam16_1_2"] = cpu_baseline("2k_qam16_1_2", 8, all_cores=False)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
