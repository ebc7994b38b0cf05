#!/usr/bin/env python3
"""bench.py -- RX Msamples/s of the DVB-T receive hot path on MI355X (BASELINE.json metric).

One step = one pass of the whole chain (ofdm_sym_acquisition -> FFT -> demod_reference_signals ->
dvbt_demap -> symbol/bit de-interleave -> viterbi_decoder -> convolutional_deinterleaver ->
reed_solomon_dec -> energy_descramble) over one batch of synthetic loopback baseband that is already
resident in HBM.  N>1: one process per GPU (torch.distributed / RCCL), every rank decodes its own
independent segment (weak scaling) and the decoded TS bytes are gathered on rank 0 once per step.

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
# independent segments run on separate HIP streams; give the runtime enough hardware queues that two
# streams of one process do not share one (the ROCm default maps them onto 4 queues round-robin)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
REALTIME_MSPS = 64.0 / 7.0       # OFDM elementary rate at the input of ofdm_sym_acquisition


class _DevView:
    """Zero-copy torch view of a library-owned device buffer."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def make_input(cfg_name, n_superframes, seed):
    """Synthetic loopback baseband from the test-utility TX generator: seeded random TS,
    one extra lead-in superframe so the superframe hunt locks before the payload region."""
    from oracle import pyoracle as po
    const, cr, mode = {"8k_qam64_7_8": (po.QAM64, po.C7_8, po.T8k), "2k_qam16_1_2": (po.QAM16, po.C1_2, po.T2k),
                       "8k_qpsk_7_8": (po.QPSK, po.C7_8, po.T8k)}[cfg_name]
    c = po.cfg(const, cr, mode)
    ibits = c.payload * c.m * c.k // c.n
    npk = (272 * ibits * (n_superframes + 1)) // (204 * 8)
    ts = po.make_ts(npk, seed)
    iq = po.tx(c, ts, lead_in=1000, tail=3 * c.N)
    return (const, cr, mode), c, iq


def cpu_baseline(cfg_name, n_superframes=3):
    """The oracle port (oracle/o_chain.c, -O3 -funroll-loops -msse2) timed on a bounded sample of the same workload on this
    box's host cores: (i) one thread end to end = the reported value; (ii) what a thread-per-block scheduler
    like GNU Radio's could reach = sample / slowest stage; (iii) segment-parallel on all cores (every thread decodes its
    own copy of the sample; ctypes releases the GIL)."""
    import threading
    from oracle import pyoracle as po
    (_, _, _), c, iq = make_input(cfg_name, n_superframes, 99)
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
    if ncores > 1:
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="8k_qam64_7_8")
    ap.add_argument("--superframes", type=int, default=64, help="payload superframes per GPU per step (SURVEY 8d: >= 64 for throughput runs)")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--segments", type=int, default=1,
                    help="independent baseband segments per GPU per step, each on its own HIP stream (each gets its own lead-in superframe)")
    ap.add_argument("--from-file-rate", action="store_true",
                    help="feed the 10 Msps file format: rational_resampler 64/70 + multiply_const run on the device in front of the chain "
                         "(SURVEY 8f row 2; samples are then counted at the 10 Msps input)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-superframes", type=int, default=3)
    a = ap.parse_args()

    # stdout carries exactly ONE line (the JSON of rank 0): libraries that print banners to the C-level stdout (RCCL does at the
    # first collective) are sent to stderr by swapping file descriptor 1; the JSON goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import gr_dvbt_amd as g

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    # the per-GPU batch is cut into `segments` independent segments (whole superframes + one lead-in each); they are
    # enqueued on separate streams so that the short sequential kernels of one overlap the Viterbi kernel of another
    nseg = max(1, a.segments)
    per = [a.superframes // nseg + (1 if i < a.superframes % nseg else 0) for i in range(nseg)]
    segs = []
    for i, nsf in enumerate(per):
        (const, cr, mode), c, iq = make_input(a.workload, nsf, 20240607 + 100 * rank + i)
        rs_kw = {}
        if a.from_file_rate:
            from oracle import pyoracle as po
            rx_const = 0.0022097087 if mode == po.T2k else 0.00055242272       # blocks_multiply_const_vxx_0 of the RX flowgraph
            iq = po.resample(iq / np.float32(rx_const), 70, 64, 1.0)            # what dvbt_tx_demo writes: the 10 Msps stream
            rs_kw = {"resample": (64, 70), "front_scale": rx_const}
        d_iq = torch.from_numpy(iq.view(np.float32)).cuda()
        rxi = g.Rx(const, cr, mode, max_samples=len(iq), device=local, viterbi_chunk_bytes=a.chunk, **rs_kw)
        segs.append({"iq": d_iq, "n": len(iq), "rx": rxi, "stream": torch.cuda.Stream()})
    nsamp = sum(sg["n"] for sg in segs)
    torch.cuda.synchronize()                       # uploads done before any segment stream reads them
    rx = segs[0]["rx"]
    ts_cap = int(nsamp * 0.45) + 4096
    ts_views = [torch.as_tensor(_DevView(sg["rx"].tap_device_ptr(g.TAP_TS), int(sg["n"] * 0.45) + 4096), device=f"cuda:{local}") for sg in segs]
    # the single exchange step: TS packets -> rank 0 over xGMI.  Double buffered and asynchronous: the gather of
    # step k travels while step k+1 decodes; every gather is complete before the closing barrier of the timed region.
    ts_send = [torch.zeros(ts_cap, dtype=torch.uint8, device=f"cuda:{local}") for _ in range(2)] if dist else None
    gathered = [[torch.empty(ts_cap, dtype=torch.uint8, device=f"cuda:{local}") for _ in range(world)] for _ in range(2)] if (dist and rank == 0) else [None, None]
    pending = [None, None]
    nstep = [0]

    def step():
        for sg in segs:
            sg["rx"].enqueue_device(sg["iq"].data_ptr(), sg["n"], sg["stream"].cuda_stream)
        reps = [sg["rx"].finish() for sg in segs]
        if dist:
            b = nstep[0] & 1
            if pending[b] is not None:
                pending[b].wait()                  # the buffer pair of two steps ago is free again
            off = 0
            for v, r in zip(ts_views, reps):
                n = int(r.n_ts_bytes)
                ts_send[b][off:off + n].copy_(v[:n])
                off += n
            ev = torch.cuda.Event(); ev.record()   # the next decode may overwrite the TS buffers only after these copies
            for sg in segs:
                sg["stream"].wait_event(ev)
            pending[b] = dist.gather(ts_send[b], gathered[b], dst=0, async_op=True)
            nstep[0] += 1
        return reps

    def drain():
        if dist:
            for b in range(2):
                if pending[b] is not None:
                    pending[b].wait(); pending[b] = None
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        reps = step()
    drain()
    for sg in segs:
        sg["rx"].enable_timing(True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        reps = step()
    drain()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        msps = world * nsamp * a.steps / dt / 1e6
        # dominant kernel = viterbi3_kernel: per launch, algorithmic bytes = bytes in (one per m coded bits) + decoded
        # bytes out (6048 + 3969 B per 8k QAM64 7/8 OFDM symbol); launch duration from HIP events
        # recorded on the segment's own stream; averages over the segments' launches
        d = rx.dims
        vit_ms = sum(sg["rx"].stage_ms("viterbi") for sg in segs) / nseg
        alg_bytes = sum(r.n_out_symbols * d.payload_length + r.n_viterbi_bytes for r in reps) / nseg
        rep = reps[0]
        n_ts = sum(int(r.n_ts_bytes) for r in reps)
        achieved = alg_bytes / (vit_ms * 1e-3) / 1e9 if vit_ms > 0 else 0.0
        # HBM bytes of that kernel from the committed rocprofv3 PMC passes (tools/pmc.sh; FETCH_SIZE doubled per the
        # gfx950 note in MI355X_MICROARCH.md), scaled per algorithmic byte of the profiled launch
        traffic = None
        valu_frac = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary_8k_qam64_7_8_65sf.json")))
            if a.workload == "8k_qam64_7_8":
                kv = pm["kernels"]["viterbi3_kernel"]
                traffic = int(kv["hbm_bytes_corrected"] / pm["viterbi_algorithmic_bytes"] * alg_bytes)
                # what actually bounds the kernel: VALU issue slots used = wavefront VALU instructions x 4 cycles / (1024 SIMDs x
                # kernel cycles); GRBM_GUI_ACTIVE sums the 8 XCDs (same committed PMC passes)
                valu_frac = round(kv["SQ_INSTS_VALU"] * 4 / 1024 / (kv["GRBM_GUI_ACTIVE"] / 8), 3)
        except Exception:
            traffic = None
        out = {
            "metric": "RX Msamples/s (baseband in -> TS out)", "value": round(msps, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 (Viterbi/RS) + f32 (front end)",
            "data": "synthetic", "x_realtime": round(msps / REALTIME_MSPS / world, 1),
            "config": {"workload": f"{a.workload} GI 1/32 RX chain, clean TX->RX loopback" + (", input at the 10 Msps file rate (resampler 64/70 + scale on the device)" if a.from_file_rate else ""), "superframes_per_gpu": a.superframes + nseg, "segments_per_gpu": nseg,
                       "samples_per_gpu_per_step": nsamp, "parallelism": f"segments x{world}" + (" + RCCL gather of TS" if world > 1 else ""),
                       "ts_bytes_per_step": n_ts, "status": [int(r.status) for r in reps], "rs_fail_words": [int(r.rs_fail_words) for r in reps]},
            "roofline": {"bound": "hbm", "kernel": "viterbi3_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "valu_issue_frac": valu_frac,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(vit_ms, 4),
                         "chain_frac": round(msps / world * 1e6 * (8 + n_ts / nsamp) / 1e9 / HBM_PEAK_GBS, 6),
                         "hbm_copy_gbs": hbm_copy_gbs(torch, f"cuda:{local}")},
            "stage_ms_per_segment": {k: round(sum(sg["rx"].stage_ms(k) for sg in segs) / nseg, 4) for k in ("acq", "fft", "demod", "inner", "viterbi", "rs", "total")},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_superframes)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    for sg in segs:
        sg["rx"].close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
