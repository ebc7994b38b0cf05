"""Multi-GPU / multi-stream glue of the DVB-T receive path.

The path shards over independent baseband segments: ONE long stream is cut near superframe
boundaries, every piece is decoded by its own chain (other handle, HIP stream or GPU -- one process
per GPU), and the decoded TS pieces are trimmed and concatenated.  The result is, byte for byte, the
TS that a single chain over the whole stream delivers (tests/test_gpu_cut.py).  There is no data-path
collective; the only exchange is ONE gather of the decoded packets on rank 0 (RCCL over xGMI with the
"nccl" backend; the same code runs under "gloo" on CPU tensors, which is how tests/ cover it here).

How a stream is cut (plan_cuts):
  * a cheap pre-scan of the stream's head (any chain run over the first ~1.3 superframes) tells where
    the reference's chain starts decoding: the call grid origin and the call that holds the first
    superframe start (demod_reference_signals_impl.cc:118-136; with d_fi_start = 2 for 8k QAM64 that
    is the start of frame 3, one frame early -- every later boundary is a multiple of 272 symbols on);
  * piece k > 0 begins PRE_SYMBOLS before its first boundary: the chain needs a CP lock and ONE WHOLE
    TPS frame (68 symbols, reference_signals_impl.cc:973-1028) before it can recognise a boundary,
    and must not see two (PRE_SYMBOLS < 136).  Cuts lie on the pre-scan's call grid, so the windows of
    ofdm_sym_acquisition coincide with those of the single chain;
  * piece k ends post_symbols() after the next boundary: the first RS words of piece k+1 mix the byte
    de-interleaver's zero fill with data (convolutional_deinterleaver_impl.cc:64-65) and its
    descrambler locks on its first NSYNC, so piece k delivers those packets;
  * piece k declares its distance to the stream's first superframe start (dvbt_rx_set_cut) so that all
    block / item roundings are the whole-stream chain's.
stitch_ts does the trimming from the pieces' reports alone (no content matching).
"""
import numpy as np

PRE_SYMBOLS = 76          # 68 (one TPS frame) + acquisition and DBPSK start-up margin; < 136


def split_superframes(n_superframes, world):
    """Contiguous, near-equal runs of whole superframes per rank: [(first, count)] * world."""
    base, rem = divmod(n_superframes, world)
    out, first = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((first, cnt))
        first += cnt
    return out


def info_bits_per_symbol(dims):
    return dims.payload_length * dims.m * dims.cr_k // dims.cr_n


def words_per_superframe(dims):
    return 272 * info_bits_per_symbol(dims) // (8 * 204)


def post_symbols(dims):
    """Symbols a piece must hold beyond the next piece's boundary: 11 words of de-interleaver fill + up to 7 to the
    next NSYNC + one group, the two-item hold-back of energy_descramble (16 words, first piece only), the even-item
    rounding (16), one Viterbi block and the ntraceback delay (< 5 words), demod's look-ahead and the last window."""
    words = 80
    return -(-words * 204 * 8 // info_bits_per_symbol(dims)) + 3


def plan_cuts(dims, n_samples, grid0, sf_call, parts, pre=PRE_SYMBOLS, post=None):
    """Cut a stream of n_samples into `parts` pieces of whole superframes.

    grid0: sample of the stream at which the pre-scan's call 0 begins (its report's segment_offset);
    sf_call: call (window of N+cp samples) that delivers the first superframe start (first_call +
    first_out_symbol of the pre-scan's report).
    Returns [{begin, end, sym_off, first, count}]: samples [begin, end) of the stream, the cut offset to declare
    (dvbt_rx_set_cut) and the superframes (counted from the stream's first superframe start) the piece owns.
    Fewer pieces than asked for come back when the stream holds fewer whole superframes."""
    L = dims.fft_length + dims.cp_length
    win = 2 * dims.fft_length + dims.cp_length + 16
    if post is None:
        post = post_symbols(dims)
    ncalls = (n_samples - grid0 - win) // L + 1
    nsf = max((ncalls - sf_call) // 272, 1)
    parts = max(1, min(parts, nsf))
    out = []
    for k, (first, count) in enumerate(split_superframes(nsf, parts)):
        begin = 0 if k == 0 else grid0 + (sf_call + 272 * first - pre) * L
        end = n_samples if k == parts - 1 else min(n_samples, grid0 + (sf_call + 272 * (first + count) + post) * L + win)
        out.append({"begin": int(begin), "end": int(end), "sym_off": 272 * first, "first": first, "count": count})
    return out


# ------------------------------------------------------------------ stitching
META_FIELDS = ("stream_symbol_offset", "ts_first_packet", "n_ts_bytes", "stream_rs_items", "status")


def piece_meta(report):
    """The report fields stitch_ts needs, as a plain dict (works for ctypes reports and for oracle dicts)."""
    get = (lambda k: report[k]) if isinstance(report, dict) else (lambda k: getattr(report, k))
    return {k: int(get(k)) for k in META_FIELDS}


def stitch_plan(metas, dims):
    """Which bytes of every piece's TS tap make up the stream's TS.

    metas: piece_meta() of the pieces in stream order.  Returns ([(byte_begin, byte_end)] per piece, total bytes).
    Piece k's TS tap starts at packet G_k = (its cut offset in RS words) + ts_first_packet of the stream; it is kept from
    G_k to G_{k+1}.  The total is what one chain over the whole stream delivers: it locks on packet q0 = G_0 and holds
    back two items at the end (energy_descramble_impl.cc:121-141): (stream_rs_items - 2 floor(q0 / 16) - 2) items."""
    wsf = words_per_superframe(dims)
    G = []
    for m in metas:
        if m["status"] & ~2:          # bit 1 alone = the CP lock ended with the piece's signal (zeros after the last symbol)
            raise ValueError(f"piece with status {m['status']} cannot be stitched")
        if m["stream_symbol_offset"] % 272:
            raise ValueError("cut offsets are whole superframes")
        G.append(m["stream_symbol_offset"] // 272 * wsf + m["ts_first_packet"])
    if metas[0]["stream_symbol_offset"] != 0:
        raise ValueError("the first piece must hold the beginning of the stream")
    q0 = G[0]
    total_packets = (metas[-1]["stream_rs_items"] - 2 * (q0 // 16) - 2) * 8
    if total_packets < 0:
        total_packets = 0
    end_packet = q0 + total_packets
    spans = []
    for k, m in enumerate(metas):
        have = m["n_ts_bytes"] // 188
        stop = G[k + 1] if k + 1 < len(metas) else end_packet
        stop = min(stop, end_packet)
        if stop < G[k]:
            stop = G[k]
        if G[k] + have < stop:
            raise ValueError(f"piece {k} ends at packet {G[k] + have}, the next starts at {stop}: post-roll too short")
        spans.append((0, (stop - G[k]) * 188))
    return spans, total_packets * 188


def stitch_ts(pieces, dims):
    """pieces: [(report or meta dict, TS bytes of that piece)] in stream order -> the stream's TS (numpy or torch,
    whatever the pieces are)."""
    metas = [piece_meta(p[0]) for p in pieces]
    spans, total = stitch_plan(metas, dims)
    chunks = [p[1][a:b] for p, (a, b) in zip(pieces, spans)]
    if isinstance(chunks[0], np.ndarray):
        out = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    else:
        import torch
        out = torch.cat(chunks)
    assert len(out) == total, (len(out), total)
    return out


# ------------------------------------------------------------------ the single exchange step
HEADER_BYTES = 64         # 5 int64 of META_FIELDS (+ padding) in front of the packets: counts travel with the data


def pack_piece(buf, meta, ts):
    """Fill a rank's send buffer (uint8 torch tensor of HEADER_BYTES + cap): header = piece_meta, then the TS bytes."""
    import torch
    hdr = torch.tensor([meta[k] for k in META_FIELDS] + [0] * (HEADER_BYTES // 8 - len(META_FIELDS)), dtype=torch.int64)
    buf[:HEADER_BYTES].copy_(hdr.view(torch.uint8).to(buf.device), non_blocking=True)
    n = meta["n_ts_bytes"]
    if n > len(buf) - HEADER_BYTES:
        raise ValueError("TS piece larger than the gather buffer")
    if ts is not None and n:
        buf[HEADER_BYTES:HEADER_BYTES + n].copy_(ts[:n], non_blocking=True)
    return buf


def unpack_piece(buf):
    import torch
    hdr = buf[:HEADER_BYTES].cpu().clone().view(torch.int64)      # (clone: a slice of a host buffer keeps its storage offset, which need not be 8-aligned)
    meta = {k: int(hdr[i]) for i, k in enumerate(META_FIELDS)}
    return meta, buf[HEADER_BYTES:HEADER_BYTES + meta["n_ts_bytes"]]


def gather_pieces(send, recv=None, dst=0, group=None, async_op=False):
    """ONE collective: every rank's packed piece (pack_piece) to `dst`.  recv: list of world buffers on dst.
    Returns the work handle when async_op, else None."""
    import torch.distributed as dist
    if dist.get_rank(group) == dst:
        return dist.gather(send, recv, dst=dst, group=group, async_op=async_op)
    return dist.gather(send, None, dst=dst, group=group, async_op=async_op)


def gather_ts(ts, meta, cap, dst=0, group=None):
    """Gather the ranks' TS pieces on `dst` in a single collective (fixed-stride buffers, the counts ride in the header).
    Returns [(meta, uint8 tensor)] per rank on dst (ready for stitch_ts), None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    send = torch.zeros(HEADER_BYTES + cap, dtype=torch.uint8, device=ts.device)
    pack_piece(send, meta, ts)
    recv = [torch.empty(HEADER_BYTES + cap, dtype=torch.uint8, device=ts.device) for _ in range(world)] if rank == dst else None
    gather_pieces(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return [unpack_piece(b) for b in recv]
