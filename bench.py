#!/usr/bin/env python3
"""Headline benchmark: AS-OF join + withRangeStats rows/sec at 1B rows on MI355X.

One step = one pass of the hot path over one batch of device-resident synthetic input:
  1. asofJoin of `rows/2` trades against `rows/2` quotes (10 000 Zipf(1.2) symbols, uniform
     microsecond timestamps over one trading day, skew mode tsPartitionVal=10 fraction=0.1,
     three joined right columns, two of them with 1 % NULLs), producing the joined table
     (key, ts, left payload + three right columns with validity) in (key, ts) order;
  2. withRangeStats(rangeBackWindowSecs=600) of one fp64 metric over `rows` ticks
     (100 symbols, Exp(1 s) inter-arrival per symbol, 0.5 % NULLs).
value = rows consumed by both operators (2 x rows) / step time, inputs already in HBM.
With --gpus N (one process per GPU, launched by torch.distributed.run) every rank generates
its own `rows` (weak scaling); the step then starts with the key shuffle (device bucketing by
owner(key) + RCCL all-to-all) and the operators run on the series each rank owns.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _requested_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def _launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as a CHILD process
    (torch.distributed.run), relay rank 0's JSON line and leave with the child's exit status.
    Nothing in this parent has touched the GPU yet, and the parent never replaces itself."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    sys.stdout.buffer.write(proc.stdout)
    sys.stdout.flush()
    sys.exit(proc.returncode)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _requested_gpus(sys.argv[1:]) > 1:
    _launch_ranks(_requested_gpus(sys.argv[1:]))

import torch  # noqa: E402

from tempo_amd import _lib, ops  # noqa: E402

BASE_US = 1_596_240_000_000_000  # 2020-08-01 00:00:00 UTC
DAY_US = 86_400_000_000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def gen_trades_quotes(dev, n_left, n_right, n_keys, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    w = 1.0 / torch.arange(1, n_keys + 1, dtype=torch.float64, device=dev) ** 1.2
    cdf = torch.cumsum(w / w.sum(), 0)

    def zipf(n):
        out = torch.empty(n, dtype=torch.int32, device=dev)
        step = 1 << 26
        for s in range(0, n, step):
            m = min(step, n - s)
            u = torch.rand(m, generator=g, device=dev, dtype=torch.float64)
            out[s:s + m] = torch.searchsorted(cdf, u).clamp_(max=n_keys - 1).to(torch.int32)
        return out

    d = {}
    d["key_l"] = zipf(n_left)
    d["ts_l"] = BASE_US + torch.randint(0, DAY_US, (n_left,), generator=g, device=dev, dtype=torch.int64)
    d["trade_pr"] = 100.0 + torch.randn(n_left, generator=g, device=dev, dtype=torch.float64)
    d["key_r"] = zipf(n_right)
    d["ts_r"] = BASE_US + torch.randint(0, DAY_US, (n_right,), generator=g, device=dev, dtype=torch.int64)
    d["bid"] = 100.0 + torch.randn(n_right, generator=g, device=dev, dtype=torch.float64)
    d["ask"] = d["bid"] + 0.05
    d["bid_valid"] = (torch.rand(n_right, generator=g, device=dev) >= 0.01).to(torch.uint8)
    d["ask_valid"] = (torch.rand(n_right, generator=g, device=dev) >= 0.01).to(torch.uint8)
    return d


def gen_ticks(dev, n, n_keys, seed, shards=1):
    """One shard of the tick stream: n_keys symbols, one tick per second and symbol on average over
    the whole stream.  With `shards` > 1 the stream is striped over that many ranks -- every shard
    is time-ordered, covers the whole period and holds 1 / shards of its ticks -- so that the table
    the ranks hold together is the same stream whatever their number (weak scaling: the period grows
    with the ranks, the ticks per second and symbol do not)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    d = {}
    d["key"] = torch.randint(0, n_keys, (n,), generator=g, device=dev, dtype=torch.int32)
    gaps = torch.empty(n, dtype=torch.float64, device=dev).exponential_(float(n_keys) / shards, generator=g)
    d["ts"] = BASE_US + (torch.cumsum(gaps, 0) * 1e6).to(torch.int64)
    del gaps
    steps = torch.randn(n, generator=g, device=dev, dtype=torch.float64)
    d["x"] = 100.0 + torch.cumsum(steps, 0) * 0.01
    del steps
    d["x_valid"] = (torch.rand(n, generator=g, device=dev) >= 0.005).to(torch.uint8)
    return d


# algorithmic bytes one launch of a kernel moves, per row it processes (every input read once,
# every output written once)
PMC_TRAFFIC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                "r03_bench1b_pmc_traffic.json")

COPY_CEILING_FILE = os.path.join(os.path.dirname(PMC_TRAFFIC_FILE), "r01_copy_ceiling.json")


def copy_ceiling():
    """Measured device-to-device copy rate (GB/s, read + write) of an MI355X, from the committed
    run of tools/copy_ceiling.py: the practical bound beside the nominal 8 TB/s peak."""
    if not os.path.exists(COPY_CEILING_FILE):
        return None
    with open(COPY_CEILING_FILE) as f:
        return json.load(f).get("copy_16GiB_GBps")


KERNEL_BYTES_PER_ROW = {
    "radix_scatter_kernel": 24.0,     # (8 B sort word + 4 B row id) in and out
    "radix_hist_kernel": 8.0,         # sort word in
    "radix_hist_kernel+words": 20.0,  # first pass: key 4 + ts 8 in (+ mask bytes), sort word 8 out
    "radix_hist_kernel+keys": 4.0,    # first pass of a time-ordered feed: the key column alone
    # ... and its scatter, which builds the words: key 4 + ts 8 + NULL flag 1 + metric 8 in, word 8 +
    # row id 4 + metric 8 out (without a travelling column: 13 in, 12 out)
    "radix_scatter_kernel+words+carried": 41.0,
    "radix_scatter_kernel+words32+carried": 37.0,   # ... with the words going out as 32-bit ones
    "range_stats_kernel<T>/words32": 69.0,          # the statistics kernel over such words (4 B less in)
    "radix_scatter_kernel+words": 25.0,
    "range_stats_kernel<T>": 73.0,    # word 8 + metric 8 in; 7 stats x 8 + flags 1 out (the row ids are ordered
                                      # straight into the caller's array by the sort, the kernel does not touch them)
    "gather_kernel<uint64_t>": 21.0,  # idx 4 + value 8 in, value 8 + flag 1 out
    "radix_scatter_kernel+payload": 40.0,  # pair 12 + one 8 B payload column, in and out
    "pack_rows_kernel": 56.0,         # 3 x 8 B columns in, one 32 B record out
    # per left row: word 8 + row id 4 in; right word 8 + 2 x last-non-null 4 + row id 4 + one 16 B
    # payload record in (the two sides have equal row counts); row id 4 + key 4 + ts 8 +
    # 3 x (value 8 + flag 1) out (the left payload column arrives in order from the tile sort)
    "asof_merge_kernel": 107.0,
    # bucketed ordering: word 8 + row id 4 in and out (+ 8 in and out for the travelling payload column)
    "tile_sort_kernel": 24.0,
    "radix_scatter_kernel+carried": 40.0,
    "tile_bounds_kernel": 8.0,
    "key_sample_kernel": 0.25,
    "tile_merge_kernel": 24.0,        # key 4 + ts 8 (+ mask bytes) in, word 8 + row id 4 out
    "ordered_check_kernel": 8.0,
    "lastnn_scan_kernel<false>": 16.0,       # word 8 in, 2 x 4 B out
    "lastnn_reduce_kernel<false>": 8.0,
    "range_scan_kernel": 12.0,        # key 4 + ts 8 in
    "range_scan_kernel+keys": 12.0,   # the same, leaving the key column's histogram for the first radix pass
    "make_comp_kernel": 24.0,         # key 4 + ts 8 in, word 8 + id 4 out (+ mask bytes on the right side)
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=1_000_000_000, help="rows per operator per GPU")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000,
                    help="rows per operator of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-shuffled", action="store_true",
                    help="skip the extra withRangeStats run on a row-permuted tick table")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: native libraries that print banners to file
    # descriptor 1 (RCCL does when it starts) are pointed at stderr until that line goes out
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1: WORLD_SIZE=1 in the environment; unset it (bench.py then starts "
                     "its own ranks) or launch with torch.distributed.run")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    # TEMPO_BENCH_BACKEND=gloo rehearses the --gpus N path where RCCL has no peers: ranks may then
    # share a GPU (rank r uses device r modulo the number of devices) and the exchange is staged
    # through host memory.  The default, and the only measured configuration, is RCCL.
    backend = os.environ.get("TEMPO_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # TEMPO_BENCH_FORCE_SHUFFLE=1 runs the shuffle path on one GPU too (a one-rank RCCL group):
    # a way to rehearse the --gpus N code path where only one GPU is at hand
    force_shuffle = os.environ.get("TEMPO_BENCH_FORCE_SHUFFLE") == "1"
    if world > 1 or force_shuffle:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        from tempo_amd.dist import all_reduce, balanced_owners, plan_wire, shuffle_by_key_start

    verbose = os.environ.get("TEMPO_BENCH_VERBOSE") == "1"

    def log(msg):
        if verbose:
            print("[rank %d %.3f] %s" % (rank, time.perf_counter(), msg), file=sys.stderr, flush=True)

    rows = args.rows
    n_left = n_right = rows // 2
    ASOF_KEYS, TICK_KEYS = 10_000, 100
    log("process group up, generating tables")
    tq = gen_trades_quotes(dev, n_left, n_right, ASOF_KEYS, 4567 + rank)
    tk = gen_ticks(dev, rows, TICK_KEYS, 2345 + rank, shards=world)
    torch.cuda.synchronize()

    profiling = not args.no_profile
    if profiling:
        _lib.profile_enable(True)

    def join(q, marks=None):
        # the joined table in (key, ts) order, as Spark's window leaves it: key / ts / trade_pr of
        # the left side and the three joined right columns, written by the merge itself
        _lib.profile_tag(1)
        if marks:
            marks[0].record()
        j = ops.asof_join_table(
            q["key_l"], q["ts_l"], q["key_r"], q["ts_r"], ASOF_KEYS, [q["trade_pr"]],
            [q["ts_r"], q["bid"], q["ask"]], [None, q["bid_valid"], q["ask_valid"]],
            skip_nulls=True, ts_partition_val=10, fraction=0.1)
        n1 = (q["ts_l"].numel(), q["ts_r"].numel())
        log("join enqueued on %d + %d rows" % n1)
        del j
        if marks:
            marks[1].record()
        return n1

    def stats(t, marks=None):
        _lib.profile_tag(2)
        if marks:
            marks[2].record()
        out = ops.range_stats(t["key"], t["ts"], t["x"], t["x_valid"], TICK_KEYS, 600)
        n2 = t["ts"].numel()
        log("range stats enqueued on %d rows" % n2)
        del out
        if marks:
            marks[3].record()
        return n2

    def start_exchanges(verify=None):
        """Exchange hashpartitioning(key): bucket on device, one all-to-all per column.  Owners come
        from sampled row counts, so that neither the Zipf head of the join keys nor the short symbol
        list of the tick table leaves one GPU with far more rows than the rest; a join key too hot
        for one GPU is cut into time slices at bracket boundaries of the skew join (the quotes then
        carry their bracket tails across slice boundaries).  All three exchanges are issued up front,
        the smallest first.  The columns travel narrowed: key + NULL flags in one or two bytes,
        timestamps as 5-byte offsets from the global minimum (dist.WirePlan).  verify=True: per column,
        the bytes all ranks sent and received must add up to the same checksum (None: as the
        environment says, off by default)."""
        _lib.profile_tag(0)
        own_ticks = balanced_owners(tk["key"], TICK_KEYS)
        own_join = balanced_owners([tq["key_l"], tq["key_r"]], ASOF_KEYS,
                                   skew=dict(ts_partition_val=10, fraction=0.1),
                                   ts=[tq["ts_l"], tq["ts_r"]])
        log("owner plans ready (join keys sliced: %s)" % (own_join.slices is not None))
        c_trades = [tq["ts_l"], tq["trade_pr"]]
        c_quotes = [tq["ts_r"], tq["bid"], tq["ask"], tq["bid_valid"], tq["ask_valid"]]
        c_ticks = [tk["ts"], tk["x"], tk["x_valid"]]
        p_ticks = shuffle_by_key_start(tk["key"], c_ticks, owners=own_ticks,
                                       wire=plan_wire(tk["key"], c_ticks, TICK_KEYS), verify=verify)
        p_trades = shuffle_by_key_start(tq["key_l"], c_trades, owners=own_join, ts=tq["ts_l"],
                                        wire=plan_wire(tq["key_l"], c_trades, ASOF_KEYS), verify=verify)
        p_quotes = shuffle_by_key_start(tq["key_r"], c_quotes, owners=own_join, ts=tq["ts_r"], halo=True,
                                        wire=plan_wire(tq["key_r"], c_quotes, ASOF_KEYS), verify=verify)
        log("exchanges issued")
        return p_ticks, p_trades, p_quotes

    def held_ticks(p_ticks):
        _lib.profile_tag(0)
        kk, (tt, xx, xv), _, _ = p_ticks.wait()
        return dict(key=kk, ts=tt, x=xx, x_valid=xv)

    def held_join_tables(p_trades, p_quotes):
        _lib.profile_tag(0)
        kl, (tl, pr), _, _ = p_trades.wait()
        kr, (tr, bid, ask, bv, av), _, _ = p_quotes.wait()
        return dict(key_l=kl, ts_l=tl, trade_pr=pr, key_r=kr, ts_r=tr, bid=bid, ask=ask, bid_valid=bv, ask_valid=av)

    def step():
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if dist is None:
            n1 = join(tq, marks)
            n2 = stats(tk, marks)
        else:
            # the rank waits for the tick table alone (14 of the 33 GB it sends); the two tables of
            # the join travel while the statistics are worked out
            p_ticks, p_trades, p_quotes = start_exchanges()
            exchange_bytes.append(p_trades.bytes_sent + p_quotes.bytes_sent + p_ticks.bytes_sent)
            n2 = stats(held_ticks(p_ticks), marks)
            n1 = join(held_join_tables(p_trades, p_quotes), marks)
        op_marks.append(marks)
        return n1, n2

    # per operator: events on the stream the operators are called on, before and after each; the
    # join orders its two sides on two streams at once, so the sum of its kernels' own durations
    # would count the overlapped time twice
    op_marks = []
    exchange_bytes = []  # per step: bytes this rank handed to the exchange (all peers, itself included)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    _lib.profile_reset()
    del op_marks[:]
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n1, n2 = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    peak_gib = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)  # of the warm-up and the timed steps
    rows_per_rank = [[n1[0], n1[1], n2]]
    rccl_ranks = 1
    if dist:
        rccl_ranks = dist.get_world_size() if dist.get_backend() == "nccl" else 0
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # what the ranks held after the last step's shuffle: every trade and tick exactly once,
        # every quote once plus the few carried across a time-slice boundary
        r = torch.tensor([n1[0], n1[1], n2], dtype=torch.int64, device=dev)
        per_rank = torch.zeros((world, 3), dtype=torch.int64, device=dev)
        per_rank[rank] = r
        all_reduce(per_rank, dist.ReduceOp.SUM)
        rows_per_rank = [[int(v) for v in row] for row in per_rank.tolist()]
        all_reduce(r, dist.ReduceOp.SUM)
        held = [int(v) for v in r.tolist()]
        if held[0] != world * n_left or held[2] != world * rows or not (
                world * n_right <= held[1] <= world * n_right * 1.01):
            sys.exit("bench.py: rows lost or duplicated by the shuffle: %r" % (held,))
    # rows the two operators consumed: the generated tables, whatever rank processed them
    rows_done = args.steps * world * (n_left + n_right + rows)

    # ---- per-kernel device time from the HIP events recorded on the launch stream
    roofline = None
    kernels = {}
    overlap = {1: 1.0, 2: 1.0}
    op_ms = {1: 0.0, 2: 0.0, 0: 0.0}
    if profiling:
        records = _lib.profile_read()
        kernel_sum = {}
        for name, tag, ms in records:
            kernel_sum[tag] = kernel_sum.get(tag, 0.0) + ms
        ev_ms = {1: sum(m[0].elapsed_time(m[1]) for m in op_marks), 2: sum(m[2].elapsed_time(m[3]) for m in op_marks)}
        # The join orders its two sides on two streams: kernels that share the chip stretch each
        # other, and their durations then add up to more than the operator took.  For the choice of
        # the dominant kernel every kernel's time is scaled by (operator time / sum of its kernels'
        # durations) where that is below one; the durations reported stay as measured.
        overlap = {t: (min(1.0, ev_ms[t] / kernel_sum[t]) if kernel_sum.get(t) else 1.0) for t in (1, 2)}
        for name, tag, ms in records:
            k = kernels.setdefault(name, dict(launches=0, ms=0.0, rows=0.0, op=False, share=0.0))
            k["op"] = k["op"] or tag in (1, 2)
            k["launches"] += 1
            k["ms"] += ms
            k["share"] += ms * overlap.get(tag, 1.0)
            op_ms[tag] = op_ms.get(tag, 0.0) + ms
            # rows one launch covers: the asof launches alternate left/right (rows/2 each), the
            # range-stats launches cover `rows` (single-GPU layout; per-rank counts differ by
            # the key hash under --gpus N and are close to these)
            k["rows"] += (rows / 2 if tag == 1 else rows)
        op_ms[1], op_ms[2] = ev_ms[1], ev_ms[2]
        # the dominant kernel of the two operators (the routing kernels of the shuffle under --gpus N
        # are listed in top_kernels_ms_per_step, they are not the path's kernel)
        of_ops = {n: k for n, k in kernels.items() if k["op"]} or kernels
        name, k = max(of_ops.items(), key=lambda kv: kv[1]["share"])
        bpr = KERNEL_BYTES_PER_ROW.get(name)
        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes
        # (FETCH_SIZE and WRITE_SIZE, separate runs of this command at the default size)
        traffic, traffic_src = None, None
        if rows == 1_000_000_000 and args.gpus == 1 and os.path.exists(PMC_TRAFFIC_FILE):
            with open(PMC_TRAFFIC_FILE) as f:
                pmc = json.load(f).get(name)
            if pmc:
                traffic = pmc["hbm_bytes_per_launch"]
                traffic_src = os.path.relpath(PMC_TRAFFIC_FILE, os.path.dirname(os.path.abspath(__file__)))
        if bpr is not None:
            gbs = k["rows"] * bpr / (k["ms"] * 1e-3) / 1e9
            roofline = dict(bound="hbm", kernel=name, achieved=round(gbs, 1), peak=HBM_PEAK_GBS,
                            unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4), traffic=traffic,
                            launches=k["launches"], avg_launch_ms=round(k["ms"] / k["launches"], 4),
                            bytes_per_row=bpr,
                            algorithmic_bytes_per_launch=int(k["rows"] / k["launches"] * bpr),
                            traffic_source=traffic_src, copy_ceiling=copy_ceiling())
        else:
            roofline = dict(bound="hbm", kernel=name, achieved=None, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=None, traffic=None)

    # ---- the operators alone on what the ranks hold after one more exchange (not part of `value`):
    # the step less the shuffle, i.e. what tables that arrive partitioned by key would cost
    held_ms, delivery = None, None
    if dist:
        _lib.profile_enable(False)
        p_ticks, p_trades, p_quotes = start_exchanges(verify=True)  # untimed: with the delivery check
        try:
            # (the verdict of the check is reduced over the ranks: all of them pass or all of them raise)
            t_held = held_ticks(p_ticks)
            q_held = held_join_tables(p_trades, p_quotes)
            delivery = "ok"
        except RuntimeError as err:
            if not str(err).startswith("key shuffle:"):
                raise   # anything else (memory, a HIP error) may concern this rank alone
            delivery = "FAILED: %s" % err
        del p_ticks, p_trades, p_quotes
        torch.cuda.synchronize()
    if delivery == "ok":
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            join(q_held)
            stats(t_held)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t1) / 2 * 1e3], dtype=torch.float64, device=dev)
        all_reduce(t, dist.ReduceOp.MAX)
        held_ms = round(float(t.item()), 3)
        del t_held, q_held

    # ---- the same withRangeStats on a row-permuted copy of the tick table (not part of `value`):
    # what the operator costs when the feed is not time-ordered -- the full (key, second) sort
    shuffled_ms = None
    if rank == 0 and args.gpus == 1 and not args.no_shuffled:
        torch.cuda.synchronize()
        p = torch.randperm(rows, device=dev)
        ts_p = {k: v[p] for k, v in tk.items()}
        del p
        ops.range_stats(ts_p["key"], ts_p["ts"], ts_p["x"], ts_p["x_valid"], TICK_KEYS, 600)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            ops.range_stats(ts_p["key"], ts_p["ts"], ts_p["x"], ts_p["x_valid"], TICK_KEYS, 600)
        torch.cuda.synchronize()
        shuffled_ms = (time.perf_counter() - t1) / 2 * 1e3
        del ts_p

    # ---- the same join on the same tables laid out in (key, ts) order (not part of `value`): what
    # the operator costs when its input arrives ordered, as it does from an upstream window operator --
    # one sweep over the columns, the left table used by reference
    sorted_ms = None
    if rank == 0 and args.gpus == 1 and not args.no_shuffled:
        sq = {}
        for side, cols in (("l", ["key_l", "ts_l", "trade_pr"]),
                           ("r", ["key_r", "ts_r", "bid", "ask", "bid_valid", "ask_valid"])):
            o = torch.argsort(tq["ts_" + side], stable=True)
            o = o[torch.argsort(tq["key_" + side][o], stable=True)]
            for c in cols:
                sq[c] = tq[c][o].contiguous()
            del o

        def sorted_join():
            return ops.asof_join_table(sq["key_l"], sq["ts_l"], sq["key_r"], sq["ts_r"], ASOF_KEYS, [sq["trade_pr"]],
                                       [sq["ts_r"], sq["bid"], sq["ask"]], [None, sq["bid_valid"], sq["ask_valid"]],
                                       skip_nulls=True, ts_partition_val=10, fraction=0.1, left_by_reference=True)

        sorted_join()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            sorted_join()
        e1.record()
        torch.cuda.synchronize()
        sorted_ms = e0.elapsed_time(e1) / 3
        del sq

    # ---- the same withRangeStats on the tick table laid out in (key, ts) order (not part of `value`):
    # no ordering pass at all, the row ids are the positions
    stats_sorted_ms = None
    if rank == 0 and args.gpus == 1 and not args.no_shuffled:
        o = torch.argsort(tk["key"], stable=True)   # the feed is time-ordered: a stable sort by key gives (key, ts)
        ts_k = {k: v[o].contiguous() for k, v in tk.items()}
        del o
        ops.range_stats(ts_k["key"], ts_k["ts"], ts_k["x"], ts_k["x_valid"], TICK_KEYS, 600)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            ops.range_stats(ts_k["key"], ts_k["ts"], ts_k["x"], ts_k["x_valid"], TICK_KEYS, 600)
        e1.record()
        torch.cuda.synchronize()
        stats_sorted_ms = e0.elapsed_time(e1) / 2
        del ts_k

    # ---- CPU baseline: the C oracle (a port of the reference plan) on a bounded sample, 1 core
    cpu = None
    if rank == 0 and args.gpus == 1 and args.cpu_sample > 0:
        cpu = cpu_baseline(args.cpu_sample)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rows_done / elapsed
        # whole-step view ("% HBM peak" of the metric): the operators' compulsory bytes -- every
        # input read once, every output written once, no sort pass or scratch credited --
        # asofJoin: 12 B key/ts per row of both sides + 3 joined 8-byte columns read and written;
        # withRangeStats: 12 B key/ts + 8 B metric read, 7 x 8 B written
        asof_bytes = (n_left + n_right) * 12 + n_right * 24 + n_left * 24
        stats_bytes = rows * (12 + 8 + 56)
        step_bytes = world * (asof_bytes + stats_bytes)
        if roofline is not None:
            roofline["step_algorithmic_bytes"] = step_bytes
            roofline["step_frac_of_peak"] = round(step_bytes / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4)
            # per operator: compulsory bytes / device time between the events around it (rank 0) / peak
            per_op = {}
            for name, tag, nbytes in (("asofJoin", 1, asof_bytes), ("withRangeStats", 2, stats_bytes)):
                ms = op_ms.get(tag, 0.0) / args.steps
                if ms > 0:
                    per_op[name] = dict(bytes=nbytes, ms=round(ms, 3),
                                        frac=round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            roofline["ops"] = per_op
            # launches the library's event pool could not hold (0 unless --steps is very large);
            # per-kernel figures would then cover only the recorded ones
            roofline["launches_not_recorded"] = _lib.profile_dropped()
        top = sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])[:8]
        line = {
            "metric": "AS-OF join + withRangeStats rows/sec at 1B rows (1-8 GPU); % HBM peak",
            "value": value, "unit": "rows/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64 keys/timestamps + f64 metrics",
            "data": "synthetic",
            "config": {
                "workload": ("asofJoin %d trades x %d quotes per GPU (10k Zipf(1.2) symbols, "
                             "rows in random order, tsPartitionVal=10, fraction=0.1, 3 joined cols) + "
                             "withRangeStats(600 s) on %d ticks per GPU (100 symbols, time-ordered rows "
                             "with interleaved keys)" % (n_left, n_right, rows)),
                "rows_per_gpu_per_op": rows,
                "rccl_ranks": rccl_ranks,
                "exchange_bytes_per_rank": (int(sum(exchange_bytes[-args.steps:]) / args.steps) if exchange_bytes else 0),
                "multi_gpu_measured": ("this line" if rccl_ranks > 1 else
                                       "no: one rank; 2/4/8-GPU timings need a multi-GPU node"),
                "rows_held_per_rank": rows_per_rank,
                "range_stats_shuffled_ms_per_step": None if shuffled_ms is None else round(shuffled_ms, 3),
                "asof_sorted_input_ms_per_step": None if sorted_ms is None else round(sorted_ms, 3),
                "range_stats_sorted_input_ms_per_step": None if stats_sorted_ms is None else round(stats_sorted_ms, 3),
                "peak_hbm_gib_rank0": peak_gib,
                # N > 1: join + statistics alone on what the ranks hold after an exchange (max over ranks)
                "operators_on_held_rows_ms_per_step": held_ms,
                # N > 1: the exchange ahead of that measurement ran with per-column checksums over all ranks
                "exchange_delivery_checked": delivery,
                "parallelism": "key shard x%d (row-count balanced owners)" % args.gpus,
                "asof_device_ms_per_step": round(op_ms.get(1, 0.0) / args.steps, 3),
                "range_stats_device_ms_per_step": round(op_ms.get(2, 0.0) / args.steps, 3),
                "shuffle_device_ms_per_step": round(op_ms.get(0, 0.0) / args.steps, 3),
                "top_kernels_ms_per_step": {n: round(k["ms"] / args.steps, 3) for n, k in top},
                # below 1: the join's kernels ran on two streams and their durations (as listed above)
                # add up to more than the operator took, by this factor's inverse
                "asof_kernel_overlap": (round(overlap[1], 3) if profiling else None),
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if dist:
        dist.destroy_process_group()


def cpu_baseline(sample_rows):
    """Time the CPU oracle on host cores over a scaled-down copy of the same workload."""
    import numpy as np

    from oracle import oracle

    rng = np.random.default_rng(99)
    nl = nr = sample_rows // 2
    nk = 10_000
    w = 1.0 / np.arange(1, nk + 1) ** 1.2
    cdf = np.cumsum(w / w.sum())
    key_l = np.searchsorted(cdf, rng.random(nl)).clip(0, nk - 1).astype(np.int32)
    key_r = np.searchsorted(cdf, rng.random(nr)).clip(0, nk - 1).astype(np.int32)
    ts_l = BASE_US + rng.integers(0, DAY_US, nl)
    ts_r = BASE_US + rng.integers(0, DAY_US, nr)
    vb = (rng.random(nr) >= 0.01).astype(np.uint8)
    va = (rng.random(nr) >= 0.01).astype(np.uint8)
    key = rng.integers(0, 100, sample_rows).astype(np.int32)
    ts = BASE_US + (np.cumsum(rng.exponential(0.01, sample_rows)) * 1e6).astype(np.int64)
    x = 100 + np.cumsum(rng.normal(0, 1, sample_rows)) * 0.01
    xv = (rng.random(sample_rows) >= 0.005).astype(np.uint8)
    oracle.lib()
    # Series are independent per key, so the host baseline shards the sample by key over the cores it
    # may use (one oracle call per shard and thread; ctypes drops the GIL during the call).
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(16, len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16"))))
    sh_l, sh_r, sh_t = key_l % threads, key_r % threads, key % threads
    join_parts = [(key_l[sh_l == i], ts_l[sh_l == i], key_r[sh_r == i], ts_r[sh_r == i],
                   vb[sh_r == i], va[sh_r == i]) for i in range(threads)]
    tick_parts = [(key[sh_t == i], ts[sh_t == i], x[sh_t == i], xv[sh_t == i]) for i in range(threads)]

    def join_one(p):
        oracle.asof_join(p[0], p[1], p[2], p[3], valid_r=[None, p[4], p[5]], ts_partition_val=10,
                         fraction=0.1)

    def ticks_one(p):
        oracle.range_stats(p[0], p[1], p[2], p[3], 600)

    with ThreadPoolExecutor(threads) as pool:
        t0 = time.perf_counter()
        list(pool.map(join_one, join_parts))
        t1 = time.perf_counter()
        list(pool.map(ticks_one, tick_parts))
        t2 = time.perf_counter()
    total = t2 - t0
    mem_total_gib = None
    try:
        with open("/proc/meminfo") as f:
            mem_total_gib = round(int(f.readline().split()[1]) / 2 ** 20, 1)
    except (OSError, ValueError, IndexError):
        pass
    return dict(value=(nl + nr + sample_rows) / total, unit="rows/s", cores=threads, kind="port",
                host_mem_total_gib=mem_total_gib,
                sample=("C oracle (oracle/tempo_oracle.c), %d threads over key shards: asofJoin %d+%d rows "
                        "%.2f s, withRangeStats %d rows %.2f s; same distributions as the GPU workload, "
                        "1/%d of its rows per operator (bounded so that the default run stays within minutes; "
                        "the oracle's union + sort plan holds about 100 bytes per row on the host, "
                        "so the full 10^9 rows per operator would need about 100 GB of host memory)"
                        % (threads, nl, nr, t1 - t0, sample_rows, t2 - t1, max(1, 1_000_000_000 // max(sample_rows, 1)))))


if __name__ == "__main__":
    main()
