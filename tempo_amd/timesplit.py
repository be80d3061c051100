"""Series that cannot be sharded by key: cut along the time axis instead.

A table without partition columns is one global series (python/tempo/tsdf.py:570-571 builds the
window with an empty partitionBy), and a hot key outside skew mode is one series too
(python/tempo/tsdf.py:121): hashing the key leaves all of its rows on one GPU.  Here the ranks
take consecutive time ranges -- rank r owns the rows with cuts[r-1] <= ts < cuts[r], the cuts being
quantiles of a sample of the time column(s) that every rank sees -- and each operator gets the
context it needs from before its range:

* withRangeStats (rangeBetween(-W, 0) over ts.cast("long"), python/tempo/tsdf.py:689-691): the
  rows of the W seconds before the range, per key.  They travel with the owned rows, flagged.
* EMA (window lags, python/tempo/tsdf.py:627-633) and withLookbackFeatures (rowsBetween(-L, -1),
  python/tempo/tsdf.py:656-659): the last window-1 / L rows per key before the range.  Every rank
  offers the tail of each of its keys, the tails are all-gathered and a rank keeps, per key, the
  last rows among those of the ranks below it.
* the plain AS-OF join (last(..., ignoreNulls) over an unbounded preceding window,
  python/tempo/tsdf.py:117-135): per key the last right row before the range and, with skipNulls,
  the last row in which each right column is not NULL.  Whole rows are carried: a row that is the
  last non-NULL one for its column is later than every other non-NULL value of that column, so
  the running "last" over carried + owned rows is the one over the whole history.

Cuts fall on whole seconds where the operator orders by ts.cast("long"), so rows of one second --
peers of one RANGE frame, ties of the window order -- are never separated; ties keep their
global input order when the ranks' inputs are consecutive chunks of the global table (rows
arrive ordered by source rank, then source position).

Context rows produce no output of their own: the results of a rank are those of its owned rows.
With hot_keys=[...] only those keys are cut along the time axis; the rows of every other key go,
whole, to the key's hash owner in the same exchange (a table with a Zipf head and a long tail).
The tails exchanged are n_keys x halo rows per rank, which is what makes this the path for one
series or a few hot ones and not for a million keys (those hash).

Every rank's part is written once, as a generator that yields its collective requests; GroupComm
answers them with torch.distributed (RCCL, or gloo in rehearsals), SimComm plays all ranks in one
process (tests on a single GPU or on the CPU).
"""
import torch
import torch.distributed as dist

from . import dist as keydist
from . import ops

SAMPLES_PER_RANK = 4096


class GroupComm:
    """The collectives of a torch.distributed process group; this process is one rank."""

    def __init__(self, group=None, partition_fn=None, take_fn=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.local = [dist.get_rank(group)]
        self._partition_fn, self._take_fn = partition_fn, take_fn

    def gather(self, reqs):
        """reqs[0]: columns of equal length.  Answer: for every rank, its columns."""
        (cols,) = reqs
        world, group = self.world, self.group
        dev = cols[0].device
        staged = keydist._host_staged(group, cols[0])
        n = cols[0].shape[0]
        cnt = torch.tensor([n], dtype=torch.int64, device="cpu" if staged else dev)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt, group=group)
        sizes = [int(c) for c in cnts]
        longest = max(sizes)
        out = [[] for _ in range(world)]
        for c in cols:
            buf = torch.zeros(longest, dtype=c.dtype, device="cpu" if staged else dev)
            buf[:n] = c
            recv = [torch.empty_like(buf) for _ in range(world)]
            if longest:
                dist.all_gather(recv, buf, group=group)
            for r in range(world):
                out[r].append(recv[r][:sizes[r]].to(dev))
        return [out]

    def exchange(self, reqs):
        """reqs[0] = (dest int32 per row, columns).  Answer: the columns of the rows sent here,
        ordered by source rank, then source position."""
        ((dest, cols),) = reqs
        ident = torch.arange(self.world, dtype=torch.int32, device=dest.device)
        _, got, _, _ = keydist.shuffle_by_key(dest, cols, self.group, self._partition_fn, self._take_fn,
                                              owners=keydist.OwnerPlan(ident))
        return [got]


class SimComm:
    """All ranks in this process, in lock-step: the same routing kernels, lists instead of links."""

    def __init__(self, world, partition_fn=None, take_fn=None):
        self.world = world
        self.local = list(range(world))
        self._partition_fn = partition_fn or keydist._hip_partition
        self._take_fn = take_fn or keydist._hip_take

    def gather(self, reqs):
        return [list(reqs) for _ in range(self.world)]

    def exchange(self, reqs):
        world = self.world
        pieces = [[None] * world for _ in range(world)]  # [destination][source] -> columns
        for s, (dest, cols) in enumerate(reqs):
            ident = torch.arange(world, dtype=torch.int32, device=dest.device)
            if dest.numel():
                perm, counts = self._partition_fn(dest, world, ident)
                staged = [self._take_fn(c, perm) for c in cols]
                ends = torch.cumsum(counts.to(torch.int64), 0).tolist()
            else:
                staged, ends = list(cols), [0] * world
            at = 0
            for r in range(world):
                pieces[r][s] = [c[at:ends[r]] for c in staged]
                at = ends[r]
        n_cols = len(reqs[0][1])
        return [[torch.cat([pieces[r][s][ci] for s in range(world)]) for ci in range(n_cols)] for r in range(world)]


def _drive(comm, rank_body):
    """Runs rank_body(rank) -- a generator -- for every rank this process plays, answering the
    ("gather", cols) / ("exchange", dest, cols) requests they yield; ("done", result) ends a rank."""
    gens = [rank_body(r) for r in comm.local]
    reqs = [next(g) for g in gens]
    while reqs[0][0] != "done":
        kind = reqs[0][0]
        assert all(r[0] == kind for r in reqs), "ranks out of step"
        if kind == "gather":
            answers = comm.gather([r[1] for r in reqs])
        else:
            answers = comm.exchange([(r[1], r[2]) for r in reqs])
        reqs = [g.send(a) for g, a in zip(gens, answers)]
    return [r[1] for r in reqs]


def _cuts(world, ts_list, unit):
    """world - 1 cut times, the same on every rank: quantiles of a sample of everybody's time
    columns, rounded down to a multiple of `unit`."""
    picks = [t[::max(1, t.numel() // SAMPLES_PER_RANK)] for t in ts_list if t.numel()]
    mine = torch.cat(picks) if picks else ts_list[0][:0]
    everyone = yield ("gather", [mine])
    sample = torch.cat([cols[0] for cols in everyone]).cpu().sort().values
    if sample.numel() == 0:
        return torch.zeros(world - 1, dtype=torch.int64)
    at = (torch.arange(1, world, dtype=torch.int64) * sample.numel()) // world
    return torch.div(sample[at], unit, rounding_mode="floor") * unit


def _owner(ts, cuts):
    return torch.bucketize(ts, cuts.to(ts.device), right=True)


class _Routing:
    """Which rows go by time range and which by key.  hot_keys=None: every key is cut along the
    time axis.  Otherwise only the listed key ids are; the rows of every other key go, whole, to
    cold_owner[key] (default: the hash owner, tmp_key_owner) and need no context from elsewhere."""

    def __init__(self, world, n_keys, hot_keys, cold_owner, dev):
        self.all_hot = hot_keys is None
        if self.all_hot:
            return
        hot = torch.zeros(n_keys, dtype=torch.bool)
        hot[torch.as_tensor(list(hot_keys), dtype=torch.int64)] = True
        if cold_owner is None:
            from . import _lib
            cold_owner = torch.tensor([_lib.lib().tmp_key_owner(k, world) for k in range(n_keys)], dtype=torch.int64)
        self.hot = hot.to(dev)
        self.cold_owner = torch.as_tensor(cold_owner).to(torch.int64).to(dev)

    def is_hot(self, key):
        if self.all_hot:
            return torch.ones(key.shape[0], dtype=torch.bool, device=key.device)
        return self.hot[key.long()]

    def owner(self, key, ts, cuts):
        o = _owner(ts, cuts)
        if self.all_hot:
            return o
        k = key.long()
        return torch.where(self.hot[k], o, self.cold_owner[k])

    def sample_of(self, key, ts):
        return ts if self.all_hot else ts[self.hot[key.long()]]


def _filled(t, n, dev, dtype, value):
    return torch.full((n,), value, dtype=dtype, device=dev) if t is None else t


def _tails(lops, key, ts, seq, seq_valid, n_keys, halo, floor_seconds):
    """Rows that are among the last `halo` of their key in window order, in that order."""
    n = ts.numel()
    if n == 0 or halo <= 0:
        return torch.zeros(0, dtype=torch.int64, device=ts.device)
    perm = lops.sort_order(key, ts, n_keys, floor_seconds, seq, seq_valid).long()
    ks = key[perm]
    seg_end = torch.searchsorted(ks, ks, right=True)
    keep = (seg_end - torch.arange(n, device=ts.device)) <= halo
    return perm[keep]


def _last_rows(lops, key, ts, seq, seq_valid, n_keys, masks):
    """Per key the last row in (ts, seq) order and, for every validity column in `masks`, the
    last row that column marks valid; each row once, in window order."""
    n = ts.numel()
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=ts.device)
    perm = lops.sort_order(key, ts, n_keys, False, seq, seq_valid).long()
    ks = key[perm].long()
    pos = torch.arange(n, device=ts.device)
    picked = []
    for m in [None] + list(masks):
        p = pos if m is None else torch.where(m[perm] != 0, pos, torch.full_like(pos, -1))
        best = torch.full((n_keys,), -1, dtype=torch.int64, device=ts.device)
        best = best.scatter_reduce(0, ks, p, "amax", include_self=True)
        picked.append(best[best >= 0])
    rows = torch.unique(torch.cat(picked))  # sorted
    return perm[rows]


def _carry(rank, everyone):
    """The rows offered by the ranks below `rank`, rank by rank (so time-ordered range by range)."""
    n_cols = len(everyone[0])
    if rank == 0:
        return [everyone[0][ci][:0] for ci in range(n_cols)]
    return [torch.cat([everyone[r][ci] for r in range(rank)]) for ci in range(n_cols)]


class _Table:
    """Named columns of equal length; None columns are left out and come back as None."""

    def __init__(self, **cols):
        self.names = [k for k, v in cols.items() if v is not None]
        self.cols = [cols[k] for k in self.names]

    def like(self, cols):
        t = _Table()
        t.names, t.cols = self.names, list(cols)
        return t

    def get(self, name):
        return self.cols[self.names.index(name)] if name in self.names else None

    def take(self, idx):
        return self.like([c[idx] for c in self.cols])


def _extras(t):
    return {"extra%d" % i: e for i, e in enumerate(t.get("extra") or [])}


def _prep(t, n_keys):
    """key / validity columns made explicit (a table without partition columns has key 0)."""
    ts = t["ts"]
    n, dev = ts.numel(), ts.device
    key = _filled(t.get("key"), n, dev, torch.int32, 0)
    return key, ts, n, dev


def range_stats_by_time(comm, tables, n_keys, range_back_secs, ts_is_seconds=False, local_ops=None,
                        hot_keys=None, cold_owner=None):
    """withRangeStats over time ranges.  tables: per rank this process plays, dict(ts=, x=,
    key=None, x_valid=None, extra=[columns to pass through]).  Returns per rank the statistics of
    its owned rows in (key, second) order, with key, ts and the extra columns alongside.
    hot_keys / cold_owner: see _Routing (only the hot keys are cut along time, the rest hash)."""
    lops = local_ops or ops
    world = comm.world
    unit = 1 if ts_is_seconds else 1_000_000
    by_rank = dict(zip(comm.local, tables))

    def body(rank):
        t = by_rank[rank]
        key, ts, n, dev = _prep(t, n_keys)
        x_valid = _filled(t.get("x_valid"), n, dev, torch.uint8, 1)
        tab = _Table(key=key, ts=ts, x=t["x"], x_valid=x_valid, **_extras(t))
        route = _Routing(world, n_keys, hot_keys, cold_owner, dev)
        cuts = yield from _cuts(world, [route.sample_of(key, ts)], unit)
        own = route.owner(key, ts, cuts)
        hot = route.is_hot(key)
        cl = cuts.tolist()
        # rank r needs the rows from second(start of its range) - W on; which lower ranks can hold
        # such rows follows from the cuts alone, so every rank loops over the same distances
        need = [None] + [(c // unit - range_back_secs) * unit for c in cl]
        d_max = 0
        for r in range(1, world):
            o = r - 1
            while o > 0 and cl[o - 1] > need[r]:
                o -= 1
            d_max = max(d_max, r - o)
        need_d = torch.tensor([0] + need[1:], dtype=torch.int64, device=dev)
        dests, sels = [own], []
        for d in range(1, d_max + 1):
            r = own + d
            idx = (hot & (r < world) & (ts >= need_d[r.clamp(max=world - 1)])).nonzero().squeeze(1)
            dests.append(r[idx])
            sels.append(idx)
        n_ctx = sum(i.numel() for i in sels)
        flag = torch.cat([torch.zeros(n, dtype=torch.uint8, device=dev), torch.ones(n_ctx, dtype=torch.uint8, device=dev)])
        cols = [torch.cat([c] + [c[i] for i in sels]) for c in tab.cols] + [flag]
        got = yield ("exchange", torch.cat(dests).to(torch.int32), cols)
        mine, flag = tab.like(got[:-1]), got[-1]
        out = lops.range_stats(mine.get("key"), mine.get("ts"), mine.get("x"), mine.get("x_valid"), n_keys,
                               range_back_secs, ts_is_seconds)
        perm = out["perm"].long()
        keep = flag[perm] == 0
        res = {k: v[keep] for k, v in out.items() if k != "perm"}
        kept = mine.take(perm[keep])
        res["key"], res["ts"] = kept.get("key"), kept.get("ts")
        res["extra"] = [kept.get(nm) for nm in kept.names if nm.startswith("extra")]
        yield ("done", res)

    return _drive(comm, body)


def _row_halo(comm, tables, n_keys, halo, ts_is_seconds, lops, col_names, run_local, hot_keys=None, cold_owner=None):
    """Common part of EMA and lookback: owned rows by time range, then per key the last `halo`
    rows from the ranges below.  run_local(table, n_carry) computes the rank's result."""
    world = comm.world
    unit = 1 if ts_is_seconds else 1_000_000
    by_rank = dict(zip(comm.local, tables))

    def body(rank):
        t = by_rank[rank]
        key, ts, n, dev = _prep(t, n_keys)
        named = {nm: t.get(nm) for nm in col_names}
        tab = _Table(key=key, ts=ts, seq=t.get("seq"), seq_valid=t.get("seq_valid"), **named, **_extras(t))
        route = _Routing(world, n_keys, hot_keys, cold_owner, dev)
        cuts = yield from _cuts(world, [route.sample_of(key, ts)], unit)
        got = yield ("exchange", route.owner(key, ts, cuts).to(torch.int32), tab.cols)
        mine = tab.like(got)
        order = lambda tb, h: _tails(lops, tb.get("key"), tb.get("ts"), tb.get("seq"), tb.get("seq_valid"), n_keys, h,
                                     not ts_is_seconds)
        offer = order(mine, halo)
        offer = offer[route.is_hot(mine.get("key")[offer])]  # keys that hash are whole where they are
        everyone = yield ("gather", mine.take(offer).cols)
        lower = tab.like(_carry(rank, everyone))
        carry = lower.take(order(lower, halo))
        n_carry = carry.cols[0].shape[0]
        full = tab.like([torch.cat([a, b]) for a, b in zip(carry.cols, mine.cols)])
        yield ("done", run_local(full, n_carry))

    return _drive(comm, body)


def ema_by_time(comm, tables, n_keys, window=30, exp_factor=0.2, ts_is_seconds=False, local_ops=None,
                hot_keys=None, cold_owner=None):
    """EMA over time ranges.  tables: dict(ts=, x=, key=None, x_valid=None, seq=None,
    seq_valid=None, extra=[]) per rank.  Returns per rank dict(key, ts, ema, extra) of its owned
    rows in window order."""
    lops = local_ops or ops

    def run_local(tb, n_carry):
        perm, out = lops.ema(tb.get("key"), tb.get("ts"), tb.get("x"), tb.get("x_valid"), n_keys, window, exp_factor,
                             ts_is_seconds, tb.get("seq"), tb.get("seq_valid"))
        perm = perm.long()
        keep = perm >= n_carry
        kept = tb.take(perm[keep])
        return dict(key=kept.get("key"), ts=kept.get("ts"), ema=out[keep],
                    extra=[kept.get(nm) for nm in kept.names if nm.startswith("extra")])

    return _row_halo(comm, tables, n_keys, window - 1, ts_is_seconds, lops, ["x", "x_valid"], run_local, hot_keys,
                     cold_owner)


def lookback_by_time(comm, tables, n_keys, lookback_size, exact_size=True, ts_is_seconds=False, local_ops=None,
                     hot_keys=None, cold_owner=None):
    """withLookbackFeatures over time ranges.  tables: dict(ts=, xs=[feature columns], key=None,
    xs_valid=[... or None], seq=None, seq_valid=None, extra=[]) per rank.  Returns per rank
    dict(key, ts, lengths, feat, feat_valid, extra) of its owned rows that are kept."""
    lops = local_ops or ops
    n_feat = len(tables[0]["xs"])
    flat = []
    for t in tables:
        t = dict(t)
        valid = t.get("xs_valid") or [None] * n_feat
        for i in range(n_feat):
            t["x%d" % i], t["v%d" % i] = t["xs"][i], valid[i]
        flat.append(t)
    names = ["x%d" % i for i in range(n_feat)] + ["v%d" % i for i in range(n_feat)]

    def run_local(tb, n_carry):
        rows, lens, feat, fv = lops.lookback(tb.get("key"), tb.get("ts"), [tb.get("x%d" % i) for i in range(n_feat)],
                                             [tb.get("v%d" % i) for i in range(n_feat)], n_keys, lookback_size,
                                             exact_size, ts_is_seconds, tb.get("seq"), tb.get("seq_valid"))
        rows = rows.long()
        keep = rows >= n_carry
        kept = tb.take(rows[keep])
        return dict(key=kept.get("key"), ts=kept.get("ts"), lengths=None if lens is None else lens[keep],
                    feat=feat[keep], feat_valid=None if fv is None else fv[keep],
                    extra=[kept.get(nm) for nm in kept.names if nm.startswith("extra")])

    return _row_halo(comm, flat, n_keys, lookback_size, ts_is_seconds, lops, names, run_local, hot_keys, cold_owner)


def asof_join_by_time(comm, lefts, rights, n_keys, skip_nulls=True, local_ops=None, hot_keys=None, cold_owner=None):
    """The plain AS-OF join over time ranges.  lefts: dict(ts=, key=None, cols=[8-byte columns])
    per rank; rights: dict(ts=, key=None, cols=[...], masks=[validity or None per column],
    seq=None, seq_valid=None).  Returns per rank what ops.asof_join_table returns for its left
    rows (in (key, ts) order)."""
    lops = local_ops or ops
    world = comm.world
    by_rank = dict(zip(comm.local, zip(lefts, rights)))

    def body(rank):
        lt, rt = by_rank[rank]
        kl, tl, _, _ = _prep(lt, n_keys)
        kr, tr, _, _ = _prep(rt, n_keys)
        masks = rt.get("masks") or [None] * len(rt["cols"])
        # a validity column shared by several right columns travels once and stays shared
        first = {}
        mask_name = [None if m is None else "m%d" % first.setdefault(id(m), i) for i, m in enumerate(masks)]
        left = _Table(key=kl, ts=tl, **{"c%d" % i: c for i, c in enumerate(lt["cols"])})
        right = _Table(key=kr, ts=tr, seq=rt.get("seq"), seq_valid=rt.get("seq_valid"),
                       **{"c%d" % i: c for i, c in enumerate(rt["cols"])},
                       **{nm: masks[i] for i, nm in enumerate(mask_name) if nm == "m%d" % i})
        route = _Routing(world, n_keys, hot_keys, cold_owner, tl.device)
        cuts = yield from _cuts(world, [route.sample_of(kl, tl), route.sample_of(kr, tr)], 1)
        got_l = yield ("exchange", route.owner(kl, tl, cuts).to(torch.int32), left.cols)
        got_r = yield ("exchange", route.owner(kr, tr, cuts).to(torch.int32), right.cols)
        left, mine = left.like(got_l), right.like(got_r)

        def last(tb):
            ms = [tb.get(nm) for nm in sorted(set(n_ for n_ in mask_name if n_))] if skip_nulls else []
            return _last_rows(lops, tb.get("key"), tb.get("ts"), tb.get("seq"), tb.get("seq_valid"), n_keys, ms)

        offer = last(mine)
        offer = offer[route.is_hot(mine.get("key")[offer])]  # keys that hash are whole where they are
        everyone = yield ("gather", mine.take(offer).cols)
        lower = right.like(_carry(rank, everyone))
        carry = lower.take(last(lower))
        full = right.like([torch.cat([a, b]) for a, b in zip(carry.cols, mine.cols)])
        j = lops.asof_join_table(left.get("key"), left.get("ts"), full.get("key"), full.get("ts"), n_keys,
                                 [left.get("c%d" % i) for i in range(len(lt["cols"]))],
                                 [full.get("c%d" % i) for i in range(len(rt["cols"]))],
                                 [None if nm is None else full.get(nm) for nm in mask_name], skip_nulls,
                                 full.get("seq"), full.get("seq_valid"))
        yield ("done", j)

    return _drive(comm, body)
