"""Multi-GPU key shuffle: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Spark's Exchange hashpartitioning(partitionCols) in front of every Window / groupBy
(python/tempo/tsdf.py:121, :571; python/tempo/resample.py:63) becomes one all-to-all per
column: rows are bucketed on the device by owner(key) (tmp_hash_partition), exchanged with
all_to_all, and every later operator runs locally on the series a rank owns -- no other
collective is on the data path.  owner(key) is mix(key) % world_size, or -- balanced_owners --
a table that spreads the keys by their row counts, so that a hot key (the skew
python/tempo/tsdf.py:164-190 works around) or a short key list does not leave one GPU with
far more rows than the others.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import ops

# Largest amount of data handed to ONE collective call, in bytes, summed over all peers.  Columns
# of 10^9 rows are several GB per rank; they go out as a sequence of all-to-alls over row ranges
# sized so that a call's whole send buffer (and whole receive buffer) stays below this.
# Why: on this RCCL build a rank's send to itself delivers a message of 1 GiB intact, but of one of
# 2 147 479 552 bytes (just under 2 GiB) everything from byte 1 073 741 824 on is lost -- the tail of
# the receive buffer is left untouched -- through every entry point (all_to_all_single with or
# without split sizes, the grouped send/recv form), so the loss is below torch
# (tools/a2a_probe.py reproduces it).  The probe is a one-rank group, where the bytes per peer and the
# bytes per call are the same number: which of the two the fault depends on is not known, so the
# bound is applied to the call as a whole -- at most half a GiB per call, at most that divided by
# the number of ranks per peer, a factor of two below the largest size seen intact.  That the bound
# also protects exchanges between different ranks is extrapolated from that one-rank probe;
# `verify=True` (TEMPO_SHUFFLE_VERIFY=1) checks every exchange by checksum.
MAX_MESSAGE_BYTES = 1 << 29


def _hip_partition(key, n_parts, table=None, slices=None, ts=None, halo=False):
    return ops.hash_partition(key, n_parts, table, slices, ts, halo)


def _hip_take(col, perm):
    return ops.gather(col, perm, want_valid=False)[0]


def _hip_counts(key, n_keys, stride):
    return ops.key_counts(key, n_keys, stride)


def assign_owners(counts, world):
    """Owner per key id from global row counts (host, deterministic: every rank computes the same
    table from the same counts).  Keys are taken in order of decreasing count and each goes to
    the rank with the fewest rows so far (longest-processing-time rule); keys nobody sampled are
    dealt out round-robin.  A key is never split: one that holds more than 1/world of the rows
    still overloads its owner."""
    counts = np.asarray(counts, dtype=np.int64)
    owners = np.empty(len(counts), dtype=np.int32)
    order = np.argsort(-counts, kind="stable")
    seen = order[counts[order] > 0]
    load = np.zeros(world, dtype=np.int64)
    # the heavy head one by one; the long tail of small keys a round at a time, onto the ranks
    # that are still below their fair share (lightest rank first)
    head = min(len(seen), 64 * world)
    for k in seen[:head]:
        r = int(np.argmin(load))
        owners[k] = r
        load[r] += counts[k]
    rest = seen[head:]
    fair = (int(counts.sum()) + world - 1) // world
    lo = 0
    while lo < len(rest):
        ranks = np.argsort(load, kind="stable")
        below = ranks[load[ranks] < fair]
        ranks = below if len(below) else ranks[:1]
        # decreasing counts dealt over increasing loads, 64 rounds per look at the loads (these
        # keys are small next to a rank's share)
        chunk = rest[lo:lo + 64 * len(ranks)]
        to = ranks[np.arange(len(chunk)) % len(ranks)]
        owners[chunk] = to
        load += np.bincount(to, weights=counts[chunk], minlength=world).astype(np.int64)
        lo += len(chunk)
    unseen = order[counts[order] <= 0]
    owners[unseen] = np.arange(len(unseen), dtype=np.int32) % world
    return owners


class OwnerPlan:
    """Who owns which key: `table` (int32[n_keys] on the device) and, when hot keys are cut along
    the time axis, `slices` (the dict ops.hash_partition takes)."""

    def __init__(self, table, slices=None):
        self.table, self.slices = table, slices


def _bracket_index(ts, v, ts_is_seconds):
    """(ts.cast("double") / v).cast("integer") of python/tempo/tsdf.py:178-179 on the host, the
    same IEEE operations as the device (truncation, saturating)."""
    d = np.float64(ts) if ts_is_seconds else np.float64(ts) / np.float64(1000000.0)
    q = np.trunc(d / np.float64(v))
    return int(min(max(q, -2147483648.0), 2147483647.0))


def balanced_owners(keys, n_keys, group=None, stride=16, counts_fn=None, skew=None, ts=None):
    """OwnerPlan that balances the rows of `keys` -- one or more key columns sharing a
    dictionary, e.g. both sides of a join -- over the ranks.  Counts every stride-th row on the
    device, sums the counts over ranks (one small all-reduce) and assigns keys on the host;
    every rank ends up with the same plan.

    skew=dict(ts_partition_val=, fraction=, ts_is_seconds=) with ts (the time columns that go
    with `keys`) allows a key with more than half a rank's fair share of the rows to be cut
    into time slices owned by different ranks.  That is exact for the skew AS-OF join only
    (python/tempo/tsdf.py:164-190: a left row sees its own time bracket and the tail of the one
    before), and the right side must then be shuffled with halo=True."""
    counts_fn = counts_fn or _hip_counts
    if isinstance(keys, torch.Tensor):
        keys = [keys]
    world = dist.get_world_size(group)
    total = None
    for k in keys:
        c = counts_fn(k, n_keys, stride)
        total = c if total is None else total + c
    all_reduce(total, dist.ReduceOp.SUM, group)
    counts = total.cpu().numpy().astype(np.int64)
    dev = keys[0].device
    ts_range = None
    if skew is not None and world > 1 and len(hot_keys(counts, world)):
        if isinstance(ts, torch.Tensor):
            ts = [ts]
        # (min, -max) over the time columns: one sweep per column, reduced over the ranks on the
        # device, read by the host once
        big = torch.iinfo(torch.int64).max
        ends = torch.full((2,), big, dtype=torch.int64, device=dev)
        for t in ts:
            if t.numel():
                mn, mx = torch.aminmax(t)
                ends = torch.minimum(ends, torch.stack([mn, -mx]))
        all_reduce(ends, dist.ReduceOp.MIN, group)
        lo, neg_hi = ends.cpu().tolist()
        ts_range = (int(lo), -int(neg_hi))
    return plan_owners(counts, world, dev, skew, ts_range)


def hot_keys(counts, world):
    """Keys holding more than half of a rank's fair share of the rows."""
    fair = max(1, int(counts.sum()) // world)
    return np.flatnonzero(counts > fair // 2)


def plan_owners(counts, world, device, skew=None, ts_range=None):
    """OwnerPlan from global per-key row counts (host arithmetic only; same inputs, same plan).
    With skew and the global (min, max) of the time columns, hot keys are cut into time slices."""
    counts = np.asarray(counts, dtype=np.int64)
    n_keys = len(counts)
    hot = hot_keys(counts, world) if (skew is not None and ts_range is not None and world > 1) else np.arange(0)
    if len(hot):
        v = skew["ts_partition_val"]
        v_is_int = isinstance(v, (int, np.integer)) and not isinstance(v, bool)
        secs = bool(skew.get("ts_is_seconds", False))
        lo, hi = ts_range
        q_lo, q_hi = _bracket_index(lo, v, secs), _bracket_index(hi, v, secs)
        # slices are cut by bracket index; that is only the join's own cut while v * q neither
        # saturates nor wraps
        usable = (lo <= hi and -2147483648 < q_lo and q_hi < 2147483647 and q_hi > q_lo
                  and (not v_is_int or max(abs(q_lo), abs(q_hi) + 1) * abs(int(v)) < 2 ** 31))
        if usable:
            fair = max(1, int(counts.sum()) // world)
            n_brackets = q_hi - q_lo + 1
            per_slice = max(1, fair // 8)
            n_slices = np.minimum(np.minimum(-(-counts[hot] // per_slice), 256), n_brackets).astype(np.int64)
            items = counts.copy()
            items[hot] = 0
            items = np.concatenate([items] + [np.full(int(S), int(c // S), dtype=np.int64)
                                              for c, S in zip(counts[hot], n_slices)])
            owners = assign_owners(items, world)
            table = owners[:n_keys].copy()
            table[hot] = -(np.arange(len(hot), dtype=np.int32) + 1)
            first = np.concatenate([[0], np.cumsum(n_slices)[:-1]]).astype(np.int32)
            slices = dict(ts_partition_val=float(v), v_is_int=v_is_int, ts_is_seconds=secs,
                          one_minus_fraction=1 - skew["fraction"], bracket_min=q_lo, bracket_count=n_brackets,
                          split_slices=torch.from_numpy(n_slices.astype(np.int32)).to(device),
                          split_first=torch.from_numpy(first).to(device),
                          slice_owner=torch.from_numpy(owners[n_keys:].astype(np.int32)).to(device))
            return OwnerPlan(torch.from_numpy(table).to(device), slices)
    return OwnerPlan(torch.from_numpy(assign_owners(counts, world)).to(device))


def _host_staged(group, t):
    """gloo has no all-to-all for device tensors: under it (rehearsals of the multi-rank path on a
    box without RCCL peers, several ranks sharing one GPU) the exchange is staged through host
    memory.  RCCL is the production path."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce(t, op, group=None):
    """dist.all_reduce, through host memory under gloo (see _host_staged)."""
    if _host_staged(group, t):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _counts_all_to_all(recv, send, group):
    if _host_staged(group, send):
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r, send.cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, group=group)


def _offsets(lens):
    off = [0]
    for v in lens[:-1]:
        off.append(off[-1] + v)
    return off


def _round_views(buf, off, lens, lo, hi):
    """Per peer, the rows [lo, hi) of that peer's segment of `buf` (segments start at off[r] and
    hold lens[r] rows; a segment shorter than `lo` contributes an empty view)."""
    return [buf[off[r] + min(lo, lens[r]): off[r] + min(hi, lens[r])] for r in range(len(lens))]


def _exchange(staged, got, send_l, recv_l, rounds, rows_per_round, group):
    """All-to-all of one column in `rounds` slices: round c moves rows [c*R, (c+1)*R) of what
    every rank sends to every other rank.  Returns the pending work handles (the collectives
    are issued asynchronously, in order).

    RCCL: per round one grouped send/recv over views of the two buffers, nothing is copied.
    gloo (tests and rehearsals; it has no list all-to-all): the round's views are gathered into
    one contiguous message for all_to_all_single and the answer is copied out again; device
    tensors are staged through host memory first."""
    if staged.dtype not in (torch.uint8, torch.int32, torch.int64, torch.float32, torch.float64):
        # element types the collectives do not take (int16 wire columns): the same bytes as uint8
        es = staged.element_size()
        return _exchange(staged.view(torch.uint8), got.view(torch.uint8), [v * es for v in send_l],
                         [v * es for v in recv_l], rounds, rows_per_round * es, group)
    if _host_staged(group, staged):
        h_got = torch.empty(got.shape, dtype=got.dtype)
        for w in _exchange(staged.cpu(), h_got, send_l, recv_l, rounds, rows_per_round, group):
            w.wait()
        got.copy_(h_got)
        return []
    if rounds == 1:
        return [dist.all_to_all_single(got, staged, output_split_sizes=recv_l, input_split_sizes=send_l,
                                       group=group, async_op=True)]
    s_off, r_off = _offsets(send_l), _offsets(recv_l)
    works = []
    for c in range(rounds):
        lo, hi = c * rows_per_round, (c + 1) * rows_per_round
        ins = _round_views(staged, s_off, send_l, lo, hi)
        outs = _round_views(got, r_off, recv_l, lo, hi)
        if dist.get_backend(group) == "gloo":
            msg = torch.empty(sum(o.numel() for o in outs), dtype=got.dtype)
            dist.all_to_all_single(msg, torch.cat(ins), output_split_sizes=[o.numel() for o in outs],
                                   input_split_sizes=[i.numel() for i in ins], group=group)
            at = 0
            for o in outs:
                o.copy_(msg[at:at + o.numel()])
                at += o.numel()
        else:
            works.append(dist.all_to_all(outs, ins, group=group, async_op=True))
    return works


def _checksum(t):
    """Sum of a column's bytes as int64 (wraps; only ever compared with another such sum)."""
    if t.numel() == 0:
        return torch.zeros((), dtype=torch.int64, device=t.device)
    return t.contiguous().view(torch.uint8).sum(dtype=torch.int64)


class WirePlan:
    """How a table's columns travel: the key together with the 0/1 mask columns in the narrowest of
    1 / 2 / 4 bytes that holds (key bits + one bit per mask), every other integer column as its offset
    from the global minimum in 4, 5, 6 or 8 bytes (a day of microseconds needs 5), everything else as
    it is.  Elementwise, so it applies to rows in any order; decode(encode(x)) == x exactly."""

    def __init__(self, key_bits, mask_cols, int_cols):
        self.key_bits = key_bits          # bits of the key field
        self.mask_cols = mask_cols        # positions (among the payload columns) folded into the key word
        self.int_cols = int_cols          # {position: (minimum, bytes)}
        width = key_bits + len(mask_cols)
        self.key_dtype = torch.uint8 if width <= 8 else (torch.int16 if width <= 16 else torch.int32)

    def bytes_per_row(self, columns):
        total = torch.empty(0, dtype=self.key_dtype).element_size()
        for i, c in enumerate(columns):
            if i in self.mask_cols:
                continue
            total += self.int_cols[i][1] if i in self.int_cols else c.element_size()
        return total

    device_kernels = True     # False: the torch expressions on device tensors too (what the tests compare with)

    def _on_device(self, first):
        """Whether the HIP kernels take the call: device tensors, and no more flag / integer columns than
        one call holds (host tensors -- the gloo tests -- and wider tables use the torch expressions)."""
        narrow = sum(1 for _, nb in self.int_cols.values() if nb != 8)
        return self.device_kernels and first.is_cuda and len(self.mask_cols) <= 8 and narrow <= 8

    def encode(self, key, columns):
        """[key, columns...] -> list of wire columns."""
        if self._on_device(key) and key.dtype == torch.int32:
            return self._encode_dev(key, columns)
        word = key.to(torch.int32)
        for b, i in enumerate(self.mask_cols):
            word = word | (columns[i].to(torch.int32) << (self.key_bits + b))
        wire = [word.to(self.key_dtype)]
        for i, c in enumerate(columns):
            if i in self.mask_cols:
                continue
            if i in self.int_cols:
                lo, nbytes = self.int_cols[i]
                if nbytes == 8:
                    wire.append(c)
                    continue
                rel = c - lo
                wire.append(rel.to(torch.int32))   # low four bytes (wraps into the sign bit: a bit pattern)
                if nbytes == 5:
                    wire.append((rel >> 32).to(torch.uint8))
                elif nbytes == 6:
                    wire.append((rel >> 32).to(torch.int16))
            else:
                wire.append(c)
        return wire

    def _encode_dev(self, key, columns):
        """encode through tmp_wire_encode: one sweep over the table instead of a dozen elementwise ones."""
        n, dev = key.numel(), key.device
        word = torch.empty(n, dtype=self.key_dtype, device=dev)
        masks = [columns[i].contiguous() for i in self.mask_cols]
        narrow = [i for i in sorted(self.int_cols) if self.int_cols[i][1] != 8]
        los = {i: torch.empty(n, dtype=torch.int32, device=dev) for i in narrow}
        his = {i: (None if self.int_cols[i][1] == 4 else
                   torch.empty(n, dtype=torch.uint8 if self.int_cols[i][1] == 5 else torch.int16, device=dev)) for i in narrow}
        ops.wire_code(True, key.contiguous(), self.key_bits, word, masks, [columns[i].contiguous() for i in narrow],
                      [self.int_cols[i][0] for i in narrow], [self.int_cols[i][1] for i in narrow],
                      [los[i] for i in narrow], [his[i] for i in narrow])
        wire = [word]
        for i, c in enumerate(columns):
            if i in self.mask_cols:
                continue
            if i in los:
                wire.append(los[i])
                if his[i] is not None:
                    wire.append(his[i])
            else:
                wire.append(c)
        return wire

    def _decode_dev(self, wire, dtypes):
        it = iter(wire)
        word = next(it)
        n, dev = word.numel(), word.device
        key = torch.empty(n, dtype=torch.int32, device=dev)
        cols = [None] * len(dtypes)
        masks = []
        for i in self.mask_cols:
            cols[i] = torch.empty(n, dtype=torch.uint8, device=dev)
            masks.append(cols[i])
        narrow, los, his = [], [], []
        for i, dt in enumerate(dtypes):
            if i in self.mask_cols:
                continue
            if i in self.int_cols and self.int_cols[i][1] != 8:
                narrow.append(i)
                los.append(next(it))
                his.append(None if self.int_cols[i][1] == 4 else next(it))
                cols[i] = torch.empty(n, dtype=torch.int64, device=dev)
            else:
                cols[i] = next(it)
        ops.wire_code(False, key, self.key_bits, word.contiguous(), masks, [cols[i] for i in narrow],
                      [self.int_cols[i][0] for i in narrow], [self.int_cols[i][1] for i in narrow],
                      [t.contiguous() for t in los], [None if t is None else t.contiguous() for t in his])
        for i in self.mask_cols:
            cols[i] = cols[i].to(dtypes[i])
        for i in narrow:
            cols[i] = cols[i].to(dtypes[i])
        return key, cols

    def decode(self, wire, dtypes):
        """Inverse of encode: (key int32, [columns in their own dtypes])."""
        if self._on_device(wire[0]):
            return self._decode_dev(wire, dtypes)
        it = iter(wire)
        word = next(it).to(torch.int32)
        if self.key_dtype == torch.int16:
            word = word & 0xffff
        key = word & ((1 << self.key_bits) - 1) if self.key_bits < 31 else word
        cols = [None] * len(dtypes)
        for b, i in enumerate(self.mask_cols):
            cols[i] = ((word >> (self.key_bits + b)) & 1).to(dtypes[i])
        for i, dt in enumerate(dtypes):
            if i in self.mask_cols:
                continue
            if i in self.int_cols:
                lo, nbytes = self.int_cols[i]
                if nbytes == 8:
                    cols[i] = next(it)
                    continue
                rel = next(it).to(torch.int64) & 0xffffffff
                if nbytes == 5:
                    rel = rel | (next(it).to(torch.int64) << 32)
                elif nbytes == 6:
                    rel = rel | ((next(it).to(torch.int64) & 0xffff) << 32)
                cols[i] = (rel + lo).to(dt)
            else:
                cols[i] = next(it)
        return key.to(torch.int32), cols


def plan_wire(key, columns, n_keys, group=None):
    """WirePlan for one table from the global value ranges of its integer columns (one min / max
    sweep per such column and one small all-reduce for the lot)."""
    dev = key.device
    ints = [i for i, c in enumerate(columns) if c.dtype in (torch.int64, torch.int32, torch.uint8)]
    big = torch.iinfo(torch.int64).max
    ends = torch.full((len(ints), 2), big, dtype=torch.int64, device=dev)   # (min, -max) per column
    for j, i in enumerate(ints):
        if columns[i].numel():
            mn, mx = torch.aminmax(columns[i])
            ends[j, 0], ends[j, 1] = mn.to(torch.int64), -mx.to(torch.int64)
    if len(ints):
        all_reduce(ends, dist.ReduceOp.MIN, group)
    ends = ends.cpu().tolist()
    key_bits = max(1, int(n_keys - 1).bit_length()) if n_keys > 1 else 1
    mask_cols, int_cols = [], {}
    for (lo, neg_hi), i in zip(ends, ints):
        hi = -neg_hi
        if lo > hi:          # no rows anywhere
            lo = hi = 0
        if columns[i].dtype == torch.uint8 and lo >= 0 and hi <= 1 and key_bits + len(mask_cols) < 31:
            mask_cols.append(i)
        elif columns[i].dtype == torch.int64:
            span = hi - lo
            nbytes = 4 if span < 2 ** 32 else (5 if span < 2 ** 40 else (6 if span < 2 ** 48 else 8))
            int_cols[i] = (lo, nbytes)
    return WirePlan(key_bits, mask_cols, int_cols)


class PendingShuffle:
    """A key shuffle whose column exchanges are in flight.  wait() makes the current stream wait
    for them and returns (key_local, columns_local, send_counts, recv_counts).  The staged send
    buffers and the receive buffers stay referenced here until then."""

    def __init__(self, works, staged, received, send_l, recv_l, wire=None, dtypes=None, sent_sums=None, group=None):
        self._works, self._staged, self._received = works, staged, received
        self.send_counts, self.recv_counts = send_l, recv_l
        self._wire, self._dtypes, self._sent_sums, self._group = wire, dtypes, sent_sums, group
        self.bytes_sent = sum(int(s.numel()) * s.element_size() for s in staged)

    def wait(self):
        for w in self._works:
            w.wait()
        self._works, self._staged = [], None
        if self._sent_sums is not None:
            # delivery check: over all ranks, the bytes that arrived add up to the bytes that left
            got = torch.stack([_checksum(c) for c in self._received])
            both = torch.stack([self._sent_sums, got])
            all_reduce(both, dist.ReduceOp.SUM, self._group)
            if not bool((both[0] == both[1]).all()):
                bad = torch.nonzero(both[0] != both[1]).flatten().tolist()
                raise RuntimeError("key shuffle: column(s) %r did not arrive as sent (checksums over all ranks "
                                   "differ)" % (bad,))
            self._sent_sums = None
        if self._wire is not None:
            key, cols = self._wire.decode(self._received, self._dtypes)
            return key, cols, self.send_counts, self.recv_counts
        return self._received[0], self._received[1:], self.send_counts, self.recv_counts


def shuffle_by_key_start(key, columns, group=None, partition_fn=None, take_fn=None, owners=None,
                         ts=None, halo=False, wire=None, verify=None):
    """Start sending every row to the rank that owns its key; returns a PendingShuffle.

    The device bucketing and the count exchange happen here (one small blocking read); the
    column exchanges are issued asynchronously, so work on other tables can be queued behind
    this call and overlaps with the transfers.
    key: int32 tensor of GLOBAL dictionary ids (the same id for the same key on every rank).
    columns: list of 1-D tensors with one entry per row (1-, 4- or 8-byte element types).
    owners: optional OwnerPlan from balanced_owners (the same one for every table that has to
    be co-located, such as the two sides of a join); default is the hash.  A plan with time
    slices needs `ts` (the table's time column, which still has to be listed in `columns` to
    travel), and halo=True for the right side of the join: rows in the carried-over tail of a
    bracket at a slice boundary are then sent to the next slice's owner as well, and arrive
    after the rank's own rows.
    partition_fn / take_fn default to the HIP kernels; tests on CPU inject host equivalents.
    wire: a WirePlan (plan_wire) -- the bucketed columns are narrowed before they travel and widened
    again on arrival; the result is the same tensors as without it.
    verify (default: TEMPO_SHUFFLE_VERIFY=1 in the environment): compare, in wait(), checksums of
    what all ranks sent and received per column; a mismatch raises.
    """
    import os
    if verify is None:
        verify = os.environ.get("TEMPO_SHUFFLE_VERIFY") == "1"
    world = dist.get_world_size(group)
    sliced = owners is not None and owners.slices is not None
    if sliced and ts is None:
        raise ValueError("an owner plan with time slices needs the table's ts column")
    table = None if owners is None else owners.table
    slices = owners.slices if sliced else None
    cols = [key] + list(columns)
    routes = [False, True] if (sliced and halo) else [False]
    parts = []  # per route: (rows per destination, the columns in bucketed order)
    fused = partition_fn is None and take_fn is None and world <= 64 and len(cols) <= 12
    partition_fn = partition_fn or _hip_partition
    take_fn = take_fn or _hip_take
    for copies in routes:
        if fused:    # one device call routes the rows and moves every column
            staged, counts = ops.partition_move(key, cols, world, table, slices, ts if sliced else None, copies)
        else:
            if owners is None:
                perm, counts = partition_fn(key, world)
            elif sliced:
                perm, counts = partition_fn(key, world, table, slices, ts, copies)
            else:
                perm, counts = partition_fn(key, world, table)
            if copies:
                perm = perm[:int(counts.sum())]
            staged = [take_fn(c, perm) for c in cols]
        if wire is not None:   # narrow the bucketed columns for the trip (elementwise: order does not matter)
            staged = wire.encode(staged[0], staged[1:])
        parts.append((counts, staged))
    dtypes = [c.dtype for c in columns]
    if wire is not None:
        cols = parts[0][1]   # the wire columns: what is exchanged from here on
    send = torch.stack([c.to(torch.int64) for c, _ in parts])          # [1 or 2][world]
    recv = torch.empty_like(send)
    for i in range(len(parts)):
        _counts_all_to_all(recv[i], send[i], group)
    # every rank must run the same number of rounds: size them by the largest pairwise message
    biggest = torch.maximum(send.max(dim=1).values, recv.max(dim=1).values)
    all_reduce(biggest, dist.ReduceOp.MAX, group)
    send_l, recv_l, biggest = send.cpu().tolist(), recv.cpu().tolist(), biggest.cpu().tolist()
    n_recv = [sum(r) for r in recv_l]
    works, staged_all, out_cols = [], [], []
    for ci, col in enumerate(cols):
        got = torch.empty(sum(n_recv), dtype=col.dtype, device=col.device)
        # a round's call carries at most rows_per_round rows to each of `world` peers
        rows_per_round = max(1, MAX_MESSAGE_BYTES // (col.element_size() * world))
        at = 0
        for i, (_, staged_cols) in enumerate(parts):
            if int(biggest[i]) == 0:  # nothing to move anywhere (all ranks agree on that)
                continue
            staged = staged_cols[ci]
            rounds = max(1, -(-int(biggest[i]) // rows_per_round))
            works += _exchange(staged, got[at:at + n_recv[i]], send_l[i], recv_l[i], rounds, rows_per_round, group)
            staged_all.append(staged)
            at += n_recv[i]
        out_cols.append(got)
    total_send = [sum(v) for v in zip(*send_l)]
    total_recv = [sum(v) for v in zip(*recv_l)]
    sent_sums = None
    if verify:
        sent_sums = torch.stack([sum((_checksum(staged_cols[ci]) for _, staged_cols in parts),
                                     torch.zeros((), dtype=torch.int64, device=key.device))
                                 for ci in range(len(cols))])
    return PendingShuffle(works, staged_all, out_cols, total_send, total_recv, wire, dtypes, sent_sums, group)


def shuffle_by_key(key, columns, group=None, partition_fn=None, take_fn=None, owners=None, ts=None,
                   halo=False, wire=None, verify=None):
    """Send every row to the rank that owns its key (blocking form of shuffle_by_key_start).
    Returns (key_local, columns_local, send_counts, recv_counts)."""
    return shuffle_by_key_start(key, columns, group, partition_fn, take_fn, owners, ts, halo, wire, verify).wait()
