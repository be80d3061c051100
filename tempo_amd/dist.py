"""Multi-GPU key shuffle: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Spark's Exchange hashpartitioning(partitionCols) in front of every Window / groupBy
(python/tempo/tsdf.py:121, :571; python/tempo/resample.py:63) becomes one all-to-all per
column: rows are bucketed on the device by owner = mix(key) % world_size
(tmp_hash_partition), exchanged with all_to_all, and every later operator runs locally on the
series a rank owns -- no other collective is on the data path.
"""
import torch
import torch.distributed as dist

from . import ops

# Largest single message handed to the collective, in bytes.  Columns of 10^9 rows are several
# GB per peer; they go out as a sequence of all-to-alls over row ranges of at most this size.
MAX_MESSAGE_BYTES = 1 << 30


def _hip_partition(key, n_parts):
    return ops.hash_partition(key, n_parts)


def _hip_take(col, perm):
    return ops.gather(col, perm, want_valid=False)[0]


def _exchange(staged, got, send_l, recv_l, rounds, rows_per_round, group):
    """All-to-all of one column in `rounds` slices: round c moves rows [c*R, (c+1)*R) of what
    every rank sends to every other rank.  Views only, nothing is copied.  Returns the pending
    work handles (the collectives are issued asynchronously, in order)."""
    s_off = [0]
    for v in send_l[:-1]:
        s_off.append(s_off[-1] + v)
    r_off = [0]
    for v in recv_l[:-1]:
        r_off.append(r_off[-1] + v)
    if rounds == 1:
        return [dist.all_to_all_single(got, staged, output_split_sizes=recv_l, input_split_sizes=send_l,
                                       group=group, async_op=True)]
    works = []
    for c in range(rounds):
        lo = c * rows_per_round
        ins = [staged[s_off[r] + min(lo, send_l[r]): s_off[r] + min(lo + rows_per_round, send_l[r])]
               for r in range(len(send_l))]
        outs = [got[r_off[r] + min(lo, recv_l[r]): r_off[r] + min(lo + rows_per_round, recv_l[r])]
                for r in range(len(recv_l))]
        works.append(dist.all_to_all(outs, ins, group=group, async_op=True))
    return works


class PendingShuffle:
    """A key shuffle whose column exchanges are in flight.  wait() makes the current stream wait
    for them and returns (key_local, columns_local, send_counts, recv_counts).  The staged send
    buffers and the receive buffers stay referenced here until then."""

    def __init__(self, works, staged, received, send_l, recv_l):
        self._works, self._staged, self._received = works, staged, received
        self.send_counts, self.recv_counts = send_l, recv_l

    def wait(self):
        for w in self._works:
            w.wait()
        self._works, self._staged = [], None
        return self._received[0], self._received[1:], self.send_counts, self.recv_counts


def shuffle_by_key_start(key, columns, group=None, partition_fn=None, take_fn=None):
    """Start sending every row to the rank that owns its key; returns a PendingShuffle.

    The device bucketing and the count exchange happen here (one small blocking read); the
    column exchanges are issued asynchronously, so work on other tables can be queued behind
    this call and overlaps with the transfers.
    key: int32 tensor of GLOBAL dictionary ids (the same id for the same key on every rank).
    columns: list of 1-D tensors with one entry per row (1-, 4- or 8-byte element types).
    partition_fn / take_fn default to the HIP kernels; tests on CPU inject host equivalents.
    """
    world = dist.get_world_size(group)
    partition_fn = partition_fn or _hip_partition
    take_fn = take_fn or _hip_take
    perm, counts = partition_fn(key, world)
    send = counts.to(torch.int64)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    # every rank must run the same number of rounds: size them by the largest pairwise message
    biggest = torch.maximum(send.max(), recv.max()).reshape(1)
    dist.all_reduce(biggest, op=dist.ReduceOp.MAX, group=group)
    send_l = [int(v) for v in send.cpu().tolist()]
    recv_l = [int(v) for v in recv.cpu().tolist()]
    biggest = int(biggest.item())
    n_recv = sum(recv_l)
    works, staged_all, out_cols = [], [], []
    for col in [key] + list(columns):
        staged = take_fn(col, perm)
        got = torch.empty(n_recv, dtype=col.dtype, device=col.device)
        rows_per_round = max(1, MAX_MESSAGE_BYTES // col.element_size())
        rounds = max(1, -(-biggest // rows_per_round))
        works += _exchange(staged, got, send_l, recv_l, rounds, rows_per_round, group)
        staged_all.append(staged)
        out_cols.append(got)
    return PendingShuffle(works, staged_all, out_cols, send_l, recv_l)


def shuffle_by_key(key, columns, group=None, partition_fn=None, take_fn=None):
    """Send every row to the rank that owns its key (blocking form of shuffle_by_key_start).
    Returns (key_local, columns_local, send_counts, recv_counts)."""
    return shuffle_by_key_start(key, columns, group, partition_fn, take_fn).wait()
