"""Host (numpy) statement of the shuffle's routing rules -- TEST INFRASTRUCTURE: what
tmp_hash_partition / tmp_key_counts compute on the device, for the CPU (gloo) tests and as
the expected value of the GPU tests."""
import numpy as np
import torch

from tempo_amd import _lib


def _t(x):
    return x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def bracket_index(ts, v, ts_is_seconds):
    d = ts.astype(np.float64) if ts_is_seconds else ts.astype(np.float64) / 1000000.0
    q = np.trunc(d / np.float64(v))
    return d, np.clip(q, -2147483648.0, 2147483647.0).astype(np.int64)


def owners_of(key, n_parts, table=None, slices=None, ts=None, halo=False):
    """Owner per row; in halo mode the part that needs a copy of the row, n_parts for none."""
    key = _t(key).astype(np.int64)
    if table is None:
        hashed = np.array([_lib.lib().tmp_key_owner(int(k), n_parts) for k in range(int(key.max()) + 1 if len(key) else 0)],
                          dtype=np.int64)
        return hashed[key] if len(key) else key
    table = _t(table).astype(np.int64)
    t = table[key]
    own = np.clip(t, 0, n_parts - 1)
    if slices is None:
        return own
    ts = _t(ts)
    v = slices["ts_partition_val"]
    d, q = bracket_index(ts, v, slices["ts_is_seconds"])
    S_all, first, sl_owner = _t(slices["split_slices"]).astype(np.int64), _t(slices["split_first"]).astype(np.int64), \
        _t(slices["slice_owner"]).astype(np.int64)
    sliced = t < 0
    slot = np.where(sliced, -(t + 1), 0)
    S = S_all[slot]

    def owner_at(qq):
        sl = np.floor_divide((qq - slices["bracket_min"]) * S, slices["bracket_count"])
        sl = np.clip(sl, 0, S - 1)
        return np.clip(sl_owner[first[slot] + sl], 0, n_parts - 1)

    o1 = np.where(sliced, owner_at(q), own)
    if not halo:
        return o1
    if slices["v_is_int"]:
        part = ((np.int64(int(v)) * q + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.float64)
    else:
        part = np.float64(v) * q.astype(np.float64)
    tail = (d - part) / np.float64(v) >= slices["one_minus_fraction"]
    o2 = owner_at(q + 1)
    return np.where(sliced & tail & (o2 != o1), o2, n_parts)


def host_partition(key, n_parts, table=None, slices=None, ts=None, halo=False):
    owner = owners_of(key, n_parts, table, slices, ts, halo)
    perm = np.argsort(owner, kind="stable")
    counts = np.bincount(owner, minlength=n_parts + 1)[:n_parts]
    return torch.from_numpy(perm.astype(np.int32)), torch.from_numpy(counts.astype(np.int64))


def host_take(col, perm):
    return col[perm.to(torch.int64)]


def host_counts(key, n_keys, stride):
    return torch.from_numpy(np.bincount(_t(key)[::stride], minlength=n_keys).astype(np.int64))
