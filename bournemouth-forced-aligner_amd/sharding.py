"""Multi-GPU layout of the alignment path: one process per GPU, utterances sharded across ranks.

Utterances are independent (the reference's batch loop carries no cross-item state,
forced_alignment.py:885-905), so the data path has NO collective: every rank aligns its own shard
from its own device-resident posteriors.  The only exchange is the final gather of the result records to one
rank: every rank packs its tuples into ONE contiguous record (bfa_pack_results: per-utterance global index / count /
offset, then the valid tuples back to back -- 16 B per tuple, 20 B with a confidence) and ONE torch.distributed
`gather` moves the records (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests).  The receiver keeps them as
they arrive and looks utterances up through an index (bfa_index_records) -- nothing is padded, concatenated or sorted.
"""
import numpy as np
import torch


def utterance_cost(T, S):
    """DP cost model: frames x CTC states (stride-4 expansion, forced_alignment.py:153-157)."""
    T = np.asarray(T, np.int64)
    S = np.asarray(S, np.int64)
    return T * (4 * S + 1)


# What one rank's shard costs, for REPORTING only (bench.py prints it beside the measured time; the partition below does not
# use it).  MI355X, round 4 (profiles/r04_c4*.json, r04_mix_workgroup_timeline.txt): the pairs of a full machine advance at one
# shared pace per frame (~0.37 us: the SIMD's issue slots are shared out among its waves), so a call takes at least the
# frames of its longest utterance at that pace plus its walk, or the machine rate over all its frames, whichever is longer.
CHAIN_MS_PER_FRAME = 0.37e-3
CHAIN_MS_FIXED = 0.20
MACHINE_FRAMES_PER_MS = 7.0e6


def predict_rank_ms(T_lens, S_lens):
    """{"chain_ms", "work_ms", "predicted_ms"} of one rank's shard: max(chain of the longest utterance, frames / machine
    rate).  A model for reporting, not a guarantee; shard_utterances balances the DP work and nothing else."""
    T = np.asarray(T_lens, np.int64)
    if T.size == 0:
        return {"chain_ms": 0.0, "work_ms": 0.0, "predicted_ms": 0.0}
    chain = CHAIN_MS_FIXED + CHAIN_MS_PER_FRAME * float(T.max())
    work = float(T.sum()) / MACHINE_FRAMES_PER_MS
    return {"chain_ms": chain, "work_ms": work, "predicted_ms": max(chain, work)}


def shard_utterances(T_lens, S_lens, world_size):
    """Longest-processing-time-first assignment of utterances to ranks on the DP work (frames x CTC states).
    A rank's time is max(chain of its longest utterance, its work / machine rate) (predict_rank_ms): whatever the
    partition, SOME rank holds the batch's longest utterance and the chains of one rank run side by side, so the maximum
    over ranks is max(longest chain of the batch, largest work share) -- balancing the work is all a partition can do.
    Returns list[world_size] of index arrays (sorted ascending inside each shard)."""
    cost = utterance_cost(T_lens, S_lens)
    order = np.argsort(-cost, kind="stable")
    load = np.zeros(world_size, np.int64)
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += int(cost[i])
    return [np.array(sorted(s), np.int64) for s in shards]


PACK_HDR = 8  # include/bfa.h, bfa_pack_results: n, total, n_cap, tuple_cap, has_conf, overflow, 0, 0


def pack_layout(n_cap, tuple_cap, has_conf):
    """Word offsets of the sections of one packed record (include/bfa.h, bfa_pack_results; csrc/bfa_pack.hip pack_layout)."""
    n4 = (int(n_cap) + 3) & ~3
    lay = {"gidx": PACK_HDR, "count": PACK_HDR + n4, "offset": PACK_HDR + 2 * n4, "tuples": PACK_HDR + 3 * n4}
    lay["conf"] = lay["tuples"] + 4 * int(tuple_cap)
    lay["words"] = lay["conf"] + (((int(tuple_cap) + 3) & ~3) if has_conf else 0)
    return lay


def pack_results(segs, seg_count, conf=None, global_index=None, n_cap=None, tuple_cap=None, gidx_base=0, out=None):
    """The result records of a call as ONE contiguous int32 record (CSR: per-utterance global index / count / offset tables,
    then the valid tuples back to back, then their confidences), written by ONE kernel (bfa_pack_results) with no host
    knowledge of the counts -- what the final gather of a sharded batch exchanges and what the copy to the host behind
    decode_alignments' list of lists takes.  segs [n, cap, 4] int32, seg_count [n] int32, conf [n, cap] float32 or None,
    global_index [n] int32 (device) or None (= gidx_base + j).  n_cap / tuple_cap: the caller's bounds (defaults n and
    n * cap).  Device tensors only: the records are produced where the alignment left its tuples."""
    if not segs.is_cuda:
        raise RuntimeError("pack_results runs on the GPU (bfa_pack_results); host-side records are made by pack_results_host")
    import ctypes  # noqa: F401
    from . import _lib
    n, cap = int(segs.shape[0]), int(segs.shape[1])
    n_cap = n if n_cap is None else int(n_cap)
    tuple_cap = n * cap if tuple_cap is None else int(tuple_cap)
    L = _lib.lib()
    dev = segs.device
    words = int(L.bfa_pack_words(n_cap, tuple_cap, 1 if conf is not None else 0))
    if out is None:
        out = torch.empty((words,), dtype=torch.int32, device=dev)
    assert out.numel() >= words and out.dtype == torch.int32 and out.is_contiguous()
    assert segs.is_contiguous() and seg_count.is_contiguous() and segs.dtype == torch.int32 and seg_count.dtype == torch.int32
    if conf is not None:
        assert conf.is_contiguous() and conf.dtype == torch.float32 and tuple(conf.shape) == (n, cap)
    if global_index is not None:
        assert global_index.is_cuda and global_index.dtype == torch.int32 and global_index.is_contiguous()
    h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device())
    with torch.cuda.device(dev):
        rc = L.bfa_pack_results(h, segs.data_ptr(), cap, seg_count.data_ptr(), conf.data_ptr() if conf is not None else None,
                                global_index.data_ptr() if global_index is not None else None, int(gidx_base), n, n_cap,
                                tuple_cap, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, h, "bfa_pack_results")
    return out[:words]


def pack_results_host(segs, seg_count, conf=None, global_index=None, n_cap=None, tuple_cap=None, gidx_base=0):
    """The same record from HOST arrays (numpy): the plumbing of `bench.py --dry-run` and the CPU tests of the exchange,
    which fabricate records without a GPU.  Not an alignment path -- nothing here computes a tuple."""
    segs = np.ascontiguousarray(np.asarray(segs, np.int32))
    n, cap = segs.shape[0], segs.shape[1]
    cnt = np.clip(np.asarray(seg_count, np.int64), 0, cap)
    n_cap = n if n_cap is None else int(n_cap)
    tuple_cap = n * cap if tuple_cap is None else int(tuple_cap)
    lay = pack_layout(n_cap, tuple_cap, conf is not None)
    out = np.zeros(lay["words"], np.int32)
    off = np.concatenate([[0], np.cumsum(cnt)])
    total = int(off[-1])
    out[:6] = [n, min(total, tuple_cap), n_cap, tuple_cap, int(conf is not None), int(total > tuple_cap)]
    g = np.full(n_cap, -1, np.int32)
    g[:n] = (np.arange(n) + gidx_base) if global_index is None else np.asarray(global_index, np.int64)
    out[lay["gidx"]:lay["gidx"] + n_cap] = g
    out[lay["count"]:lay["count"] + n] = cnt
    out[lay["offset"]:lay["offset"] + n] = off[:-1]
    if n_cap > n:
        out[lay["offset"] + n:lay["offset"] + n_cap] = total
    valid = np.arange(cap)[None, :] < cnt[:, None]
    keep = min(total, tuple_cap)
    out[lay["tuples"]:lay["tuples"] + 4 * keep] = segs[valid][:keep].reshape(-1)
    if conf is not None:
        out[lay["conf"]:lay["conf"] + keep] = np.asarray(conf, np.float32)[valid][:keep].view(np.int32)
    return out


class GatheredRecords:
    """What rank `dst` holds after the final gather: the ranks' packed records side by side in ONE [world, words] int32 tensor
    (no reordering, no padding to per-utterance capacity).  `index()` -> (owner, offset, count) per GLOBAL utterance index --
    one kernel on the GPU (bfa_index_records), built when first asked for; `rows(g)` / `to_lists()` / `to_padded()` read
    through it."""

    def __init__(self, records, n_total):
        self.records = records
        self.n_total = int(n_total)
        self.world = int(records.shape[0])
        self.words = int(records.shape[1])
        self._index = None
        self._host = None

    def _meta(self):
        # n_cap / tuple_cap / has_conf are the same in every record (the ranks agreed on them): read them from the layout
        hdr = self.records[0, :PACK_HDR].cpu().numpy() if self._host is None else self._host[0][0, :PACK_HDR]
        return int(hdr[2]), int(hdr[3]), bool(hdr[4])

    def layout(self):
        n_cap, tuple_cap, has_conf = self._meta()
        return pack_layout(n_cap, tuple_cap, has_conf), n_cap, tuple_cap, has_conf

    def index(self):
        if self._index is not None:
            return self._index
        rec = self.records
        dev = rec.device
        owner = torch.full((self.n_total,), -1, dtype=torch.int32, device=dev)
        offset = torch.zeros((self.n_total,), dtype=torch.int32, device=dev)
        count = torch.zeros((self.n_total,), dtype=torch.int32, device=dev)
        if rec.is_cuda:
            from . import _lib
            L = _lib.lib()
            h = _lib.handle(dev.index if dev.index is not None else torch.cuda.current_device())
            lay, n_cap, _, _ = self.layout()
            with torch.cuda.device(dev):
                rc = L.bfa_index_records(h, rec.data_ptr(), self.world, self.words, n_cap, self.n_total, owner.data_ptr(),
                                         offset.data_ptr(), count.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(rc, h, "bfa_index_records")
        else:  # host records (gloo / dry run)
            lay, n_cap, _, _ = self.layout()
            r = rec.numpy()
            g = r[:, lay["gidx"]:lay["gidx"] + n_cap]
            ok = (g >= 0) & (g < self.n_total)
            rr = np.broadcast_to(np.arange(self.world, dtype=np.int32)[:, None], g.shape)
            owner.numpy()[g[ok]] = rr[ok]
            offset.numpy()[g[ok]] = r[:, lay["offset"]:lay["offset"] + n_cap][ok]
            count.numpy()[g[ok]] = r[:, lay["count"]:lay["count"] + n_cap][ok]
        self._index = (owner, offset, count)
        return self._index

    def overflowed(self):
        """True if any rank's tuples did not fit its tuple_cap (the record was cut)."""
        return bool((self.records[:, 5] != 0).any())

    def host(self):
        """(records, owner, offset, count) as numpy arrays: ONE copy of the records to the host."""
        if self._host is None:
            owner, offset, count = self.index()
            self._host = (self.records.cpu().numpy(), owner.cpu().numpy(), offset.cpu().numpy(), count.cpu().numpy())
        return self._host

    def rows(self, g):
        """utterance g's tuples [count, 4] int32 and confidences [count] float32 (or None), from the host copy"""
        rec, owner, offset, count = self.host()
        lay, _, _, has_conf = self.layout()
        r, o, c = int(owner[g]), int(offset[g]), int(count[g])
        if r < 0:
            raise KeyError(f"utterance {g} is in no gathered record")
        t = rec[r, lay["tuples"] + 4 * o:lay["tuples"] + 4 * (o + c)].reshape(c, 4)
        cf = rec[r, lay["conf"] + o:lay["conf"] + o + c].view(np.float32) if has_conf else None
        return t, cf

    def to_lists(self):
        """list[n_total] of list[(phoneme_id, start, end, target_seq_idx)] in the ORIGINAL utterance order"""
        return [[tuple(x) for x in self.rows(g)[0].tolist()] for g in range(self.n_total)]

    def to_padded(self, cap):
        """(segs [n_total, cap, 4], count [n_total], conf [n_total, cap] or None) in the original order: the padded arrays
        a single call would have returned (tests; device-agnostic torch indexing, not part of the exchange)."""
        owner, offset, count = self.index()
        lay, _, tuple_cap, has_conf = self.layout()
        dev = self.records.device
        k = torch.arange(cap, device=dev, dtype=torch.int64)[None, :]
        valid = k < count.long()[:, None]
        row = (offset.long()[:, None] + k).clamp(max=max(tuple_cap - 1, 0))
        own = owner.long().clamp(min=0)[:, None]
        tup = self.records[:, lay["tuples"]:lay["tuples"] + 4 * tuple_cap].reshape(self.world, tuple_cap, 4)
        segs = torch.where(valid[..., None], tup[own, row], torch.zeros((), dtype=torch.int32, device=dev))
        conf = None
        if has_conf:
            cf = self.records[:, lay["conf"]:lay["conf"] + tuple_cap].view(torch.float32)
            conf = torch.where(valid, cf[own, row], torch.zeros((), dtype=torch.float32, device=dev))
        return segs, count.clone(), conf


def gather_packed(record, n_total, dst=0, group=None, out=None):
    """The final exchange: every rank contributes ONE packed record of the same size (pack_results with the n_cap /
    tuple_cap all ranks agreed on); rank `dst` receives them side by side.  One `torch.distributed.gather` (RCCL over xGMI
    with backend "nccl", "gloo" in the CPU tests), nothing else: no size round trip, no reordering on the device.
    Returns GatheredRecords on `dst`, None elsewhere.  `out` [world, words] int32: optional receive buffer on `dst`."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    words = record.numel()
    if record.is_cuda and dist.get_backend(group) == "gloo":
        # A host-side process group under device-resident records (several ranks sharing ONE GPU: RCCL refuses two ranks on
        # a device, `bench.py --oversubscribe`): the record leaves the GPU through pinned memory and gloo gathers host tensors;
        # the receiver reads the records from the host (GatheredRecords' host path).  Same pack kernel, same record layout,
        # same single gather as over RCCL.
        host = torch.empty((words,), dtype=torch.int32, pin_memory=True)
        host.copy_(record, non_blocking=True)
        torch.cuda.current_stream(record.device).synchronize()
        record, out = host, None
    bufs = None
    if rank == dst:
        if out is None:
            out = torch.empty((world, words), dtype=torch.int32, device=record.device)
        bufs = [out[r] for r in range(world)]
    dist.gather(record, bufs, dst=dst, group=group)
    return GatheredRecords(out, n_total) if rank == dst else None


def agree_on_bounds(n_local, tuples_local, group=None, device=None):
    """(n_cap, tuple_cap) = the maxima over ranks of the shard sizes and of the tuple bounds, for callers whose ranks do NOT
    know each other's shapes (one all_gather of two integers and a host read; ranks that partitioned a batch they all
    know compute the same maxima from the partition and skip this)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([int(n_local), int(tuples_local)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    return max(int(s[0]) for s in sizes), max(int(s[1]) for s in sizes)


def gather_results(segs, seg_count, conf, global_index, n_total, dst=0, group=None, n_cap=None, tuple_cap=None):
    """Final gather of per-rank results to rank `dst`: pack (one kernel) + gather (one collective).

    segs [n_local, cap, 4] int32, seg_count [n_local] int32, conf [n_local, cap] float32 (or None), global_index
    [n_local]: position of each local utterance in the original batch.  n_cap / tuple_cap: the record bounds every rank
    uses (largest shard, largest number of tuples a shard can hold); when omitted the ranks agree on them with one
    all_gather of their own (n_local, n_local * cap).  Returns GatheredRecords on `dst` (None on the other ranks)."""
    n, cap = int(segs.shape[0]), int(segs.shape[1])
    if n_cap is None or tuple_cap is None:
        n_cap, tuple_cap = agree_on_bounds(n, n * cap, group, segs.device)
    if segs.is_cuda:
        gi = global_index
        if gi is not None and not (isinstance(gi, torch.Tensor) and gi.is_cuda and gi.dtype == torch.int32):
            gi = torch.as_tensor(np.asarray(gi.cpu() if isinstance(gi, torch.Tensor) else gi)).to(device=segs.device, dtype=torch.int32)
        rec = pack_results(segs.contiguous(), seg_count.to(torch.int32).contiguous(), conf, gi, n_cap, tuple_cap)
    else:
        gi = None if global_index is None else np.asarray(global_index)
        rec = torch.from_numpy(pack_results_host(segs.numpy(), seg_count.numpy(), None if conf is None else conf.numpy(), gi,
                                                 n_cap, tuple_cap))
    return gather_packed(rec, n_total, dst, group)
