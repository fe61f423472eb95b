"""Multi-GPU layout of the alignment path: one process per GPU, utterances sharded across ranks.

Utterances are independent (the reference's batch loop carries no cross-item state,
forced_alignment.py:885-905), so the data path has NO collective: every rank aligns its own shard
from its own device-resident posteriors.  The only exchange is the final gather of the small result
records (<= S tuples of 4 int32 + 1 float32 per utterance) to one rank -- torch.distributed
`gather` (RCCL over xGMI with backend "nccl"; "gloo" in the CPU tests).
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


def gather_results(segs, seg_count, conf, global_index, n_total, dst=0, group=None):
    """Final gather of per-rank results to rank `dst`.

    segs [n_local, cap, 4] int32, seg_count [n_local] int32, conf [n_local, cap] float32 (or None),
    global_index [n_local] int64: position of each local utterance in the original batch.
    Shards may have different sizes: every rank pads to the largest shard.  Returns, on `dst`, tensors
    of n_total utterances in the original order (None on the other ranks)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = segs.device
    n_local = torch.tensor([segs.shape[0], segs.shape[1]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    n_max = max(int(s[0]) for s in sizes)
    cap = max(int(s[1]) for s in sizes)
    has_conf = conf is not None

    def pad(x, shape, fill):
        out = torch.full(shape, fill, dtype=x.dtype, device=dev)
        out[tuple(slice(0, d) for d in x.shape)] = x
        return out

    # one packed int32 record per utterance: [global_index, count, cap x 4 segment ints, cap conf bits]
    width = 2 + 4 * cap + (cap if has_conf else 0)
    rec = torch.zeros((n_max, width), dtype=torch.int32, device=dev)
    n = segs.shape[0]
    rec[:, 0] = -1
    rec[:n, 0] = global_index.to(device=dev, dtype=torch.int32)
    rec[:n, 1] = seg_count.to(torch.int32)
    rec[:n, 2:2 + 4 * cap] = pad(segs.to(torch.int32), (n, cap, 4), 0).reshape(n, 4 * cap)
    if has_conf:
        rec[:n, 2 + 4 * cap:] = pad(conf.to(torch.float32), (n, cap), 0.0).view(torch.int32)
    bufs = [torch.empty_like(rec) for _ in range(world)] if rank == dst else None
    dist.gather(rec, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    allrec = torch.cat(bufs, dim=0)
    allrec = allrec[allrec[:, 0] >= 0]
    order = torch.argsort(allrec[:, 0].to(torch.int64))
    allrec = allrec[order]
    assert allrec.shape[0] == n_total, "gathered utterance count does not match"
    out_segs = allrec[:, 2:2 + 4 * cap].reshape(n_total, cap, 4).contiguous()
    out_cnt = allrec[:, 1].contiguous()
    out_conf = allrec[:, 2 + 4 * cap:].contiguous().view(torch.float32) if has_conf else None
    return out_segs, out_cnt, out_conf
