"""-m gpu: the packed result records (bfa_pack_results / bfa_index_records, include/bfa.h ABI v5) against the host
restatement of the layout, and BASELINE.json configs[3] at its REAL size (32 768 utterances, one call) -- status, tuple
counts and monotonic tuples on every utterance, a stratified 512-utterance sample against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _records(rng, n, cap, with_conf=True):
    segs = rng.integers(0, 5000, size=(n, cap, 4)).astype(np.int32)
    cnt = rng.integers(0, cap + 1, size=n).astype(np.int32)
    if n > 3:
        cnt[rng.integers(0, n, size=max(1, n // 5))] = 0      # empty utterances sit between the others
    conf = rng.random((n, cap)).astype(np.float32) if with_conf else None
    return segs, cnt, conf


@pytest.mark.parametrize("n,cap", [(1, 1), (5, 3), (63, 7), (64, 42), (65, 42), (257, 9), (4096, 42), (1000, 122)])
def test_pack_results_matches_the_host_layout(gpu_device, n, cap):
    """k_pack == sharding.pack_results_host word for word (header, the three tables, tuples, confidences) for shard sizes
    around the 64-utterance workgroups, with and without confidences / explicit global indices / slack in the bounds."""
    from bournemouth_forced_aligner_amd.sharding import pack_layout, pack_results, pack_results_host
    dev = gpu_device
    rng = np.random.default_rng(100 * n + cap)
    for with_conf, with_gidx, slack in ((True, True, 0), (False, False, 0), (True, False, 37)):
        segs, cnt, conf = _records(rng, n, cap, with_conf)
        gidx = rng.permutation(10 * n)[:n].astype(np.int32) if with_gidx else None
        n_cap, tuple_cap = n + slack, int(cnt.sum()) + slack
        want = pack_results_host(segs, cnt, conf, gidx, n_cap, tuple_cap, gidx_base=7)
        got = pack_results(torch.from_numpy(segs).to(dev), torch.from_numpy(cnt).to(dev),
                           torch.from_numpy(conf).to(dev) if with_conf else None,
                           torch.from_numpy(gidx).to(dev) if with_gidx else None, n_cap, tuple_cap, gidx_base=7)
        got = got.cpu().numpy()
        lay = pack_layout(n_cap, tuple_cap, with_conf)
        assert got.shape[0] == lay["words"] == want.shape[0]
        total = int(cnt.sum())
        assert list(got[:8]) == list(want[:8]) and got[1] == total and got[5] == 0
        for key in ("gidx", "count", "offset"):
            assert np.array_equal(got[lay[key]:lay[key] + n_cap], want[lay[key]:lay[key] + n_cap]), key
        assert np.array_equal(got[lay["tuples"]:lay["tuples"] + 4 * total], want[lay["tuples"]:lay["tuples"] + 4 * total])
        if with_conf:
            assert np.array_equal(got[lay["conf"]:lay["conf"] + total], want[lay["conf"]:lay["conf"] + total])


def test_pack_results_overflow_is_reported_not_written(gpu_device):
    """a tuple bound smaller than the shard's tuples: the record is cut at the bound, word 5 says so, nothing is written
    beyond the tuple section"""
    from bournemouth_forced_aligner_amd.sharding import pack_layout, pack_results
    rng = np.random.default_rng(4)
    segs, cnt, conf = _records(rng, 300, 11)
    total = int(cnt.sum())
    bound = total - 50
    lay = pack_layout(300, bound, True)
    out = torch.full((lay["words"] + 64,), -77, dtype=torch.int32, device=gpu_device)
    pack_results(torch.from_numpy(segs).to(gpu_device), torch.from_numpy(cnt).to(gpu_device),
                 torch.from_numpy(conf).to(gpu_device), None, 300, bound, out=out)
    got = out.cpu().numpy()
    assert got[1] == bound and got[5] == 1 and (got[lay["words"]:] == -77).all()
    valid = np.arange(11)[None, :] < cnt[:, None]
    assert np.array_equal(got[lay["tuples"]:lay["tuples"] + 4 * bound], segs[valid][:bound].reshape(-1))


def test_gathered_records_index_and_views(gpu_device):
    """bfa_index_records over several records side by side (what a gather leaves on the destination): owner / offset /
    count per global index, rows(), to_padded() and to_lists() in the original order; the device index equals the host one."""
    from bournemouth_forced_aligner_amd.sharding import GatheredRecords, pack_results
    dev = gpu_device
    rng = np.random.default_rng(12)
    world, cap, n_total = 5, 9, 1000
    owner_of = rng.integers(0, world, size=n_total)
    owner_of[:3] = [4, 4, 0]
    segs, cnt, conf = _records(rng, n_total, cap)
    shards = [np.flatnonzero(owner_of == r) for r in range(world)]
    for r in range(world):
        rng.shuffle(shards[r])
    n_cap = max(len(s) for s in shards)
    tuple_cap = max(int(cnt[s].sum()) for s in shards)
    recs = [pack_results(torch.from_numpy(segs[s]).to(dev), torch.from_numpy(cnt[s]).to(dev), torch.from_numpy(conf[s]).to(dev),
                         torch.from_numpy(s.astype(np.int32)).to(dev), n_cap, tuple_cap) for s in shards]
    g = GatheredRecords(torch.stack(recs), n_total)
    owner, offset, count = (x.cpu().numpy() for x in g.index())
    assert np.array_equal(owner, owner_of) and np.array_equal(count, cnt)
    h = GatheredRecords(torch.stack(recs).cpu(), n_total)       # the host index of the same records
    assert all(np.array_equal(a.numpy(), b) for a, b in zip(h.index(), (owner, offset, count)))
    ps, pc, pf = g.to_padded(cap)
    keep = np.arange(cap)[None, :] < cnt[:, None]
    assert np.array_equal(pc.cpu().numpy(), cnt)
    assert np.array_equal(ps.cpu().numpy(), np.where(keep[:, :, None], segs, 0))
    assert np.array_equal(pf.cpu().numpy(), np.where(keep, conf, 0))
    lists = g.to_lists()
    for i in (0, 1, 2, 17, 500, n_total - 1):
        t, cf = g.rows(i)
        assert np.array_equal(t, segs[i, :cnt[i]]) and np.array_equal(cf, conf[i, :cnt[i]])
        assert lists[i] == [tuple(x) for x in segs[i, :cnt[i]].tolist()]


def test_c4_at_full_size_one_call(ora, gpu_device):
    """BASELINE.json configs[3] at its real size: the 32 768 mixed-length utterances (T ~ U[200, 3000], S = T // 25,
    seed 1004; 52.5 M frames, 14 GB of posteriors) synthesised on the device and aligned as ONE bfa_align_batch call.
    EVERY utterance: status OK, at most one tuple per token (exactly one below the sentinel regime), tuples ordered and inside [0, T).  A stratified 512-utterance
    sample (every length stratum + the 32 longest) is re-synthesised from the global indices and compared with the oracle;
    the tuples are read from the PACKED record (bfa_pack_results), so the exchange format sees the full batch too."""
    sys.path.insert(0, ROOT)
    from tools.synth import c4_lengths, c4_utterances
    from bournemouth_forced_aligner_amd import AlignmentUtils
    from bournemouth_forced_aligner_amd.sharding import GatheredRecords, pack_results
    dev = gpu_device
    C, seed, n_total = 67, 1004, 32768
    free, _ = torch.cuda.mem_get_info(dev)
    T, S = c4_lengths(n_total, seed)
    Tmax, Smax = int(T.max()), int(S.max())
    need = n_total * Tmax * C * 4
    assert free > need * 1.5, f"the GPU box has {free >> 30} GiB free; the full C4 batch needs {need >> 30} GiB padded"
    order = np.argsort(-T, kind="stable")                    # longest first, as bench.py's one-call plan does
    lp = torch.empty((n_total, Tmax, C), dtype=torch.float32, device=dev)
    tk = torch.empty((n_total, Smax), dtype=torch.int32, device=dev)
    for i in range(0, n_total, 512):
        sub = order[i:i + 512]
        a, b = c4_utterances(sub, T[sub], S[sub], C, seed, dev, Tpad=Tmax, Spad=Smax)
        lp[i:i + len(sub)] = a
        tk[i:i + len(sub)] = b
        del a, b
    To, So = T[order], S[order]
    au = AlignmentUtils(blank_id=C - 1, silence_id=0)
    cap = Smax + 2
    hint = au.viterbi_decoder.class_mask_hint(To, So, has_sil=False, n_classes=C)
    res = au.decode_alignments_device(lp, tk, torch.from_numpy(To.astype(np.int32)).to(dev),
                                      torch.from_numpy(So.astype(np.int32)).to(dev), class_mask=hint, seg_cap=cap)
    gidx = torch.from_numpy(order.astype(np.int32)).to(dev)
    rec = pack_results(res.segs, res.seg_count, None, gidx, n_total, int(np.minimum(S, cap).sum()))
    torch.cuda.synchronize()
    st = res.status.cpu().numpy()
    assert (st == 0).all(), f"status values {np.unique(st)}"
    cnt = res.seg_count.cpu().numpy()
    # planted posteriors: every token gets its frames -- until an utterance is long enough for its path score to reach the
    # reference's finite -1000 sentinel (beyond ~2000 frames here), after which the reference's own result loses tokens
    assert (cnt >= 1).all() and (cnt <= So).all() and np.array_equal(cnt[To <= 1500], So[To <= 1500])
    g = GatheredRecords(rec.unsqueeze(0), n_total)
    assert not g.overflowed()
    recs, owner, offset, count = g.host()
    assert (owner == 0).all() and np.array_equal(count[order], cnt)
    lay = g.layout()[0]
    tup = recs[0, lay["tuples"]:lay["tuples"] + 4 * int(count.sum())].reshape(-1, 4)
    first = np.zeros(tup.shape[0], bool)
    first[offset] = True
    assert (tup[:, 1] < tup[:, 2]).all() and (tup[:, 1] >= 0).all()
    assert (tup[1:, 1][~first[1:]] >= tup[:-1, 2][~first[1:]]).all(), "tuples of an utterance overlap"
    last = offset + count - 1
    assert (tup[last, 2] <= T).all() and (tup[:, 3] >= 0).all() and (tup[:, 3] < np.repeat(So, cnt)).all()
    short = T <= 1500
    assert (tup[offset[short], 3] == 0).all() and (tup[last[short], 3] == S[short] - 1).all()
    del lp, tk, res
    # the oracle on a stratified sample, re-synthesised from the global indices alone
    by_len = np.argsort(T, kind="stable")
    sample = np.unique(np.concatenate([by_len[-32:], by_len[np.linspace(0, n_total - 1, 480).astype(np.int64)]]))
    prm = ora.make_params(C - 1, 0)
    mism = 0
    for i in range(0, len(sample), 64):
        sub = sample[i:i + 64]
        a, b = c4_utterances(sub, T[sub], S[sub], C, seed, dev)
        exp = ora.decode_alignments(a.cpu().numpy(), b.cpu().numpy(), T[sub], S[sub], prm, seg_cap=cap)
        for k, gi in enumerate(sub):
            c = int(exp["seg_count"][k])
            rows, _ = g.rows(int(gi))
            mism += int(rows.shape[0] != c or not (rows == exp["seg"][k, :c]).all())
    assert mism == 0 and len(sample) >= 500 and int(T[sample].max()) == Tmax


def test_to_lists_compact_and_full_records_agree(gpu_device):
    """AlignmentResult.to_lists(): the 8-byte tuple record (bfa_pack_results16: frame counts / ids that fit 16 bits) and the
    16-byte one (frame arrays of 65 536 frames or more) give the same lists as the padded arrays they were packed from,
    incl. target index -1, end == 65 535 and empty utterances."""
    from bournemouth_forced_aligner_amd.forced_alignment import AlignmentResult
    dev = gpu_device
    rng = np.random.default_rng(8)
    n, cap = 700, 33
    segs = np.stack([rng.integers(0, 128, size=(n, cap)), rng.integers(0, 65535, size=(n, cap)),
                     rng.integers(0, 65536, size=(n, cap)), rng.integers(-1, 32767, size=(n, cap))], axis=2).astype(np.int32)
    segs[3, 0] = [127, 0, 65535, -1]
    cnt = rng.integers(0, cap + 1, size=n).astype(np.int32)
    cnt[3] = 5
    cnt[10:20] = 0
    want = [[tuple(int(v) for v in segs[b, k]) for k in range(cnt[b])] for b in range(n)]
    for Tmax in (64, 70000):   # the second: frame arrays too long for 16-bit frame numbers -> the 16-byte record
        res = AlignmentResult(torch.from_numpy(segs).to(dev), torch.from_numpy(cnt).to(dev), torch.zeros(n, dtype=torch.int32, device=dev),
                              None, torch.zeros((n, Tmax), dtype=torch.int32, device=dev), None, None, None)
        got = res.to_lists(check_status=True)
        assert got.tolist() == want, Tmax
        assert got[3][0] == (127, 0, 65535, -1) and got[10] == [] and len(got) == n
