"""ensure_target_coverage with ensure_completeness=True (core.py:462-679), host side.

The default (ensure_completeness=False) runs on the device inside bfa_postprocess.  The completing variant is
rare-case repair logic over a handful of rows per utterance -- sequential, data dependent, full of Python
rounding -- so it stays on the host, between the device alignment and the device soft-boundary / confidence
passes.  Rows are (id, start_frame, end_frame, target_idx[, is_estimated]).

    1. rows whose target index is -1 or >= len(targets) are dropped (:509-513);
    2. a target aligned more than once keeps ONE row: its rows are merged where they touch or overlap, the
       longest merged span wins, the earliest on ties; it is re-appended after the untouched rows (:516-540);
    3. every run of consecutive missing targets gets estimated rows (is_estimated=True):
         between two aligned neighbours -> the gap between them, shared out evenly (at least one frame each);
         before the first aligned target  -> the frames just before it;
         nothing aligned at all           -> frames 0..n;
         after the last aligned target    -> trailing silences are skipped, for the others room is made by
                                             shortening silences (then anything), last row first, and shifting
                                             the rows behind them; each gets one frame after the last row;
       all clamped to the alignment's original extent (the largest end frame before any change);
    4. stable sort by start frame, `False` appended to rows that were really aligned (:659-666);
    5. every target except the skipped trailing silences must now be covered exactly once (:668-677).
"""


def _merge_longest(rows):
    """:527-538 for the rows of one repeated target."""
    rows = sorted(rows, key=lambda r: r[1])
    spans = [list(rows[0][:4])]
    for r in rows[1:]:
        if r[1] <= spans[-1][2]:
            spans[-1][2] = max(spans[-1][2], r[2])
        else:
            spans.append(list(r[:4]))
    return tuple(max(spans, key=lambda s: s[2] - s[1]))  # first maximum = earliest


def _shorten(rows, k, give):
    """row k ends `give` frames earlier and everything behind it moves up (:601-605)."""
    r = rows[k]
    rows[k] = (r[0], r[1], r[2] - give) + tuple(r[3:])
    for j in range(k + 1, len(rows)):
        q = rows[j]
        rows[j] = (q[0], q[1] - give, q[2] - give) + tuple(q[3:])


def complete_target_coverage(targets, rows, silence_class=0):
    """One utterance.  `targets`: the unpadded target ids; `rows`: aligned 4-tuples.  Returns the completed,
    start-sorted 5-tuples.  Raises Exception on a coverage mismatch like the reference (:676-677)."""
    targets = list(targets)
    n_t = len(targets)
    rows = [tuple(r) for r in rows]
    extent = max((r[2] for r in rows), default=0)
    found = [0] * n_t
    invalid = set()
    for r in rows:
        ti = int(r[3])
        if ti < n_t and ti != -1:
            found[ti] += 1  # (a target index below -1 counts from the end, as in the reference)
        else:
            invalid.add(ti)
    repeated = {i for i, c in enumerate(found) if c > 1}
    missing = [i for i, c in enumerate(found) if c == 0]
    if invalid:
        rows = [r for r in rows if int(r[3]) not in invalid]
    if repeated:
        again = {}
        kept = []
        for r in rows:
            if int(r[3]) in repeated:
                again.setdefault(int(r[3]), []).append(r)
            else:
                kept.append(r)
        rows = kept + [_merge_longest(again[ti]) for ti in sorted(again)]
    skipped_sil = set()
    if missing:
        by_target = {int(r[3]): r for r in rows}
        runs = [[missing[0]]]
        for ti in missing[1:]:
            if ti == runs[-1][-1] + 1:
                runs[-1].append(ti)
            else:
                runs.append([ti])
        for run in runs:
            before = next((by_target[t] for t in range(run[0] - 1, -1, -1) if t in by_target), None)
            after = next((by_target[t] for t in range(run[-1] + 1, n_t) if t in by_target), None)
            if before and after:
                g0, g1 = before[2], after[1]
            elif before:
                skipped_sil.update(t for t in run if targets[t] == silence_class)
                wanted = [t for t in run if targets[t] != silence_class]
                if not wanted:
                    continue
                need = len(wanted)
                rows = sorted(rows, key=lambda r: r[1])
                freed = 0
                for only_silence in (True, False):
                    for k in range(len(rows) - 1, -1, -1):
                        if freed >= need:
                            break
                        span = rows[k][2] - rows[k][1]
                        if span > 1 and (rows[k][0] == silence_class or not only_silence):
                            give = min(span - 1, need - freed)
                            _shorten(rows, k, give)
                            freed += give
                by_target = {int(r[3]): r for r in rows}
                tail = max(r[2] for r in rows)
                for i, t in enumerate(wanted):
                    s = min(tail + i, extent - 1)
                    new = (targets[t], s, min(s + 1, extent), t, True)
                    rows.append(new)
                    by_target[t] = new
                continue
            elif after:
                g1 = after[1]
                g0 = max(0, g1 - len(run))
            else:
                g0, g1 = 0, len(run)
            per = max(g1 - g0, len(run)) / len(run)
            for i, t in enumerate(run):
                s = min(int(g0 + i * per), extent - 1)
                e = min(max(int(g0 + (i + 1) * per), s + 1), extent)
                new = (targets[t], s, e, t, True)
                rows.append(new)
                by_target[t] = new
    rows.sort(key=lambda r: r[1])
    rows = [r if len(r) != 4 else r + (False,) for r in rows]
    expected = n_t - len(skipped_sil)
    covered = [0] * n_t
    for r in rows:
        ti = int(r[3])
        if ti < n_t and ti != -1:
            covered[ti] += 1
    if sum(covered) != expected or len(rows) != expected:
        raise Exception(f"Post-processing error: target coverage mismatch. Expected {expected}, got {sum(covered)} "
                        f"covered, {len(rows)} aligned. Skipped SIL: {skipped_sil}")
    return rows


def ensure_target_coverage(phoneme_sequences, aligned_frames, seq_lens=None, silence_class=0):
    """core.py:462 for a batch with ensure_completeness=True; returns new lists."""
    out = []
    for b, rows in enumerate(aligned_frames):
        seq = phoneme_sequences[b]
        seq = seq.tolist() if hasattr(seq, "tolist") else list(seq)
        n = int(seq_lens[b]) if seq_lens is not None else len(seq)
        out.append(complete_target_coverage(seq[:n], rows, silence_class))
    return out
