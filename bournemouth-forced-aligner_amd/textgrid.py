"""Praat long-format TextGrid text for an alignment result dict (the reference's `dict_to_textgrid`,
bournemouth_aligner/utils.py:152-411).  Host-side text formatting, no device work; kept because the reference's
end-to-end parity check compares TextGrid files.  The exact text (tier order, number
formatting via Python's float repr, no trailing newline) is pinned by the reference's own example pairs under
tests/golden/textgrid/.

    tiers            source list     label field     order (plain)   order (with confidence)
    "phonemes"       phoneme_ts      ipa_label       1               2
    "words"          words_ts        word            2               1
    "groups"         group_ts        group_label     3               3
"""

_TIER_SOURCE = {"phonemes": ("phoneme_ts", "ipa_label"), "words": ("words_ts", "word"), "groups": ("group_ts", "group_label")}
_ORDER_PLAIN = ("phonemes", "words", "groups")      # utils.py:209-275
_ORDER_CONFIDENCE = ("words", "phonemes", "groups")  # utils.py:333-401


def _lines(segment, order, with_confidence):
    tiers = [(name, segment.get(_TIER_SOURCE[name][0], []) or [], _TIER_SOURCE[name][1]) for name in order]
    present = [t for t in tiers if t[1]]
    if present:   # utils.py:176-183
        ends = [max(item["end_ms"] for item in items) / 1000.0 for _, items, _ in present]
        xmax = max(ends + [0] * (3 - len(ends)) + [segment.get("end", 0)])
    else:
        xmax = segment.get("end", 1.0)
    yield 'File type = "ooTextFile"'
    yield 'Object class = "TextGrid"'
    yield ""
    yield "xmin = 0"
    yield f"xmax = {xmax}"
    yield "tiers? <exists>"
    yield f"size = {len(present)}"
    yield "item []:"
    for number, (name, items, label_key) in enumerate(present, 1):
        yield f"    item [{number}]:"
        yield '        class = "IntervalTier"'
        yield f'        name = "{name}"'
        yield "        xmin = 0"
        yield f"        xmax = {xmax}"
        yield f"        intervals: size = {len(items)}"
        for k, item in enumerate(items, 1):
            label = item[label_key]
            if with_confidence:
                label += " ({:.2f})".format(item.get("confidence", 0))
            yield f"        intervals [{k}]:"
            yield f"            xmin = {item['start_ms'] / 1000.0}"
            yield f"            xmax = {item['end_ms'] / 1000.0}"
            yield f'            text = "{label}"'


def dict_to_textgrid(data, output_file=None, include_confidence=False):
    """data: the dict `process_sentence` returns ({'segments': [{'phoneme_ts': [...], 'group_ts': [...],
    'words_ts': [...], 'end': s}, ...]}); only the first segment is written, like the reference.  Returns the text,
    or writes it to `output_file` (utf-8) and returns None."""
    segments = data["segments"]
    if not segments:
        raise ValueError("No segments found in data")
    order = _ORDER_CONFIDENCE if include_confidence else _ORDER_PLAIN
    text = "\n".join(_lines(segments[0], order, include_confidence))
    if output_file:
        with open(output_file, "w", encoding="utf-8") as f:
            f.write(text)
        print(f"TextGrid saved to {output_file}")
        return None
    return text


def dict_to_textgrid_with_confidence(data, output_file=None, include_confidence=True):
    """utils.py:280-411: the words / phonemes / groups order, labels optionally followed by ' (0.87)'."""
    segments = data["segments"]
    if not segments:
        raise ValueError("No segments found in data")
    text = "\n".join(_lines(segments[0], _ORDER_CONFIDENCE, include_confidence))
    if output_file:
        with open(output_file, "w", encoding="utf-8") as f:
            f.write(text)
        print(f"TextGrid saved to {output_file}")
        return None
    return text
