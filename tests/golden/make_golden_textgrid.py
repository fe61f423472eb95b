#!/usr/bin/env python3
"""tests/golden/textgrid/: LJ001-000{1,2}.vs.json / .TextGrid are the reference's own example pairs
(examples/samples/LJSpeech, data files).  This script adds the expected text of the with-confidence variant
(utils.py:280-411), produced by the reference's writer loaded by path.  Run in the build container."""
import importlib.util
import json
import os

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "textgrid")
spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/bournemouth_aligner/utils.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
out = {}
for stem in ("LJ001-0001", "LJ001-0002"):
    d = json.load(open(os.path.join(HERE, stem + ".vs.json")))
    assert mod.dict_to_textgrid(d) == open(os.path.join(HERE, stem + ".TextGrid"), encoding="utf-8").read()
    out[stem] = mod.dict_to_textgrid(d, include_confidence=True)
    # a variant without words / groups tiers and one without any tier
    d2 = {"segments": [dict(d["segments"][0])]}
    d2["segments"][0].pop("words_ts", None)
    out[stem + ":no_words"] = mod.dict_to_textgrid(d2)
    out[stem + ":no_words:conf"] = mod.dict_to_textgrid(d2, include_confidence=True)
d3 = {"segments": [{"start": 0.0, "end": 2.5}]}
out["empty"] = mod.dict_to_textgrid(d3)
json.dump(out, open(os.path.join(HERE, "expected_variants.json"), "w"), ensure_ascii=False, indent=0)
print("wrote", len(out), "variants")
