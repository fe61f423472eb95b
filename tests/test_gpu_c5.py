"""-m gpu: BASELINE.json configs[4] harness (tools/c5_pipeline.py) as a dry run -- a directory of synthetic posterior
dumps with silences (targets with SIL, planted silent stretches -> the silence-anchored mode) through the device
pipeline to TextGrids, diffed against TextGrids built from the ORACLE's rows.  The real LJSpeech run is blocked on the
cupe2i checkpoint (not available offline), not on code."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def test_c5_directory_to_textgrids_against_oracle(ora, gpu_device, tmp_path):
    from tools import c5_pipeline as c5
    from test_oracle_golden import oracle_level2
    from bournemouth_forced_aligner_amd import PhonemeTimestampAligner
    from bournemouth_forced_aligner_amd.textgrid import dict_to_textgrid
    d_in, d_gpu, d_ora = str(tmp_path / "in"), str(tmp_path / "gpu"), str(tmp_path / "ora")
    assert c5.synth(d_in, n=64, seed=5) == 64
    assert c5.run(d_in, d_gpu, batch=24, device="cuda:0") == 64

    # the same batches through the oracle's stages (rows), then the same host-side shaping and writer
    os.makedirs(d_ora)
    host = PhonemeTimestampAligner(preset=None, device="cpu", group_id_to_label={i: f"g{i}" for i in range(17)},
                                   phoneme_id_to_label={i: f"p{i}" for i in range(67)})
    n_seg_mode = n_total = 0
    for us in c5.batches(c5.load_dir(d_in), 24):
        lc, lg, spec = c5.pad_batch(us)
        B = len(us)
        smax = max(len(u["ph66"]) for u in us)
        tk = np.full((B, smax), 66, np.int32)
        gk = np.full((B, smax), 16, np.int32)
        for b, u in enumerate(us):
            tk[b, :len(u["ph66"])] = u["ph66"]
            gk[b, :len(u["pg16"])] = u["pg16"]
        slens = np.array([len(u["ph66"]) for u in us], np.int32)
        wl = [u["wav_len"] for u in us]
        heads = {}
        for key, logits, toks, blank in (("phoneme_timestamps", lc.numpy(), tk, 66), ("group_timestamps", lg.numpy(), gk, 16)):
            heads[key] = oracle_level2(ora, logits, toks, slens, spec, wl, [0.0] * B, blank, 3)
        lp = np.stack([ora.log_softmax_rows(lc.numpy()[b]) for b in range(B)])
        modes = ora.decode_alignments(lp, tk, spec, slens, ora.make_params(66, 0))["mode"]
        n_seg_mode += int((modes == ora.MODE_SEGMENTED).sum())
        n_total += B
        for b, u in enumerate(us):
            rows = {}
            for key in heads:
                ri, cf, sm, em = heads[key][b]
                rows[key] = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), bool(r[4]), float(c), float(s), float(e))
                             for r, c, s, e in zip(ri, cf, sm, em)]
            d = c5.segment_dict(host, u, rows)
            with open(os.path.join(d_ora, u["name"] + ".TextGrid"), "w", encoding="utf-8") as f:
                f.write(dict_to_textgrid(d))
    n, bad = c5.diff_dirs(d_gpu, d_ora)
    assert n == 64 and not bad, bad[:3]
    assert n_seg_mode >= n_total // 3, f"only {n_seg_mode} of {n_total} utterances took the silence-anchored mode"
