import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from bournemouth_forced_aligner_amd import AlignmentUtils
dev = torch.device("cuda", 0)
au = AlignmentUtils(66, 0, silence_anchors=10)
for B, T, S in ((1, 3000, 120), (1, 1500, 60), (1, 3000, 180), (16, 3000, 180), (512, 3000, 180)):
    lp, toks = bench.synth_batch(B, T, S, 67, 7, dev)
    Tl = torch.full((B,), T, dtype=torch.int32, device=dev); Sl = torch.full((B,), S, dtype=torch.int32, device=dev)
    hint = au.viterbi_decoder.class_mask_hint([T]*B, [S]*B, has_sil=False, n_classes=67)
    fn = lambda: au.viterbi_decoder.align_batch(lp, toks, Tl, Sl, class_mask=hint)
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    print(f"B={B} T={T} S={S} hint={hint:#x}: {(time.perf_counter()-t0)/50*1e6:.1f} us/call", flush=True)
