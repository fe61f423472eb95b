import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from bournemouth_forced_aligner_amd import AlignmentUtils
dev = torch.device("cuda", 0)
au = AlignmentUtils(66, 0, silence_anchors=10)
for B, T, S in ((1, 1000, 40), (256, 600, 20), (1, 600, 20)):
    lp, toks = bench.synth_batch(B, T, S, 67, 7, dev)
    Tl = torch.full((B,), T, dtype=torch.int32, device=dev); Sl = torch.full((B,), S, dtype=torch.int32, device=dev)
    hint = au.viterbi_decoder.class_mask_hint([T]*B, [S]*B, has_sil=False, n_classes=67)
    vd = au.viterbi_decoder
    rows = []
    for _ in range(8):
        vd.align_batch(lp, toks, Tl, Sl, class_mask=hint); torch.cuda.synchronize()
        w = vd._ws.buf.view(torch.int32).cpu().numpy()
        k = np.nonzero(w == 0x5eedbeef)[0]
        if k.size == 0: print("no stamp"); break
        c = w[k[0] + 1:k[0] + 5].astype(np.int64) & 0xffffffff
        rows.append([(c[1]-c[0]) % 2**32, (c[2]-c[1]) % 2**32, (c[3]-c[2]) % 2**32])
    r = np.median(np.array(rows), axis=0) / 100.0
    print(f"B={B} T={T} S={S}: DP {r[0]:.1f} us ({r[0]*1e3/T:.1f} ns/frame), DP-end -> walk start {r[1]:.1f} us, walk {r[2]:.1f} us ({r[2]*1e3/T:.1f} ns/frame)")
