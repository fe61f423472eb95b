"""Debug: dump the origin table (last carve of the workspace) and follow the chain of utterance 0."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.synth import synth_ragged
from bournemouth_forced_aligner_amd import AlignmentUtils
B, tlo, thi = 4, 2990, 3000
dev = torch.device("cuda:0")
lp, tk, T_len, S_len = synth_ragged(B, tlo, thi, 67, 1004, dev)
au = AlignmentUtils(blank_id=66, silence_id=0)
hint = au.viterbi_decoder.class_mask_hint(T_len.tolist(), S_len.tolist(), has_sil=False, n_classes=67)
Td, Sd = T_len.to(dev), S_len.to(dev)
res = au.decode_alignments_device(lp, tk, Td, Sd, class_mask=hint)
torch.cuda.synchronize()
ws = res._keepalive[2]
Tmax = lp.shape[1]; nch = (Tmax + 63) // 64; R = 8
size = B * nch * 64 * R * 2
print("ws bytes", ws.numel(), "org bytes", size)
raw = ws.cpu().numpy().view(np.uint8)
for slack in range(0, 512, 2):
    pass
org = raw[len(raw) - size - int(sys.argv[1]) if len(sys.argv) > 1 else len(raw) - size:][:size].view(np.uint16).reshape(B, nch, 64 * R)
fidx = res.frame_phonemes_idx.cpu().numpy(); fph = res.frame_phonemes.cpu().numpy()
T0 = int(T_len[0]); L = 4 * int(S_len[0]) + 1
print("T0", T0, "L", L)
# states from the serial result (token frames only): state = 4*idx+1
e = L - 1
for c in range((T0 + 63) // 64 - 1, 0, -1):
    t = 64 * c - 1
    true_tok = fidx[0, t]
    o = int(org[0, c, e]) if e < 64 * R else -1
    print("chunk", c, "end state", e, "-> origin", o, "| frame", t, "idx", true_tok, "expected state ~", 4 * true_tok + 1 if true_tok >= 0 else "blank")
    if o == 65535 or o >= L: break
    e = o
print("row 46 around 470..481:", org[0, 46, 470:482])
print("row 46 first 16:", org[0, 46, :16])
print("row 1 first 16:", org[0, 1, :16], "row 1 256..272", org[0, 1, 256:272])
