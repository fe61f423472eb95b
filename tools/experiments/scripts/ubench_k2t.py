"""K2 phase times (BFA_K2_TIMING build): status[1..7] of the batch carry utterance 0's walk counters."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.synth import synth_ragged
from bournemouth_forced_aligner_amd import AlignmentUtils
B, tlo, thi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
lp, tk, T_len, S_len = synth_ragged(B, tlo, thi, 67, 1004, dev)
au = AlignmentUtils(blank_id=66, silence_id=0)
hint = au.viterbi_decoder.class_mask_hint(T_len.tolist(), S_len.tolist(), has_sil=False, n_classes=67)
Td, Sd = T_len.to(dev), S_len.to(dev)
for _ in range(3):
    res = au.decode_alignments_device(lp, tk, Td, Sd, class_mask=hint)
torch.cuda.synchronize()
print(res.frame_phonemes_idx[:4, :8].cpu().tolist()); sys.exit(0)
print("T0", int(T_len[0]), "S0", int(S_len[0]), "R*100+NC", st[1], "chunks", st[2], "wait_cyc", st[3] * 16, "walk_cyc", st[4] * 16,
      "out_cyc", st[5] * 16, "iters", st[6], "stages", st[7])
