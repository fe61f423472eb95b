import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, cases
from oracle import oracle as ora
from bournemouth_forced_aligner_amd import AlignmentUtils
C=67; peak=0.5
rng = np.random.default_rng(100 + C + int(peak * 10))
lps, toks = [], []
for _ in range(48):
    T = int(rng.integers(8, 420)); S = int(rng.integers(1, max(2, min(90, T))))
    lp, tk, _ = cases.planted_case(rng, T, S, C=C, blank=C-1, peak=peak, sigma=1.0, repeat_rate=0.1)
    lps.append(lp); toks.append(tk)
lp, tk, T_len, S_len = cases.pad_batch(lps, toks, C, C-1)
dev=torch.device('cuda',0)
for tf in (True,):
    au = AlignmentUtils(C-1, 0, silence_anchors=0, truly_forced=tf)
    res = au.viterbi_decoder.align_batch(torch.from_numpy(lp).to(dev), torch.from_numpy(tk), T_len, S_len, anchor_pauses=False, seg_cap=lp.shape[1]+1)
    torch.cuda.synchronize()
    exp = ora.decode_alignments(lp, tk, T_len, S_len, ora.make_params(C-1, 0, 0, True, tf))
    fph=res.frame_phonemes.cpu().numpy(); fidx=res.frame_phonemes_idx.cpu().numpy()
    for b in range(48):
        T=int(T_len[b]); S=int(S_len[b])
        if exp['status'][b]!=0: continue
        bad=np.flatnonzero((fph[b,:T]!=exp['frame_ph'][b,:T])|(fidx[b,:T]!=exp['frame_idx'][b,:T]))
        stride=4
        for s2 in (3,2,1):
            if stride*S+1>T: stride=s2
        L=stride*S+1
        print("item",b,"T",T,"S",S,"L",L,"R",(L+63)//64,"band",(max(L//4,20) if L>60 else 0),"nbad",len(bad), ("first %d last %d"%(bad[0],bad[-1])) if len(bad) else "")
