import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch, cases
from oracle import oracle as ora
from bournemouth_forced_aligner_amd import AlignmentUtils
C=67
rng = np.random.default_rng(1)
T,S=64,10
lp, tk, _ = cases.planted_case(rng, T, S, C=C, peak=2.0, sigma=2.0)
au = AlignmentUtils(C-1, 0)
dev=torch.device('cuda',0)
got = au.viterbi_decoder.prepare_emissions(torch.from_numpy(lp[None]).to(dev), torch.from_numpy(tk[None]), [T],[S]).cpu().numpy()[0]
rc,want=ora.prepare_emissions(lp, tk, ora.make_params(C-1,0))
neq=(got.view(np.int32)!=want.view(np.int32))
print("mismatch elements", neq.sum(), "of", neq.size, "rows with mismatch", neq.any(1).sum())
print("cols with mismatch:", np.flatnonzero(neq.any(0)))
r=np.flatnonzero(neq.any(1))[:3]
for i in r:
    print("row",i,"maxabs diff", np.abs(got[i]-want[i]).max(), "diffs unique", np.unique((got[i]-want[i])[neq[i]])[:5], "ncols", neq[i].sum())
# also no-boost floor-only / boost-only
