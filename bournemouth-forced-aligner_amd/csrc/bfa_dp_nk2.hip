// K1 instantiation for posterior widths C <= 32 (see bfa_dp.inc); C == 17 takes the bfa_dp3.inc hot path
// (launchers + generic kernels; the class kernels of the hot path compile in bfa_dp_nk2_p2..p5.hip, see BFA_PART)
#define BFA_NK 2
#define BFA_DP3_NFULL 1
#define BFA_DP3_TAIL 1
#define BFA_PART 1
#include "bfa_dp.inc"
