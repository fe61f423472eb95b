// K1 instantiation for posterior widths C <= 80 (see bfa_dp.inc); C == 67 takes the bfa_dp3.inc hot path
// (launchers + generic kernels; the class kernels of the hot path compile in bfa_dp_nk5_p2..p5.hip, see BFA_PART)
#define BFA_NK 5
#define BFA_DP3_NFULL 4
#define BFA_DP3_TAIL 3
#define BFA_PART 1
#include "bfa_dp.inc"
