// K1 instantiation for posterior widths C <= 80 (see bfa_dp.inc); C == 67 takes the bfa_dp3.inc hot path
#define BFA_NK 5
#define BFA_DP3_NFULL 4
#define BFA_DP3_TAIL 3
#include "bfa_dp.inc"
