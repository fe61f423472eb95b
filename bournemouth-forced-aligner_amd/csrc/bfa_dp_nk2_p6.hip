// K1 class kernels of the hot path, width class 2, part 6: the merged narrow-class kernel of the silence-anchored mode (see BFA_PART in bfa_dp.inc)
#define BFA_NK 2
#define BFA_DP3_NFULL 1
#define BFA_DP3_TAIL 1
#define BFA_PART 6
#include "bfa_dp.inc"
