// K1 class kernels of the hot path, width class 2, part 9: mixed-length calls -- DP + walk of every narrow class in one kernel (k_mix; see BFA_PART in bfa_dp.inc)
#define BFA_NK 2
#define BFA_DP3_NFULL 1
#define BFA_DP3_TAIL 1
#define BFA_PART 9
#include "bfa_dp.inc"
