// K1 class kernels of the hot path, width class 5, part 9: mixed-length calls -- DP + walk of every narrow class in one kernel (k_mix; see BFA_PART in bfa_dp.inc)
#define BFA_NK 5
#define BFA_DP3_NFULL 4
#define BFA_DP3_TAIL 3
#define BFA_PART 9
#include "bfa_dp.inc"
