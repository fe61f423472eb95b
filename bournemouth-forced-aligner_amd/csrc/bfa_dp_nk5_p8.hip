// K1 class kernels of the hot path, width class 5, part 8: the exact-window classes (see BFA_PART in bfa_dp.inc)
#define BFA_NK 5
#define BFA_DP3_NFULL 4
#define BFA_DP3_TAIL 3
#define BFA_PART 8
#include "bfa_dp.inc"
