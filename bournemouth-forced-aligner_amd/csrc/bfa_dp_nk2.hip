// K1 instantiation for posterior widths C <= 32 (see bfa_dp.inc)
#define BFA_NK 2
#include "bfa_dp.inc"
