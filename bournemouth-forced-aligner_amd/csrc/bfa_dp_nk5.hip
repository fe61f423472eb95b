// K1 instantiation for posterior widths C <= 80 (see bfa_dp.inc)
#define BFA_NK 5
#include "bfa_dp.inc"
