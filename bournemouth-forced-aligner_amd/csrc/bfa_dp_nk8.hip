// K1 instantiation for posterior widths C <= 128 (see bfa_dp.inc)
#define BFA_NK 8
#include "bfa_dp.inc"
