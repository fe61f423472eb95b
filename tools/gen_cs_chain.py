#!/usr/bin/env python3
"""tools/gen_cs_chain.py -- prints the inline-asm block of detect_silence_w's serial float64 sum (bfa_segment.hip): 64 steps,
lanes >= j add x[j]; the two v_readlane broadcasts of step j + D are issued ahead of the add of step j into NP rotating SGPR
pairs (s20 ...), so that no add waits for the scalar-register write of its own broadcast."""
D, NP = 5, 6


def pair(j):
    k = j % NP
    return 20 + 2 * k, 21 + 2 * k


def rl(j):
    a, b = pair(j)
    return f"v_readlane_b32 s{a}, %[xlo], {j}\\n\\tv_readlane_b32 s{b}, %[xhi], {j}\\n\\t"


print("asm volatile(")
for j in range(D):
    print(f'    "{rl(j)}"')
for j in range(64):
    a, b = pair(j)
    parts = []
    if j + D < 64:
        parts.append(rl(j + D))
    if j > 0:
        parts.append("s_lshl_b64 exec, exec, 1\\n\\t")
    parts.append(f"v_add_f64 %[acc], %[acc], s[{a}:{b}]\\n\\t")
    print('    "' + "".join(parts) + '"')
print('    "s_mov_b64 exec, -1"')
print('    : [acc] "+v"(acc)\n    : [xlo] "v"(xlo), [xhi] "v"(xhi)')
print('    : "scc", ' + ", ".join(f'"s{r}"' for r in range(20, 20 + 2 * NP)) + ");")
