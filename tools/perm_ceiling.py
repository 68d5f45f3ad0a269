"""Instruction-cost ceiling of the Poseidon2 permutation on one MI355X (bench.py: valu.ceiling_perms_per_s).

Dynamic VALU instruction mix of ONE wave-level call of rsv::poseidon2() (64 permutations), derived from the source
(recursive-stwo_amd/csrc/poseidon2.hpp) and checked against the hardware count:

  S-box with its pre-reduction, x -> x^5 (142 of them: 8 full rounds x 16 + 14 partial rounds x 1)
      fold2 (lshr, add) + round constant and canonicalisation in one (2 add-literal, min)
      + pow5 (2 x [add, mad, lshr, add, add-literal, min] + [mad, lshr, add])
      = 14 fast, 3 v_min_u32, 3 v_mad_u64_u32 without addend            (the partial-round S-box has no fold2: 12 fast)
  full-round linear layer mds16_2x (9 of them; none carries round constants: they are literals of the S-box reduction)
      per 4-word group 2 mad (no addend) + 4 mad (addend) + 2 lshl_add_u64 + 2 lshl_add_u64 (plain adds)        = 40
      column sums 12 + 16 lshl_add_u64                                                                          = 28
  partial round (14): 2 mad (no addend) + 14 + 1 + 15 mad (addend) + 1 lshl_add_u64, 16 fold2 = 32 fast
  output: 16 x (lshr, add, add-literal, min)

  class                      count   cycles/instr at 4 waves/SIMD, expressed at 2.4 GHz (tools/valu_lab.hip, measured r2)
  fast  (add/sub/lshr/and)    2456   2.50     (a 32-bit literal operand does not change the class: 2.52)
  v_min_u32                    442   4.27
  v_mad_u64_u32, no addend     526   4.54
  v_lshl_add_u64               410   4.48
  v_mad_u64_u32, with addend   564   5.10     (SGPR multiplier or live 64-bit addend: 5.05-5.15)
  total                       4398            = the static count: the function is straight-line code since the constants
                                              became literals (before: 4 428 with 736 addend-mads, 15 650 cycles, 10.05 G/s)

=> 15 129 cycles-at-2.4-GHz per 64 permutations per SIMD => 1024 SIMDs x 2.4e9 / 15 129 x 64 = 10.40 G permutations/s.
(The lab's "cycles at 2.4 GHz" are wall time x 2.4 GHz.  Round 5 separated clock from issue cost (tools/valu_clock.sh,
profiles/r5_valu_lab_*): under the lab's dense VALU load GRBM_GUI_ACTIVE holds 2.35-2.40 GHz — v_and_b32 at 4 waves per
SIMD: 2.242 ms at 2.383 GHz for 2 097 152 wave-instructions per SIMD = 2.55 REAL cycles each — so the fast class's 2.5
against the nominal 2 (MI355X_MICROARCH.md) is issue overhead of the SIMD, not a power state; one wave alone issues every
5.1 cycles.  The per-wave clock64 column of the lab shows the arbiter instead: it favours a SIMD's oldest wave, the waves
finish one after the other, and only the longest-lived one spans the launch.)

Usage: python tools/perm_ceiling.py [path/to/asm]   — with an assembly listing (hipcc -S --cuda-device-only) it also
prints the STATIC opcode histogram of rsv::poseidon2 as a cross-check of the class membership."""
import collections
import re
import sys

MIX = [("fast", 2456, 2.50), ("v_min_u32", 442, 4.27), ("v_mad_u64_u32 (no addend)", 526, 4.54),
       ("v_lshl_add_u64", 410, 4.48), ("v_mad_u64_u32 (addend)", 564, 5.10)]
SIMDS, LAB_GHZ = 1024, 2.4


def ceiling():
    cycles = sum(n * c for _, n, c in MIX)
    return SIMDS * LAB_GHZ * 1e9 / cycles * 64.0, cycles, sum(n for _, n, _ in MIX)


def main():
    perms, cycles, insts = ceiling()
    print(f"{insts} VALU instructions, {cycles:.0f} cycles@2.4GHz per wave-level call -> ceiling {perms / 1e9:.2f} G permutations/s")
    if len(sys.argv) > 1:
        s = open(sys.argv[1]).read()
        m = re.search(r"\n_ZN3rsv9poseidon2ENS_7State16E:.*?s_setpc_b64", s, re.S)
        ops = collections.Counter(l.split()[0] for l in m.group(0).splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";")))
        print("static histogram of rsv::poseidon2 (straight-line code: equals the dynamic count):", ops.most_common(12))


if __name__ == "__main__":
    main()
