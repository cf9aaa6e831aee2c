#!/usr/bin/env python3
"""Control-flow listing of one kernel of the built libarmenv.so (no GPU): every branch with its direction and span, and the
instruction count -- the instrument for the branch-layout work of round 4 (a taken branch costs a wave that has its SIMD
to itself 20-50 ns, tests/tools/exp/fwd_branch_probe.hip, branch_cost_probe.hip: DESIGN.md section 4e).
  python tests/tools/branches.py reach_rollout_f64_kuka_p0 [--asm FILE]   (kernel short name as tests/tools/isa.py prints it)"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa

def main():
    pat = sys.argv[1]
    dump = sys.argv[sys.argv.index("--asm") + 1] if "--asm" in sys.argv else None
    for elf in isa.code_objects():
        dis = isa.disassemble(elf)
        dem = isa.demangle(dis.keys())
        for name, ins in dis.items():
            if isa.short_name(dem[name]) != pat:
                continue
            base = ins[0].addr
            pos = {i.addr: k for k, i in enumerate(ins)}
            if dump:
                with open(dump, "w") as f:
                    for k, i in enumerate(ins):
                        f.write("%5d %6x  %s\n" % (k, i.addr - base, i.text))
            print("%s: %d instructions" % (pat, len(ins)))
            for k, i in enumerate(ins):
                if i.mnem.startswith(("s_cbranch", "s_branch")):
                    m = re.match(r"(-?\d+)", i.ops)
                    off = int(m.group(1))
                    off = off - 65536 if off >= 32768 else off
                    tgt = i.addr + 4 + 4 * off
                    tk = pos.get(tgt)
                    print("  %5d  %-18s -> %5s  (%s)" % (k, i.mnem, tk, "BACK %d" % (k - tk) if tk is not None and tk <= k else "fwd +%s" % (None if tk is None else tk - k)))
            return
    print("no kernel named", pat)

if __name__ == "__main__":
    main()
