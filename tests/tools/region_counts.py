#!/usr/bin/env python3
"""Two lanes per env for the cube tasks: the instruction-count model, from the built code (no GPU).

push / pick rollouts at 32 768 envs run as 1 024 half-filled waves: lanes 32..63 of every wave are masked off.  The proposal:
keep them as helpers of env `lane - 32` and split each IK trip's work across the pair, exchanging through
v_permlane32_swap_b32.  A wavefront executes ONE instruction stream: a helper lane only saves time where owner and helper run
the SAME instructions on DIFFERENT data (a loop over joints / matrix entries cut in two).  Work that differs between the two
(helper: orientation error, owner: Jacobian) is a divergent branch -- the wave issues both sides one after the other under
complementary EXEC masks and nothing is saved.

This script compiles tests/tools/exp/regions.hip (every region of an IK trip as a kernel of its own, f64, KUKA fast path),
counts the vector instructions of each region in the code object, and prices the split:
   data-parallel regions  rotate_small over the seven joints (7 -> 4 per lane), dtheta = J^T y (7 rows -> 4), q += dtheta
   sequential regions     FK (a chain over the joints), quat_from_frame + orientation_error (one scalar dependency chain),
                          LDL^T + the two triangular solves (6 dependent pivots)
   structure-dependent    J J^T: 21 entries, but every entry is specialised at compile time by the exact zeros / +-1 of the
                          chain (jj_term): two lanes running the same code means giving that up (uniform 7-term dot products)
Usage: python tests/tools/region_counts.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import isa

d = tempfile.mkdtemp(prefix="regions_")
so = os.path.join(d, "regions.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off",
                       os.path.join(ROOT, "tests", "tools", "exp", "regions.hip"), "-o", so], stderr=subprocess.DEVNULL)
rows = {}
for sn, dmn, md, ins in isa.all_kernels(so):
    m = isa.mix(ins)
    name = dmn.split("(")[0].replace("void ", "")
    valu = m.get("valu_f64", 0) + m.get("valu_other", 0) + m.get("cndmask", 0)
    rows[name] = dict(valu=valu, f64=m.get("valu_f64", 0), vmem=m.get("vmem", 0), salu=m.get("salu", 0), total=m["total"])
for k in sorted(rows):
    print("%-28s VALU %4d (f64 %4d)  vmem %3d  salu %3d  total %4d" % (k, rows[k]["valu"], rows[k]["f64"], rows[k]["vmem"], rows[k]["salu"], rows[k]["total"]))


def net(name, loads, stores):
    """vector instructions of the region without its load / store scaffolding (address arithmetic: ~1 VALU per access pair)"""
    r = rows[name]
    return r["valu"] - (loads + stores) // 2 - 4


fk = net("region_fk", 14, 51)
orient = net("region_orientation", 9, 3)
dls = net("region_dls", 51, 7)
dls7 = net("region_dls_7col", 54, 7)
rot7, rot4, rot1 = net("region_rotate<7>", 21, 14), net("region_rotate<4>", 12, 8), net("region_rotate<1>", 3, 2)
swap = (rows["region_swap8"]["valu"] - rows["region_io_only"]["valu"] // 3) / 8.0
trip = fk + orient + dls + rot7 + 7
print()
print("one IK trip (update) ~ %d vector instructions: FK %d, orientation %d, DLS (Jacobian + J J^T + LDL^T + solves + J^T y + clamp) %d, "
      "7 x rotate_small %d (%.0f each), q += dtheta 7" % (trip, fk, orient, dls, rot7, (rot7 - rot4) / 3.0))
print("exchange: one f64 from lane i + 32 to lane i = 2 x v_permlane32_swap_b32 at best (priced so below); as the compiler packs it today "
      "(region_swap8): %.1f vector instructions per value" % swap)
# the split
per_rot = (rot7 - rot4) / 3.0
save_rot = rot7 - rot4                    # joints 4..6 on the helper
xchg_rot = 2 * 3 * 2 + 3 * 2              # helper needs dtheta[4..6] (3 values in) and returns (cos, sin)[4..6] (6 values out): 9 f64 = 18 swaps
jty = 7 * 6 - 6                           # dtheta = J^T y: ~6 fma per row, compile-time zeros in row 0
save_jty = 3 * 6
xchg_jty = 6 * 2 + 3 * 2                  # helper needs y[0..5] (6 in), returns 3 values
print("splittable with the same code on both lanes:")
print("  rotate_small 7 -> 4 per lane: saves %d, exchange %d  -> net %d" % (save_rot, xchg_rot, save_rot - xchg_rot))
print("  J^T y rows 7 -> 4 per lane:   saves ~%d, exchange ~%d -> net %d (the helper also needs the Jacobian columns of its rows: +%d swaps: negative)" % (save_jty, xchg_jty, save_jty - xchg_jty, 3 * 6 * 2))
print("  J J^T: the specialised build is %d instructions against %d for all seven columns; a lane-uniform form has 21 x 7 = 147 fma + the "
      "exchange of the helper's ~10 entries (20 swaps): no gain over the specialised %d" % (dls, dls7, dls))
best = save_rot - xchg_rot
print("best case: %d of ~%d vector instructions per trip = %.1f %%; push pays 5.36 trips per wave-step of 3 867 instructions: "
      "%d -> ~%d (target of VERDICT r03 #3: <= 3 300)" % (best, trip, 100.0 * best / trip, 3867, 3867 - int(5.36 * best)))
