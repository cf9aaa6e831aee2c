"""Mechanical guards on the gfx950 code objects of libarmenv.so (no GPU: llvm-objdump on the built library).

The kernels lean on three hand-made patterns that live outside hipcc's own hazard / waitcnt bookkeeping; each has
produced silently wrong results or a hidden stall at least once.  The parity suite on the GPU is the
functional guard; these tests pin the STRUCTURE, so that a compiler update or a change in register pressure that breaks
an assumption fails here, on the CPU, before any number is wrong:

  1. no scratch in the env-step kernels and the rollout kernels without a fused actor (every per-lane array must stay in
     registers: a runtime subscript or a spill would show up as private-segment use);
  2. the rollout's action prefetch (armenv_env.h prefetch_issue / prefetch_settle): three global_load_dword straight into
     accumulation registers, settled by s_waitcnt vmcnt(0) + three v_accvgpr_read after the IK -- NOTHING else may touch
     those three AGPRs in between (the compiler believes they are defined at the issue; a copy or a spill ahead of the wait
     would move stale data);
  3. the f16x3 actor's k-loop (armenv_actor.h actor_forward_wg_f16x3): between the first and the last
     v_mfma_f32_32x32x16_f16 of a kernel the only memory instructions are the ring's own direct-to-LDS loads -- a
     compiler-generated VMEM load or scratch access there forces s_waitcnt vmcnt(0) and drains the DMA queue.
"""
import os
import re
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
import isa  # noqa: E402


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(isa.LIB):
        pytest.skip("libarmenv.so is not built")
    if not os.path.exists(os.path.join(isa.LLVM, "llvm-objdump")):
        pytest.skip("the ROCm LLVM tools (llvm-objdump, llvm-readelf) are not installed")
    rows = isa.all_kernels()
    assert len(rows) > 100
    return rows


def _env_kernels(rows, pred):
    return [(sn, md, ins) for sn, _, md, ins in rows if pred(sn)]


def test_every_kernel_variant_is_present(kernels):
    names = {sn for sn, *_ in kernels}
    for task in ("reach", "push", "pick"):
        for prec in ("f64", "f32"):
            for chain in ("kuka", "diana", "generic"):
                assert f"{task}_step_{prec}_{chain}" in names
                for p in range(4):
                    assert f"{task}_rollout_{prec}_{chain}_p{p}" in names
                for p in range(2):                                   # two-waves-per-SIMD and lane-asynchronous variants
                    assert f"{task}_rollout_{prec}_{chain}_p{p}_w2" in names
                    assert f"{task}_rollout_async_{prec}_{chain}_p{p}" in names


def test_no_scratch_in_step_and_unfused_rollout_kernels(kernels):
    sel = _env_kernels(kernels, lambda sn: re.search(r"_(step|reset)_f(64|32)_[a-z]+$", sn) or re.search(r"_rollout_f(64|32)_[a-z]+_p[01]$", sn)
                       or re.search(r"_rollout_async_f(64|32)_[a-z]+_p[01]$", sn) or sn.startswith(("fk_", "ik_")))
    assert len(sel) >= 3 * 2 * 3 * 3
    for sn, md, ins in sel:
        # no scratch TRAFFIC (vgpr_spill_count may be > 0: spills into AGPRs, not memory).  A private segment of a few dozen bytes that no
        # instruction touches is tolerated: hipcc reserves an emergency slot for its register scavenger in a frame with spilled SGPRs
        # (the f32 generic-chain rollout with the in-kernel policy: 68 bytes, zero scratch_* / buffer_* instructions)
        assert not [i.text for i in ins if i.mnem.startswith(("scratch_", "buffer_"))], sn
        assert md["scratch"] <= 128, (sn, md)


def _prefetch_triples(ins):
    """[(index, [a_x, a_y, a_z])] of three consecutive global_load_dword into AGPRs from one address at offsets 0 / 4 / 8"""
    out = []
    for k in range(len(ins) - 2):
        t = ins[k:k + 3]
        if all(i.mnem == "global_load_dword" for i in t):
            m = [re.match(r"(a\d+), (v\[\d+:\d+\]), off(?: offset:(\d+))?$", i.ops) for i in t]
            if all(m) and len({x.group(2) for x in m}) == 1 and [int(x.group(3) or 0) for x in m] == [0, 4, 8]:
                out.append((k, [x.group(1) for x in m]))
    return out


def _reachable_before(ins, start, stop):
    """Indices of the instructions that can execute after ins[start] without having passed ins[stop] (control-flow walk:
    fall-through and branch targets; out-of-line cold blocks included)."""
    by_addr = {i.addr: k for k, i in enumerate(ins)}
    seen, todo = set(), [start]
    while todo:
        k = todo.pop()
        if k in seen or k == stop or k >= len(ins):
            continue
        seen.add(k)
        i = ins[k]
        if i.mnem in ("s_endpgm",) or i.mnem.startswith("s_setpc"):
            continue
        t = isa.branch_target(i)
        if t is not None and t in by_addr:
            todo.append(by_addr[t])
        if i.mnem != "s_branch":
            todo.append(k + 1)
    return seen


def test_action_prefetch_agprs_are_untouched_between_issue_and_settle(kernels):
    sel = _env_kernels(kernels, lambda sn: re.search(r"_rollout_f(64|32)_[a-z]+_p0$", sn))
    assert len(sel) == 18
    for sn, md, ins in sel:
        triples = _prefetch_triples(ins)
        assert len(triples) == 1, (sn, triples)
        k0, regs = triples[0]
        pat = re.compile(r"\b(%s)\b" % "|".join(regs))
        # the settle: s_waitcnt vmcnt(0) immediately followed by the three v_accvgpr_read of exactly these registers
        settles = [k for k in range(1, len(ins) - 2)
                   if all(ins[k + j].mnem == "v_accvgpr_read_b32" and ins[k + j].ops.split(", ")[1] == regs[j] for j in range(3))
                   and ins[k - 1].mnem == "s_waitcnt" and "vmcnt(0)" in ins[k - 1].ops]
        assert len(settles) == 1, (sn, settles)
        k1 = settles[0]
        # everything that can run between the issue and the wait -- the whole IK, its cold blocks included -- leaves the three
        # registers alone: no copy (v_accvgpr_mov / read), no spill into them, no reuse
        window = _reachable_before(ins, k0 + 3, k1 - 1)
        assert len(window) > 500, (sn, len(window))                     # the IK really lies in between
        touched = [ins[k].text for k in sorted(window) if pat.search(ins[k].ops)]
        assert not touched, (sn, touched)
        # and the loads themselves are not re-issued inside the window (one prefetch in flight)
        assert not any(k0 <= k < k0 + 3 for k in window), sn


def test_f16x3_k_loop_has_no_compiler_vmem(kernels):
    sel = [(sn, md, ins) for sn, _, md, ins in kernels if any(i.mnem == "v_mfma_f32_32x32x16_f16" for i in ins)]
    # the fused-actor rollout of every (task, precision, chain) + the two standalone actors + the fused DATD3 rollout of every (task,
    # precision, chain) (ONE copy of the (obs + 3)-input pass inside the loop over the four nets) + its two standalone kernels
    assert len(sel) == 18 + 2 + 18 + 2
    for sn, md, ins in sel:
        m = [k for k, i in enumerate(ins) if i.mnem == "v_mfma_f32_32x32x16_f16"]
        # two copies of the env-tile pass since round 4 (armenv_actor.h actor_forward_wg_f16x3_impl): the full-workgroup one with its
        # sixteen k-steps unrolled -- layer 1 of the first row tile (three f16 passes), then eight row tiles of two k-steps of 24 with
        # layer 1 of the next row tile (3) between them -- and the ragged-workgroup one with the rolled row-tile loop
        # (the unrolled copy's layer 1 of a ninth row tile is dead code: 3 + 8 * 51 - 3; the standalone actor kernels have four live
        # waves by construction and no ragged copy)
        standalone = sn.startswith(("actor_", "datd3"))
        assert len(m) == (408 if standalone else 408 + 54), (sn, len(m))
        # the k-step regions: runs of f16 MFMAs less than 150 instructions apart with at least 48 of them (the three MFMAs of an env
        # tile's first layer 1 sit apart, wherever the compiler lays the top of the env-tile loop out)
        runs, cur = [], [m[0]]
        for k in m[1:]:
            if k - cur[-1] < 150:
                cur.append(k)
            else:
                runs.append(cur); cur = [k]
        runs.append(cur)
        big = [r for r in runs if len(r) >= 48]
        # (408: the unrolled copy's first layer 1 happens to sit within 150 instructions of its first k-step, as the ragged copy's can)
        assert sorted(len(r) for r in big) in ([405], [51, 405], [54, 405], [408], [51, 408], [54, 408]) and (len(big) == 1) == standalone, (sn, [len(r) for r in runs])
        for r in big:
            if len(r) in (54, 408):
                r = r[3:]         # the copy's first layer 1 sits right in front of its loop / its first k-step
            body = ins[r[0]:r[-1] + 1]
            mem = [i for i in body if isa.classify(i.mnem) == "vmem"]
            assert mem and all(i.mnem == "global_load_lds_dwordx4" for i in mem), (sn, sorted({i.mnem for i in mem}))
            # every vmcnt wait inside is one of the hand-placed drains in front of a barrier
            waits = [k for k, i in enumerate(body) if i.mnem == "s_waitcnt" and "vmcnt" in i.ops]
            for k in waits:
                assert "vmcnt(0)" in body[k].ops and body[k + 1].mnem == "s_barrier", (sn, body[k].text, body[k + 1].text)
            # no branch inside the unrolled copy: the k-step index is a compile-time constant there
            if len(r) == 405:
                assert not any(i.mnem.startswith(("s_cbranch", "s_branch")) for i in body), sn


def test_register_budget_of_the_headline_kernels(kernels):
    """One wave per SIMD by design (512 registers per lane): the f64 step / external-rollout kernels use the whole VGPR file
    and park the overflow in AGPRs, never in scratch; the f32 engine's step kernel fits two waves per SIMD; the _w2 variants of
    the rollout kernel fit two waves per SIMD in every precision (at most 256 registers, the overflow in scratch)."""
    by = {sn: md for sn, _, md, _ in kernels}
    w2 = [(sn, md) for sn, _, md, _ in kernels if sn.endswith("_w2")]
    assert len(w2) == 3 * 2 * 3 * (2 * 2 + 1)        # rollouts: lockstep and lane-asynchronous x two policies; the step kernel
    for sn, md in w2:
        assert md["vgpr"] <= 256 and md["agpr"] == 0 and md["lds"] == 0, (sn, md)
    assert by["reach_rollout_f64_kuka_p0"]["vgpr"] <= 512 and by["reach_rollout_f64_kuka_p0"]["scratch"] == 0
    assert by["reach_step_f32_kuka"]["vgpr"] <= 256
    assert by["reach_rollout_f64_kuka_p0"]["lds"] == 0 and by["reach_step_f64_kuka"]["lds"] == 0
