"""Property tests (hypothesis) of the oracle's kinematics and of the host-side sharding logic; CPU only."""
import math

import numpy as np
from hypothesis import given, settings, strategies as st

qs = st.lists(st.floats(-3.0, 3.0, allow_nan=False), min_size=7, max_size=7)


@settings(max_examples=60, deadline=None)
@given(q=qs, robot=st.sampled_from(["kuka", "diana"]))
def test_fk_frames_are_rigid_and_reach_is_bounded(O, q, robot):
    ch = O.make_chain(robot)
    p, R, zs, ps = O.fk_full(ch, q)
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-9 and abs(np.linalg.det(R) - 1.0) < 1e-9
    assert np.abs(np.linalg.norm(zs, axis=1) - 1.0).max() < 1e-9
    reach = sum(np.linalg.norm(x) for x in O.ROBOTS[robot]["xyz"])
    assert np.linalg.norm(p) <= reach + 1e-9
    # consecutive pivots are exactly one link offset apart
    off = [np.linalg.norm(x) for x in O.ROBOTS[robot]["xyz"]]
    d = np.linalg.norm(np.diff(np.vstack([np.zeros(3), ps]), axis=0), axis=1)
    assert np.abs(d - off).max() < 1e-9
    # the last joint rotates the tool about its own axis: the position does not depend on q7
    q2 = list(q); q2[6] += 1.234
    assert np.abs(O.fk_full(ch, q2)[0] - p).max() < 1e-12


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 2 ** 31 - 1), scale=st.floats(0.001, 0.02))
def test_ik_converges_on_reachable_targets(O, kuka, seed, scale):
    """From a converged pose, a small reachable displacement is reached to the residual with few updates, and the
    tool orientation stays at the target quaternion."""
    cfg = O.default_config()
    rng = np.random.default_rng(seed)
    st_ = O.ReachState(1)
    O.reach_reset_with_goal(kuka, cfg, st_, [[0.45, 0.0, 0.3]])
    for _ in range(3):
        O.reach_step(kuka, cfg, st_, np.zeros((1, 3)))
    p0, _ = O.fk(kuka, st_.q)
    tgt = p0 + rng.normal(0, scale, (1, 3))
    q1, it = O.ik(kuka, cfg, st_.q, tgt)
    p1, quat = O.fk(kuka, q1)
    assert np.linalg.norm(p1 - tgt) < 1e-4 and 1 <= it[0] <= 6
    qt = np.array(cfg.target_quat[:])
    assert min(np.abs(quat[0] - qt).max(), np.abs(quat[0] + qt).max()) < 1e-3


@settings(max_examples=200, deadline=None)
@given(total=st.integers(0, 10 ** 7), world=st.integers(1, 64))
def test_shard_ranges_tile_the_env_index_space(total, world):
    from armenv.dist import shard_range
    prev = 0
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        assert lo == prev and hi >= lo and hi - lo in (total // world, total // world + 1)
        prev = hi
    assert prev == total


@settings(max_examples=50, deadline=None)
@given(seed=st.integers(0, 2 ** 63 - 1), env=st.integers(0, 2 ** 40), ep=st.integers(0, 2 ** 31 - 1))
def test_philox_draws_are_uniform_numbers_in_unit_interval(O, seed, env, ep):
    u = O.draw6(seed, env, ep, 0)
    assert all(0.0 <= x < 1.0 for x in u) and len(set(u)) == 6
    assert u != O.draw6(seed, env, ep, 1) and u != O.draw6(seed, env + 1, ep, 0) and u != O.draw6(seed ^ 1, env, ep, 0)
