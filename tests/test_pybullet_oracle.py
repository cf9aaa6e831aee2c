"""Real-oracle parity, run only where `pybullet` is importable (it is not in the build image nor on the GPU box, so these
tests skip there -- SURVEY.md section 8c: "parity unpinned").  They issue the same p.* call sequence as the reference env
(/root/reference/envs/rl_reach_env.py:132-319, DIRECT mode) and compare it per step with the CPU oracle, which is what
the HIP path is tested against.  If they ever disagree, the named IK switches of OrcConfig / ArmEnvConfig are the knobs."""
import math

import numpy as np
import pytest

pb = pytest.importorskip("pybullet", reason="pybullet not installed: step-level parity against Bullet stays unpinned")
pybullet_data = pytest.importorskip("pybullet_data")


class _BulletReach:
    """the reference's reset()/step() p.* sequence with an explicit goal, no GUI, no debug lines"""

    def __init__(self):
        import os
        self.p = pb
        self.cid = pb.connect(pb.DIRECT)
        self.root = pybullet_data.getDataPath()
        self.os = os
        self.init_q = [0.006418, 0.413184, -0.011401, -1.589317, 0.005379, 1.137684, -0.006539]
        self.orn = pb.getQuaternionFromEuler([0., -math.pi, math.pi / 2.])
        self.damping = [0.00001] * 7

    def reset(self):
        p = self.p
        p.resetSimulation()
        p.setGravity(0, 0, -10)
        p.loadURDF(self.os.path.join(self.root, "plane.urdf"), basePosition=[0, 0, -0.65])
        self.kuka = p.loadURDF(self.os.path.join(self.root, "kuka_iiwa/model.urdf"), useFixedBase=True)
        p.loadURDF(self.os.path.join(self.root, "table/table.urdf"), basePosition=[0.5, 0, -0.65])
        self.nj = p.getNumJoints(self.kuka)
        for i in range(self.nj):
            p.resetJointState(self.kuka, i, self.init_q[i])
        pos = p.getLinkState(self.kuka, self.nj - 1)[4]
        p.stepSimulation()
        return np.array(pos)

    def q(self):
        return np.array([self.p.getJointState(self.kuka, i)[0] for i in range(self.nj)])

    def step(self, action, dv=0.02):
        p = self.p
        cur = p.getLinkState(self.kuka, self.nj - 1)[4]
        lim = [(0.2, 0.7), (-0.3, 0.3), (0.0, 0.55)]
        new = [min(max(cur[k] + action[k] * dv, lim[k][0]), lim[k][1]) for k in range(3)]
        jp = p.calculateInverseKinematics(bodyUniqueId=self.kuka, endEffectorLinkIndex=self.nj - 1, targetPosition=new,
                                          targetOrientation=self.orn, jointDamping=self.damping)
        for i in range(self.nj):
            p.resetJointState(self.kuka, i, jp[i])
        p.stepSimulation()
        return np.array(p.getLinkState(self.kuka, self.nj - 1)[4]), np.array(jp)

    def close(self):
        self.p.disconnect()


def test_fk_init_pose_against_bullet(O, kuka):
    b = _BulletReach()
    pos = b.reset()
    p, _ = O.fk(kuka, O.INIT_Q)
    assert np.abs(pos - p[0]).max() < 1e-6
    b.close()


def test_reach_steps_against_bullet(O, kuka):
    """north_star tolerance: joint positions and eef within 1e-4 per step (teacher-forced from Bullet's state), away
    from joint limits and the table."""
    b = _BulletReach()
    b.reset()
    cfg = O.default_config()
    rng = np.random.default_rng(0)
    worst_q = worst_p = 0.0
    for t in range(60):
        a = np.clip(rng.normal(0, 0.686, 3), -0.7, 0.7)
        q0 = b.q()
        pos, jp = b.step(a)
        st = O.ReachState(1)
        st.q[0] = q0; st.goal[0] = [0.4, 0.0, 0.3]
        obs, rew, done, succ, it = O.reach_step(kuka, cfg, st, a[None])
        worst_q = max(worst_q, np.abs(st.q[0] - jp).max())
        worst_p = max(worst_p, np.abs(obs[0, :3] - pos).max())
    b.close()
    assert worst_q < 1e-4 and worst_p < 1e-4, (worst_q, worst_p)
