"""Mirror of the reference's global config singleton (/root/reference/config.py:29-123): every field name and default
value, so that ``from armenv.config import opt`` can stand in for ``from config import opt`` in a caller that reads
fields this build does not use itself (visdom, DDPG / DARC knobs).  ``opt`` is a class-attribute singleton updated with
``opt._parse(dict)`` exactly like the reference's."""
import warnings

import torch


class DefaultConfig(object):
    env = 'RLReachEnv'          # config.py:31
    algo = 'DADDPG_MLP'         # config.py:32 (the reference's default; this build ships the TD3 learner only)
    vis_name = 'Reach_DADDPG'   # visdom env (config.py:35-38; plotting is outside this build)
    vis_port = 8097
    jsonfile = "visdata/push/updata_TD3/TD3.json"
    csvname = "visdata/push/updata_TD3/updata_TD3_"

    # reach env parameter (config.py:41-42)
    reach_ctr = 0.02            # arm moving rate every step
    reach_dis = 0.01            # target distance

    # train parameter (config.py:45-52)
    use_gpu = True
    device = torch.device('cuda') if use_gpu else torch.device('cpu')     # config.py:46
    random_seed = 0
    num_episodes = 500
    n_train = 40
    minimal_episodes = 5
    max_steps_one_episode = 500

    # net parameter (config.py:55-58)
    actor_lr = 1e-3
    critic_lr = 1e-3
    hidden_dim = 256
    batch_size = 256

    # public algo parameter (config.py:61-64)
    sigma = 0.1
    tau = 0.005
    gamma = 0.98
    buffer_size = 1000000

    # DDPG / DARC knobs (config.py:66-67,75-76), carried for callers that read them
    epsilon = 0.01
    target_update = 10
    q_weight = 0.2
    regularization_weight = 0.005

    # TD3 (config.py:71-73)
    policy_noise = 0.2
    noise_clip = 0.5
    policy_freq = 3

    # HER (config.py:80)
    her_ratio = 0.8

    def _parse(self, kwargs):
        """config.py:81-101: setattr every key, warn on unknown ones."""
        for k, v in kwargs.items():
            if not hasattr(self, k):
                warnings.warn("Warning: opt has not attribut %s" % k)
            setattr(self, k, v)
        type(self).device = torch.device('cuda') if self.use_gpu else torch.device('cpu')    # config.py:92


opt = DefaultConfig()
