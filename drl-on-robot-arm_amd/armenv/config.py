"""Mirror of the reference's global config singleton (/root/reference/config.py:29-123), limited to the
fields the env hot path and its rollout caller read.  ``opt`` is a class-attribute singleton updated
with ``opt._parse(dict)`` exactly like the reference's."""
import warnings


class DefaultConfig(object):
    env = 'RLReachEnv'          # config.py:31
    algo = 'TD3_MLP'

    # reach env parameter (config.py:41-42)
    reach_ctr = 0.02            # arm moving rate every step
    reach_dis = 0.01            # target distance

    # train parameter (config.py:45-52)
    use_gpu = True
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


opt = DefaultConfig()
