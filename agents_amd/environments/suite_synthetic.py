"""Device-resident stand-ins for `suite_gym.load` / `suite_mujoco.load`: there is no gym or MuJoCo in
this image, so an environment NAME selects a `RandomTFEnvironment` with the observation / action
specs of that task (the shapes the reference's own benchmarks use:
benchmark/dqn_benchmark_test.py:66-83).  Anything that implements `TFEnvironment` can be passed to
the train_eval scripts through `env_load_fn` instead."""
import torch

from agents_amd.environments import random_tf_environment
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import time_step as ts

_B = tensor_spec.BoundedTensorSpec
SPECS = {
    # name -> (observation spec, action spec)
    "CartPole-v0": (_B((4,), torch.float32, -4.0, 4.0), _B((), torch.int64, 0, 1)),
    "CartPole-v1": (_B((4,), torch.float32, -4.0, 4.0), _B((), torch.int64, 0, 1)),
    "Pong-v0": (tensor_spec.TensorSpec((84, 84, 4), torch.uint8), _B((), torch.int64, 0, 5)),
    "HalfCheetah-v2": (_B((17,), torch.float32, -10.0, 10.0), _B((6,), torch.float32, -1.0, 1.0)),
    "Humanoid-v2": (_B((376,), torch.float32, -10.0, 10.0), _B((17,), torch.float32, -0.4, 0.4)),
}


def load(env_name, batch_size=1, seed=0, episode_end_probability=0.02, device=None):
    if env_name not in SPECS:
        raise ValueError(f"no synthetic environment named {env_name!r}: one of {sorted(SPECS)} "
                         "or pass env_load_fn")
    obs_spec, action_spec = SPECS[env_name]
    kw = {} if device is None else {"device": device}
    return random_tf_environment.RandomTFEnvironment(
        ts.time_step_spec(obs_spec), action_spec, batch_size=batch_size,
        episode_end_probability=episode_end_probability, seed=seed, **kw)
