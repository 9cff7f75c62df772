"""NormalProjectionNetwork (tf_agents/networks/normal_projection_network.py:40-190) as a
CONFIGURATION object: a Dense(D) means layer initialised with VarianceScaling(scale =
init_means_output_factor), means squashed into the spec by tanh (`tanh_squash_to_spec`), and a
state-independent standard deviation softplus(bias) with bias = std_bias_initializer_value.

No distribution network is materialised: `ActorDistributionNetwork` turns this configuration into
the fused head of `agents/ppo/ppo_actor_network.TanhNormalActorNet` (csrc/ppo.hip:
aa_ppo_head_forward / _backward), which evaluates exactly this projection."""


def tanh_squash_to_spec(inputs=None, spec=None):
    """Marker for the default mean transform (normal_projection_network.py:30-37)."""
    return "tanh_squash_to_spec"


class NormalProjectionNetwork:
    def __init__(self, sample_spec=None, activation_fn=None, init_means_output_factor=0.1,
                 std_bias_initializer_value=0.0, mean_transform=tanh_squash_to_spec,
                 std_transform="softplus", state_dependent_std=False, scale_distribution=False,
                 seed=None, seed_stream_class=None, name="NormalProjectionNetwork"):
        if activation_fn is not None:
            raise NotImplementedError("projection activation_fn is not supported")
        if state_dependent_std or scale_distribution:
            raise NotImplementedError("state_dependent_std / scale_distribution are outside the "
                                      "hot-path scope")
        if mean_transform is not tanh_squash_to_spec and mean_transform is not None:
            raise NotImplementedError("mean_transform must be tanh_squash_to_spec or None")
        if std_transform not in ("softplus",) and getattr(std_transform, "__name__", "") != \
                "softplus":
            raise NotImplementedError("std_transform must be softplus")
        self.sample_spec = sample_spec
        self.init_means_output_factor = float(init_means_output_factor)
        self.std_bias_initializer_value = float(std_bias_initializer_value)
        self.squash_means = mean_transform is tanh_squash_to_spec
        self.seed = seed
