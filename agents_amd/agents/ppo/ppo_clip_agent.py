"""PPOClipAgent: PPOAgent with the clipped surrogate only -- KL cutoff and adaptive KL disabled
(tf_agents/agents/ppo/ppo_clip_agent.py:66-180)."""
from agents_amd.agents.ppo import ppo_agent


class PPOClipAgent(ppo_agent.PPOAgent):
    def __init__(self, time_step_spec, action_spec, optimizer=None, actor_net=None, value_net=None,
                 greedy_eval=True, importance_ratio_clipping=0.0, lambda_value=0.95,
                 discount_factor=0.99, entropy_regularization=0.0, policy_l2_reg=0.0,
                 value_function_l2_reg=0.0, shared_vars_l2_reg=0.0, value_pred_loss_coef=0.5,
                 num_epochs=25, use_gae=False, use_td_lambda_return=False, normalize_rewards=True,
                 reward_norm_clipping=10.0, normalize_observations=True, log_prob_clipping=0.0,
                 gradient_clipping=None, value_clipping=None, check_numerics=False,
                 compute_value_and_advantage_in_train=True, update_normalizers_in_train=True,
                 aggregate_losses_across_replicas=True, debug_summaries=False,
                 summarize_grads_and_vars=False, train_step_counter=None, name="PPOClipAgent",
                 seed=0):
        super().__init__(
            time_step_spec, action_spec, optimizer, actor_net, value_net, greedy_eval,
            importance_ratio_clipping, lambda_value, discount_factor, entropy_regularization,
            policy_l2_reg, value_function_l2_reg, shared_vars_l2_reg, value_pred_loss_coef,
            num_epochs, use_gae, use_td_lambda_return, normalize_rewards, reward_norm_clipping,
            normalize_observations, log_prob_clipping,
            kl_cutoff_factor=0.0, kl_cutoff_coef=0.0, initial_adaptive_kl_beta=0.0,
            adaptive_kl_target=0.0, adaptive_kl_tolerance=0.0,
            gradient_clipping=gradient_clipping, value_clipping=value_clipping,
            check_numerics=check_numerics,
            compute_value_and_advantage_in_train=compute_value_and_advantage_in_train,
            update_normalizers_in_train=update_normalizers_in_train,
            aggregate_losses_across_replicas=aggregate_losses_across_replicas,
            debug_summaries=debug_summaries, summarize_grads_and_vars=summarize_grads_and_vars,
            train_step_counter=train_step_counter, name=name, seed=seed)
