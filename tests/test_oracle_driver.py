"""Pins oracle/driver.py to the reference's golden driver trajectory
(tf_agents/drivers/dynamic_step_driver_test.py:168-199 with the mocks of drivers/test_utils.py)."""
from oracle import driver


def _col(seen, k):
    return [t[k] for t in seen]


def test_golden_trajectory_six_steps():
    env, pol = driver.MockEnv(), driver.MockPolicy()
    seen, final_ts, _ = driver.run_step_driver(env, pol, num_steps=6)
    assert _col(seen, "step_type") == [0, 1, 2, 0, 1, 2, 0, 1]
    assert _col(seen, "observation") == [0, 1, 3, 0, 1, 3, 0, 1]
    assert _col(seen, "action") == [1, 2, 1, 1, 2, 1, 1, 2]
    assert _col(seen, "policy_info") == [2, 4, 2, 2, 4, 2, 2, 4]
    assert _col(seen, "next_step_type") == [1, 2, 0, 1, 2, 0, 1, 2]
    assert _col(seen, "reward") == [1, 1, 0, 1, 1, 0, 1, 1]
    assert _col(seen, "discount") == [1, 0, 1, 1, 0, 1, 1, 0]
    assert final_ts["step_type"] == 2


def test_one_step_and_continue():  # testOneStepUpdatesObservers / two runs share env state
    env, pol = driver.MockEnv(), driver.MockPolicy()
    seen, ts, st = driver.run_step_driver(env, pol, num_steps=1)
    assert len(seen) == 1
    seen2, ts2, _ = driver.run_step_driver(env, pol, num_steps=1, time_step=ts, policy_state=st)
    assert seen2[0]["step_type"] == 1 and seen2[0]["action"] == 2


def test_boundary_steps_are_not_counted():
    env, pol = driver.MockEnv(), driver.MockPolicy()
    seen, _, _ = driver.run_step_driver(env, pol, num_steps=3)
    # FIRST, MID, LAST(boundary, not counted), FIRST -> 4 iterations for 3 counted steps
    assert _col(seen, "step_type") == [0, 1, 2, 0]


def test_maximum_iterations():
    env, pol = driver.MockEnv(), driver.MockPolicy()
    seen, _, _ = driver.run_step_driver(env, pol, num_steps=100, maximum_iterations=5)
    assert len(seen) == 5
