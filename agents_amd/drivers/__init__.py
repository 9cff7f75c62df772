from agents_amd.drivers import driver, dynamic_episode_driver, dynamic_step_driver  # noqa: F401
