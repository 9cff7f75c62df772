from agents_amd.drivers import driver, dynamic_step_driver  # noqa: F401
