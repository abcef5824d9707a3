"""moved to scenarios/imu_inputs.py (product-side tools use it too); kept as an alias for the tests"""
from scenarios.imu_inputs import *  # noqa: F401,F403
from scenarios.imu_inputs import CFG, make_state, make_steps  # noqa: F401
