"""Entry scripts mirroring RobotLearning/omniisaacgymenvs/scripts (random policy driver, PPO harness)."""
