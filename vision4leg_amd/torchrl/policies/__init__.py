from .continuous_policy import *  # noqa: F401,F403
