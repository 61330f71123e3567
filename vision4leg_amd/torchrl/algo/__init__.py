from .on_policy import PPO  # noqa: F401

__all__ = ["PPO"]
