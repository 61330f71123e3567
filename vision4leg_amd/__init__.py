"""vision4leg_amd — MI355X (gfx950) native PPO hot path behind the vision4leg `torchrl` module surface.

`vision4leg_amd.torchrl` mirrors the names of the reference's in-tree `torchrl` package for the hot path
(networks, policies, algo.PPO, replay_buffers.on_policy); put this directory on sys.path (see
INTEGRATION.md) and the reference's `starter/ppo_*.py` wire-up imports resolve to the HIP implementation.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
