from .base import BaseReplayBuffer  # noqa: F401
from .on_policy import OnPolicyReplayBuffer, DeviceOnPolicyReplayBuffer  # noqa: F401
