from .nets import *  # noqa: F401,F403
from .base import *  # noqa: F401,F403
from .init import *  # noqa: F401,F403
