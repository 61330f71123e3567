from .on_policy import VecOnPolicyCollector  # noqa: F401
