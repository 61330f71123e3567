"""Drop-in surface for the reference's `torchrl` package (hot-path subset)."""

__v4l_shell__ = True  # vision4leg_amd.overlay refuses to overlay the shell onto itself
