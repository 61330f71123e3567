"""Drop-in surface for the reference's `torchrl` package (hot-path subset)."""
