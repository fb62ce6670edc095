"""Drop-in name for the reference's validate.compute_trajectory (validate.py:61-103); implementation in trajectory.py."""
from .trajectory import compute_trajectory  # noqa: F401
