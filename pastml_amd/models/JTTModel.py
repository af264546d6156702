"""Import path of the reference (pastml.models.JTTModel); implementation in _eigen.py."""
from pastml_amd.models._eigen import JTTModel, JTT, JTT_STATES, JTT_FREQUENCIES, JTT_RATE_MATRIX, NUM_AA  # noqa: F401
