"""Import path of the reference (pastml.models.HKYModel); implementation in _closed_form.py."""
from pastml_amd.models._closed_form import HKYModel, HKY, HKY_STATES, A, C, G, T, KAPPA  # noqa: F401
