"""Import path of the reference (pastml.models.CustomRatesModel); implementation in _eigen.py."""
from pastml_amd.models._eigen import CustomRatesModel, CUSTOM_RATES, load_custom_rates  # noqa: F401
