"""Import path of the reference (pastml.models.F81Model); implementation in _closed_form.py."""
from pastml_amd.models._closed_form import F81Model, F81  # noqa: F401
