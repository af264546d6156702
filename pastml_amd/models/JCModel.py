"""Import path of the reference (pastml.models.JCModel); implementation in _closed_form.py."""
from pastml_amd.models._closed_form import JCModel, JC  # noqa: F401
