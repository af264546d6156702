"""Import path of the reference (pastml.models.EFTModel); implementation in _closed_form.py."""
from pastml_amd.models._closed_form import EFTModel, EFT  # noqa: F401
