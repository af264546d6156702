"""Import path of the reference (pastml.models.generator); implementation in _eigen.py."""
from pastml_amd.models._eigen import save_matrix, get_normalised_generator, get_diagonalisation, get_pij_matrix  # noqa: F401
