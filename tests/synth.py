"""Synthetic categorical tables of SURVEY.md 8(d) (the generator lives with the product: repair/synth.py)."""
from repair.synth import CARDS, balanced_weights, make_table, make_table_parallel  # noqa: F401
