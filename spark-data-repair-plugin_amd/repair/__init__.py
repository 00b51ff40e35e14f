"""MI355X-native mirror of the reference's ``repair`` package (python/repair/ in
maropu/spark-data-repair-plugin) for its repair-model training + inference hot path.

The import name matches the reference (``from repair.api import Delphi``) so that this
directory can replace ``python/`` on ``sys.path``.
"""
