"""varsens_amd -- the Saltelli hot path of LoLab-MSM/varsens on MI355X (gfx950).

Drop-in surface (``import varsens_amd as varsens``): ``Varsens``, ``Sample``,
``Objective`` and ``scale`` as in varsens/__init__.py:1-4 of the reference.
"""
from . import scale
from .scale import *  # noqa: F401,F403  (reference: `from varsens.scale import *`, varsens/__init__.py:2)
from .saltelli import Varsens, Sample, Objective
from . import objectives

__all__ = ['scale', 'Varsens', 'Sample', 'Objective']
