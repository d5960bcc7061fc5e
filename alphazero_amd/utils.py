"""Small helpers of the reference's utils.py that the self-play path touches (utils.py:21-39)."""
import numpy as np

from .base import dotdict  # noqa: F401  (re-exported like the reference's utils.dotdict)


def fair_max(elements, key=lambda x: x):
    """argmax with a uniform random tie-break; the random draw happens even for a single maximum (utils.py:28-34)"""
    elements = list(elements)
    best = key(max(elements, key=key))
    ties = [x for x in elements if key(x) == best]
    return ties[np.random.choice(len(ties))]


def remove_ext(filename):
    return filename.split(".")[0]
