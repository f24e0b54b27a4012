"""Locates the input decks shipped in inputs/ (Athena-style "<block>" / "key = value")."""
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def path(name):
    p = os.path.join(_ROOT, "inputs", name if name.endswith(".in") else name + ".in")
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    return p


def load(name):
    with open(path(name)) as f:
        return f.read()
