"""Drop-in for the `fastpathz` module the reference imports (phanotate.py:9,56-64):

    import phanotate_amd.fastpathz as fz
    fz.empty_graph(); fz.add_edge("src\\tdst\\tweight"); path = fz.get_path(source=..., target=...)

Edges arrive as the text of Edge.__str__ (edges.py:17-23): two node names and weight*1000 as a decimal
string, possibly in scientific notation.  The weight is expanded to an arbitrary-precision integer, its
fractional digits dropped (what fastpathz does with GMP, CHANGELOG.md:11-13,54-57), and the exact
shortest path is solved on the GPU by libphx (phx_solve).  Module-global state like the original.
"""
from decimal import Decimal

from .api import Annotator

_names = {}
_src, _dst, _w = [], [], []
_ann = None


def empty_graph():
    _names.clear()
    del _src[:], _dst[:], _w[:]


def _id(name):
    if name not in _names:
        _names[name] = len(_names)
    return _names[name]


def add_edge(text):
    s, d, w = text.split("\t")
    _src.append(_id(s))
    _dst.append(_id(d))
    _w.append(int(Decimal(w)))  # truncation toward zero
    return None


def get_path(source=None, target=None):
    global _ann
    if source not in _names or target not in _names:
        return []
    if _ann is None:
        _ann = Annotator()
    path, dist = _ann.solve(len(_names), _src, _dst, _w, _names[source], _names[target])
    rev = {v: k for k, v in _names.items()}
    return [rev[v] for v in path]
