"""``src/``-layout import shim.

The reference's driver does ``from models.frame_transformer import FrameTransformer`` /
``from models.transformer import SimpleTransformer`` with ``src/`` as the working directory
(/root/reference/src/main.py:14-15, constructed at :37-44).  Putting THIS directory's parent on ``PYTHONPATH``

    cd <reference>/src && PYTHONPATH=<repo>/shim/src python main.py

makes those imports resolve to the MI355X build, with ``main.py`` unchanged: the reference's own ``src/models`` has no
``__init__.py``, i.e. it is a namespace-package portion, and a regular package found anywhere on ``sys.path`` takes
precedence over namespace portions -- also over the one in the script directory.  Modules the build does not replace
(``models.LSTM`` main.py:13, ``basicmlp``, ``contrastivemodel``, ``pretrained``: out of scope, SURVEY section 2 rows
8-11) keep resolving to the reference's files: every other ``models`` directory on ``sys.path`` is appended to this
package's search path.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(_HERE)))
if _REPO not in sys.path:
    sys.path.append(_REPO)
import dvt_amd  # noqa: E402,F401  (registers the package that lives in data-efficient-video-transformers_amd/)

for _entry in list(sys.path):
    _cand = os.path.join(_entry or os.getcwd(), "models")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != _HERE and _cand not in __path__:
        __path__.append(_cand)
