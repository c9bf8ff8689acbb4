"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the video-clip hot path.

Nothing in the product path may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
use it, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  The restatement in ``oracle/clip_path.py`` is checked
(tests/test_oracle_golden.py) against golden vectors produced by importing the
reference's own ``src/models/vit.py`` in the build container
(``tools/gen_golden.py`` -> ``tests/golden/*.npz``).  The reference ships no
tests or fixtures of its own for this path.
"""
