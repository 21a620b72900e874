"""CPU oracle for the padertorch PIT hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain numpy / torch-CPU, the reference algorithms of
the hot path.  It exists to *check* the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``padertorch_amd/`` imports
it, and the product path fails loudly when the HIP library is missing.

Parity pin: every function here is checked against golden vectors produced by
importing the real reference from ``/root/reference`` (see
``tests/golden/make_golden.py``) and against the literal known answers the
reference's own tests/doctests hold (``tests/test_oracle_*.py``).  The window /
biorthogonal-window / frame-count helpers restate third-party ``paderbox``
(github.com/fgnt/paderbox, unpinned in the reference: ``setup.py:135``), which is
absent from this image; they are pinned by the reference's own known answers
(``padertorch/contrib/cb/transform.py:219-232``,
``tests/test_ops/test_stft.py:44-70,139-165``) - see ``oracle/stft_np.py``.
"""
