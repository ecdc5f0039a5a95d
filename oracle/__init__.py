"""TEST INFRASTRUCTURE ONLY — CPU oracle for the FF-RAFT hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline — never as a fallback for
the HIP path (``focusflow_official_amd`` raises if its HIP library is absent).

Parity status: PINNED.  ``oracle/ffraft_ref.py`` is checked against golden
vectors produced by running the reference's own ``FF_RAFT_Core`` modules in
the authoring container (``tests/golden/make_golden.py`` is the generator,
``tests/test_oracle_golden.py`` the check).
"""
