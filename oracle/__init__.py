"""CPU oracle for the Impulcifer hot path -- TEST INFRASTRUCTURE ONLY.

A NumPy (float64) restatement of the reference algorithms for sweep deconvolution, IR
peak/crop/decay and minimum-phase FIR generation, each function citing the reference file:line
it follows (paths relative to the reference checkout, 115dkk/Impulcifer-pip313 v2.13.3) or the
SciPy 1.15.3 routine it restates (the reference pins only scipy>=1.12.0, pyproject.toml:33).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker.  The product (impulcifer-pip313_amd/impulse_hip) never imports it and
has no CPU fallback.

Parity is PINNED: tests/test_oracle_golden.py checks every function here against fixtures under
tests/golden/ that were produced by running the reference itself (tests/golden/make_goldens.py)
and against the golden artefact the reference ships (data/demo/room-responses.wav, FC-left
track).  The only un-pinned corner is documented where it occurs (none at present).
"""
