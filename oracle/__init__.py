"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference's generator path (xcmyz/FastVocoder
``model/generator``): ``fv_oracle.c`` + ``ops.py`` (C, double accumulation),
``generators.py`` (the four graphs in numpy on top of it) and ``torch_port.py``
(the reference's ATen op sequence, the CPU baseline of bench.py).

Import rule: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker / the timed
CPU reference -- never as a fallback: ``fastvocoder_amd`` does not import it and
raises without its HIP library or a ROCm device.

Parity pin: PINNED.  The reference has no tests or golden vectors for this path
(SURVEY.md section 4); the restatements are checked against outputs of the reference
itself, produced in the build container by ``tests/golden/make_golden.py`` (which
imports /root/reference read-only) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` runs that check on CPU.  ``oracle/_ref`` does not
exist: the reference is pure Python and cannot travel to the GPU box.
"""
