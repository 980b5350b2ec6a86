"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A functional (state-dict driven) restatement in plain CPU fp32 PyTorch of the
reference's conv-net hot path.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this package; the
product path (``medical-image-segmentation-and-classification_amd/``) never
does and fails loudly when its HIP library is missing.

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference's own model classes and its
own ``train()`` imported in the build container by ``oracle/make_golden.py``;
the resulting vectors are committed under ``tests/golden/``.
"""
