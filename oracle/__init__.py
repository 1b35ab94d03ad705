"""CPU oracle for the RSSFormer training-step hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 CPU restatement of the reference algorithm
(`/root/reference/RSSFormer-TIP2023`, file:line cited per function in
`rssformer_cpu.py`).  It exists to CHECK the HIP product path, never to serve it:

* only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
  import it; nothing under `representationlearning_amd/` does (a test enforces that);
* it is pinned against golden vectors emitted by importing the real reference in the
  build container (`oracle/make_golden.py` -> `tests/golden/*.npz`).  The reference
  ships no tests or fixtures of its own, so those vectors are the
  only authority; trainer internals that live in the un-vendored `ever` package
  (loss-key summation, AMP/DDP flags, PixelMetric) are "parity unpinned".
"""
