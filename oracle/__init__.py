"""TEST INFRASTRUCTURE ONLY (oracle for parity tests, smoke() and bench.py's cpu_baseline leg).

Nothing under ``elevation_mapping_cupy_amd/`` may import this package: the product path is the HIP
library behind ``include/emap_hip.h`` and it fails loudly when that library is missing.

* ``oracle.ref_kernels``  -- the reference's own kernel source compiled for the host (sequential execution);
  built by ``oracle/build_ref.py`` into ``oracle/_ref`` from the sources under /root/reference.
* ``oracle.emap_oracle``  -- a plain-C restatement of the frame (``oracle/emap_oracle.c``) implementing the
  deterministic two-phase contract documented in DESIGN.md; validated against ``ref_kernels`` and pinned by
  the golden vectors under ``tests/golden``.
"""
