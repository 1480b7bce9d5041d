"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (PyTorch-CPU fp32, plain tensor math) of the reference's
coordinate-MLP hot path, written from the math in SURVEY.md Appendix A, with
every function citing the reference file:line it follows.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the *checker*, never as the thing
measured or shipped.  Nothing under ``mri-implicit-neural-representations_amd/``
imports it; the product path fails loudly when the HIP library is missing.

Parity status: PINNED.  The reference's own tests hold no golden vectors for
this path, so the oracle is pinned against outputs of the
reference itself, generated in the build container by importing the reference's
model / loss classes (``tools/make_golden.py``) and committed as data-only
fixtures under ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks
every oracle function against those fixtures.  The eval chain (centred FFT,
RSS, PSNR) depends on the third-party ``fastmri`` package which is absent from
the container: that stage follows the published formulas and is "parity
unpinned" (SURVEY.md section 8c).
"""
from .inr_oracle import *  # noqa: F401,F403
from . import inr_oracle_bf16 as bf16  # noqa: F401  (the bf16 path's rounding model, checker of tests/test_gpu_bf16.py)
