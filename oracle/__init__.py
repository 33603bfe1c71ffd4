"""CPU oracle for the CCST hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (fp32, CPU) restatement of the reference's
algorithm for the AdaIN style-transfer path and the per-client ResNet /
FedAvg path.  It is the checker the HIP path is compared against; it is never
the thing measured or shipped.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``ccst_amd/`` imports it, and ``ccst_amd`` raises if its HIP library is
missing rather than falling back to this code.

Parity pinning: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself: ``tools/make_golden.py`` (runs only where /root/reference exists)
imports the reference's ``net.py`` / ``function.py``, AST-extracts the
script-level functions, runs them on seeded inputs and commits the outputs
under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks this oracle
against those vectors bit-for-bit on CPU.  The ResNet residual blocks live in
torchvision (absent from /root/reference and from this image); they are
restated from torchvision's published BasicBlock/Bottleneck (v1.5: stride on
the 3x3) -- that part of the parity is "unpinned" beyond the reference's own
stem/_make_layer/head, which ARE exercised through the reference's ResNet
class with these blocks injected (see tools/make_golden.py).
"""
from . import adain_ref, resnet_ref, fed_ref  # noqa: F401
