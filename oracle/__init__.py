"""CPU oracle for the Darknet53 detection hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference implementation of this path lives in the un-vendored `pytoolkit` submodule
(reference .gitmodules:1-3), which is empty in /root/reference, and the reference ships no tests, fixtures or golden
vectors for it.  This package is therefore a CPU *restatement* of the algorithm as specified by
reference docs/MODEL.md and by the tensor-layout facts leaked by check_assign.py:25-27 / check_generator.py:21; every
rule the reference does not pin is marked [BUILD-DEFINED] where it is frozen.  It is cross-checked against
torch.nn.functional (conv/BN) and against fp64 re-evaluation in tests/test_oracle_*.py, and pinned by the golden
fixtures under tests/golden/ that scripts/make_golden.py generates from it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  The product path
(object_detector_amd/) never does: it fails loudly when libodhip.so is missing.
"""
