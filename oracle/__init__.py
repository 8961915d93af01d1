"""CPU oracle for the COMO photometric-GN / DepthCov hot path.  TEST INFRASTRUCTURE ONLY.

A torch-CPU / numpy restatement of the reference algorithm (edexheim/como @ 2024-12-20),
written from the maths, each function citing the reference file:line it follows.
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import
this package -- always as the checker, never as the thing measured or shipped.  The
product (como_amd/) never imports it and fails loudly if its HIP library is missing.

Parity pinning: the reference ships NO tests, golden vectors or fixtures
(SURVEY.md section 4), so this oracle is pinned against outputs of the reference itself,
generated in the build container by importing /root/reference
(tests/golden/make_golden.py -> tests/golden/*.npz, committed) and, for the two
native DepthCov ops, against the reference's own C++ CPU source compiled into
oracle/_ref/ (oracle/build_ref.py).  Third-party arithmetic the reference delegates
to packages that are not vendored (lietorch SE3.exp, torchvision resize/grayscale)
is restated from the published closed forms: for those call sites parity is UNPINNED.
"""
