"""CPU oracle for the HypDiff hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch, vectorised torch-CPU restatement of the
reference's arithmetic for the hyperbolic encode + direction-diffusion path
(manifolds/, layers/hyp_layers.py, models/direction_diffusion.py,
utils/data_utils.py of RingBDStack/HypDiff).  Every function cites the
reference file:line it follows.

Rules (enforced by tests/test_no_oracle_in_product.py):
  * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
    import it.  Nothing under hypdiff_amd/ imports it; the product path fails
    loudly when the HIP library is missing instead of falling back here.
  * It never runs on the GPU and is never the thing that is measured as the
    product.

Pinning: the reference ships no tests or golden vectors of its own (SURVEY.md
section 4).  The oracle is pinned against outputs of the reference itself,
captured in the build container by oracle/capture_golden.py (which imports
/root/reference) and committed as tests/golden/*.npz.  The exception is the
PyG-based `gcndec` denoiser (torch_geometric is not installed): PARITY
UNPINNED for that function, see DESIGN.md.
"""
