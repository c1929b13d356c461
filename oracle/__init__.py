"""CPU oracle for the parrot hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under oracle/ may be imported by the product (parrot_amd/).  Allowed importers: tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, always as the checker / reported baseline,
never as the thing measured or shipped.

What is here
  quantize_ref.py   NumPy restatement of reference quantize.py (bit-exact contract).  PINNED: checked
                    against the reference module itself (run in the build container, vectors in
                    tests/golden/quantize_golden.npz) and against the docstring known answers
                    (quantize.py:55-63).
  parrot_ref.py     torch-CPU (float64 by default) restatement of model.py: Blocks bricks, encoder,
                    Parrot.compute_cost and Parrot.sample_model (MSE head).
  samplernn_ref.py  torch-CPU restatement of sampleRNN/lib/ops.py + models/conditional/three_tier.py
                    (tiers, compute_cost, greedy generation loop).

PARITY UNPINNED for parrot_ref.py / samplernn_ref.py: the reference ships no tests, golden vectors or
checkpoints for these paths, and Theano/Blocks (un-vendored, un-pinned third-party dependencies) cannot
be installed or run here.  The restatements follow the cited reference lines and
published Blocks >= 0.2 semantics; the in-repo twin of the GRU algebra (sampleRNN/lib/ops.py:364-393)
cross-checks the Blocks GatedRecurrent formula.
"""
