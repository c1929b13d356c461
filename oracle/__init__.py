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
  refshim/          eager torch-backed stand-ins for Theano and the Blocks bricks + a loader that executes the
                    reference's own Python-2 sources unmodified (read from /root/reference at fixture-generation
                    time, translated in memory).

PINNED (round 2): parrot_ref.py and samplernn_ref.py reproduce, to 1e-10, vectors produced by EXECUTING the
reference's own code -- model.py (Parrot.compute_cost incl. every parameter gradient, TBPTT carry, GMM cost,
layer_norm, softmax attention, sharpening / timing; Parrot.sample_model_fun), sampleRNN/lib/ops.py (Linear with
weight norm, __GRUStep, __LSTMStep, Embedding, softmax_and_argmax) and three_tier.py (compute_cost with every
gradient for GRU-1 / LSTM-2 / GRU-2, the generate_and_save_samples loop) -- on the refshim stand-ins
(tests/golden/make_ref_golden.py -> tests/golden/ref_golden.npz, tests/test_ref_golden_cpu.py).
What remains restated rather than executed: the semantics of the Theano ops and the algebra of the five Blocks bricks
model.py instantiates (neither package is in /root/reference nor installable here); the Blocks GatedRecurrent algebra is
cross-checked against the reference's in-repo twin __GRUStep on a reference-executed vector.  Stochastic paths
(Theano's MRG stream) are outside the pin.
"""
