"""Generates tests/golden/quantize_golden.npz by RUNNING THE REFERENCE quantize.py
(/root/reference/quantize.py, importable under Python 3 / NumPy 2 -- SURVEY.md 8c).

Run once in the build container:  python tests/golden/make_quantize_golden.py
The GPU box has no /root/reference; tests only read the committed .npz.
"""
import importlib.util
import os

import numpy as np

REF = "/root/reference/quantize.py"
spec = importlib.util.spec_from_file_location("ref_quantize", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
batch_quantize = getattr(ref, "__batch_quantize")

rng = np.random.RandomState(1234)
x = rng.randn(8, 4000).astype(np.float32)
x[1] *= 1e-3
x[2] = np.sign(x[2]) * np.abs(x[2]) ** 3  # heavy tails
x[3, :3] = [-1.0, 0.0, 1.0]
x[4] = np.linspace(-1, 1, 4000, dtype=np.float32)
mu = batch_quantize(x.copy(), 256, "mu-law")
lin = batch_quantize(x.copy(), 256, "linear")
dec = ref.mu2linear(mu)
# docstring known answers (quantize.py:55-63)
kat_in = np.array([[-1.0, 0.0, 1.0]])
kat_mu = ref.linear2mu(kat_in)
kat_dec = ref.mu2linear(np.array([0, 255], dtype=np.int16))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quantize_golden.npz")
np.savez_compressed(out, x=x, mu=mu, lin=lin, dec=dec, kat_in=kat_in, kat_mu=kat_mu, kat_dec=kat_dec)
print("wrote", out, mu.dtype, lin.dtype, dec.dtype, kat_mu, kat_dec)
