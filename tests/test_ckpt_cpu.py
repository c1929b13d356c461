"""Checkpoint bridge (scope row f3), SampleRNN half: a pickle WRITTEN BY THE REFERENCE's own lib.save_params
(sampleRNN/lib/__init__.py:96-102, executed by tests/golden/make_ckpt_golden.py) is read by the product's load_params;
and, where /root/reference exists, a pickle written by the product is read by the reference's own load_params (:104-109)."""
import os
import pickle

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PKL = os.path.join(HERE, "golden", "ref_samplernn_ckpt.pkl")


def test_product_loads_reference_written_pickle(tmp_path):
    from parrot_amd.sampleRNN import lib
    with open(PKL, "rb") as f:
        blob = pickle.load(f)
    assert len(blob) == 59 and "BigFrameLevel.GRU1.Step.Input.W0" in blob and "SampleLevel.Embedding" in blob
    lib.delete_all_params()
    lib.set_device("cpu")
    try:
        lib.load_params(PKL)
        named = lib.named_params()
        assert set(named) == set(blob)
        for n, v in blob.items():
            got = named[n].detach().numpy()
            assert got.dtype == np.float32 and got.shape == np.asarray(v).shape, n
            assert np.array_equal(got, np.asarray(v, dtype=np.float32)), n
        # and back out: product writer -> plain pickle of {name: ndarray}, same names, same values
        out = tmp_path / "again.pkl"
        lib.save_params(str(out))
        with open(out, "rb") as f:
            again = pickle.load(f)
        assert set(again) == set(blob)
        for n, v in blob.items():
            assert isinstance(again[n], np.ndarray) and np.array_equal(again[n], np.asarray(v, dtype=np.float32)), n
    finally:
        lib.delete_all_params()


@pytest.mark.skipif(not os.path.isdir(os.environ.get("PARROT_REFERENCE", "/root/reference")),
                    reason="needs the reference checkout to execute its load_params")
def test_reference_loads_product_written_pickle():
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_ckpt_golden as G
    lib, vals = G.reference_registry()
    assert G.product_roundtrip_through_reference(lib, vals) == 59
