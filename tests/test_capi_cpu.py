"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares (no compute calls without a GPU), and the host-side parameter plumbing is consistent."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from parrot_amd import _lib, build
    build.build(verbose=False)
    return _lib.load()


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "parrot_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:parrot|samplernn)_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound(lib):
    from parrot_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/parrot_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in parrot_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"


def test_version_string(lib):
    assert b"gfx950" in lib.parrot_hip_version()


def test_bad_arguments_are_rejected_without_gpu(lib):
    import ctypes as C
    from parrot_amd import _lib
    plan = C.c_void_p()
    d = _lib.DecoderDesc()
    assert lib.parrot_decoder_create(C.byref(d), C.byref(plan)) == 10001
    assert lib.parrot_gemm(None, 1, 0, None, 1, 0, None, 1, 1, 1, 1, None, 1.0, 0, 0, 1, 0, 0, 0, 1, None) == 10001


def test_cpu_tensors_fail_loudly():
    from parrot_amd import _lib, ops
    with pytest.raises(_lib.HipCallError):
        ops.gemm(torch.zeros(2, 2), torch.zeros(2, 2))
    with pytest.raises(_lib.HipCallError):
        ops.batch_quantize(torch.zeros(2, 8))


def test_struct_sizes_match_header(lib):
    """ctypes mirrors of the descriptor structs must have the C layout (checked via a tiny C program)."""
    import subprocess
    import tempfile
    from parrot_amd import _lib
    import ctypes as C
    src = r'''
#include <stdio.h>
#include "parrot_hip.h"
int main(void){printf("%zu %zu %zu %zu %zu\n", sizeof(ParrotGruSeqDesc), sizeof(ParrotDecoderDesc), sizeof(ParrotSampleDesc), sizeof(SampleRnnGenDesc), sizeof(ParrotLstmSeqDesc));return 0;}
'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [C.sizeof(_lib.GruSeqDesc), C.sizeof(_lib.DecoderDesc), C.sizeof(_lib.SampleDesc),
                     C.sizeof(_lib.SampleRnnGenDesc), C.sizeof(_lib.LstmSeqDesc)]


def test_parrot_parameter_names_match_oracle():
    """The product's Blocks-style parameter names/shapes == the oracle's (same checkpoint keys)."""
    from oracle import parrot_ref as R
    from parrot_amd.model import Parrot
    for kw in (dict(num_layers=3, encoder_type='bidirectional'),
               dict(num_layers=2, encoder_type='bidirectional', weak_feedback=True, use_speaker=True),
               dict(num_layers=3, encoder_type='bidirectional', full_feedback=True, which_cost='GMM')):
        small = dict(rnn_h_dim=16, readouts_dim=12, encoder_dim=4, input_dim=6, speaker_dim=5, num_speakers=4)
        cfg = R.default_config(**small, **kw)
        m = Parrot(device='cpu', **small, **kw).allocate()
        mine = {k: tuple(v.shape) for k, v in m.get_parameter_dict().items()}
        ref = {k: tuple(v) for k, v in R.param_shapes(cfg).items()}
        assert mine == ref
        # packed views alias the flat buffer
        m.store.flat.fill_(1.0)
        assert all(float(v.min()) == 1.0 for v in m.get_parameter_dict().values())
        # set/get round trip
        vals = R.init_params(cfg, seed=5, dtype=torch.float32)
        m.set_parameter_values(vals)
        got = m.get_parameter_values()
        for k in vals:
            assert (torch.as_tensor(got[k]) == vals[k]).all(), k


def test_blocks_checkpoint_roundtrip(tmp_path):
    """tar + `_parameters` npz with '/' -> '|' (Blocks Checkpoint layout, train.py:157-173, sample.py:39-42)."""
    import tarfile
    import numpy as np
    from oracle import parrot_ref as R
    from parrot_amd.checkpoint import dump_parameters, load_parameters
    cfg = R.default_config(rnn_h_dim=8, readouts_dim=6, encoder_type='bidirectional', encoder_dim=4, input_dim=5)
    vals = {k: v.numpy() for k, v in R.init_params(cfg, seed=1, dtype=torch.float32).items()}
    path = str(tmp_path / "best_x.tar")
    dump_parameters(path, vals, carry={"B4|last_k": np.ones((4, 10), dtype='float32')})
    with tarfile.open(path) as t:
        assert t.getnames() == ['_parameters']
    got, carry = load_parameters(path, with_carry=True)
    assert set(got) == set(vals) and all(np.array_equal(got[k], vals[k]) for k in vals)
    assert '/parrot/rnn1.state_to_state' in got and carry["B4|last_k"].shape == (4, 10)


def test_integration_md_stub_runs_against_the_library():
    """The ctypes stub INTEGRATION.md shows a maintainer is real code: it executes against the built library (every symbol it
    binds exists)."""
    import ctypes as C
    import os
    import re
    import torch  # noqa: F401  (the library must be loaded after PyTorch-ROCm's own HIP runtime)
    from parrot_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", txt, re.S).group(1)
    assert 'C.CDLL("libparrot_hip.so")' in code
    code = code.replace('C.CDLL("libparrot_hip.so")', 'C.CDLL(%r)' % _lib.LIB_PATH)
    g = {}
    exec(code, g)
    g['lib'].parrot_hip_version.restype = C.c_char_p
    assert b"gfx950" in g['lib'].parrot_hip_version()
