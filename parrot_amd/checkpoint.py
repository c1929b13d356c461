"""Checkpoint format bridge.

* Parrot: Blocks `Checkpoint(..., save_separately=['log'], save_main_loop=False)` writes a tar archive
  whose member ``_parameters`` is a NumPy ``.npz`` with one array per parameter, keyed by the brick path
  with '/' replaced by '|' (blocks.serialization.dump / load_parameters; call sites train.py:157-173,
  sample.py:39-42).  `load_parameters` / `dump_parameters` read and write that layout, so a checkpoint
  trained with the reference can be loaded with `Parrot.set_parameter_values` (the parameter names are the
  same Blocks paths, e.g. ``/parrot/lookuptable.W`` -- sample.py:83) and vice versa.
  The TBPTT carry (`last_*`), which the reference does not checkpoint, is stored under ``carry|...`` keys.
* SampleRNN: pickled {name: ndarray} dicts -- `parrot_amd.sampleRNN.lib.save_params / load_params`
  (lib/__init__.py:96-109).
"""
from __future__ import annotations

import io
import tarfile

import numpy

BLOCKS_SEPARATOR, NPZ_SEPARATOR = '/', '|'


def dump_parameters(path_or_file, parameter_values, carry=None):
    """Writes a Blocks-style tar with a `_parameters` npz member."""
    arrays = {k.replace(BLOCKS_SEPARATOR, NPZ_SEPARATOR): numpy.asarray(v) for k, v in parameter_values.items()}
    for k, v in (carry or {}).items():
        arrays['carry' + NPZ_SEPARATOR + k] = numpy.asarray(v)
    buf = io.BytesIO()
    numpy.savez(buf, **arrays)
    data = buf.getvalue()
    own = isinstance(path_or_file, (str, bytes))
    f = open(path_or_file, 'wb') if own else path_or_file
    try:
        with tarfile.open(fileobj=f, mode='w') as tar:
            info = tarfile.TarInfo('_parameters')
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))
    finally:
        if own:
            f.close()


def load_parameters(path_or_file, with_carry=False):
    """blocks.serialization.load_parameters: returns {brick path: ndarray}."""
    own = isinstance(path_or_file, (str, bytes))
    f = open(path_or_file, 'rb') if own else path_or_file
    try:
        with tarfile.open(fileobj=f, mode='r') as tar:
            npz = numpy.load(io.BytesIO(tar.extractfile(tar.getmember('_parameters')).read()))
            params, carry = {}, {}
            for k in npz.files:
                if k.startswith('carry' + NPZ_SEPARATOR):
                    carry[k[len('carry') + 1:]] = npz[k]
                else:
                    params[k.replace(NPZ_SEPARATOR, BLOCKS_SEPARATOR)] = npz[k]
    finally:
        if own:
            f.close()
    return (params, carry) if with_carry else params
