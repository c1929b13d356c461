"""Executes the reference's end-of-utterance heuristic -- the `for idx in range(args.num_samples)` loop of sample.py
(:145-163) as it stands in the reference script -- on seeded attention windows and commits inputs and outcomes as
tests/golden/eou_golden.npz; tests/test_utils_cpu.py holds parrot_amd.utils.end_of_utterance to it.

    python tests/golden/make_eou_golden.py      # needs /root/reference (PARROT_REFERENCE overrides)
"""
import ast
import contextlib
import io
import os

import numpy

REF = os.environ.get('PARROT_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_loop():
    src = open(os.path.join(REF, 'sample.py')).read()
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    tree = ast.parse(str(tool.refactor_string(src + '\n', 'sample.py')))
    loops = [n for n in tree.body if isinstance(n, ast.For) and 'this_features_length' in ast.dump(n)]
    assert len(loops) == 1
    return compile(ast.Module(body=[loops[0]], type_ignores=[]), 'sample.py', 'exec')


def inputs():
    rs = numpy.random.RandomState(11)
    N, S, U = 6, 120, 30
    lens = numpy.array([12, 20, 29, 8, 25, 16])
    mask = (numpy.arange(U)[None, :] < lens[:, None]).astype('float32')
    phi = numpy.zeros((N, S, U), dtype='float32')
    for n in range(N):
        speed = [0.35, 0.2, 0.1, 0.5, 0.28, 0.0][n]  # the third never reaches the end, the last never moves
        for t in range(S):
            kappa = speed * t
            phi[n, t] = numpy.exp(-0.5 * (numpy.arange(U) - kappa) ** 2) + 1e-4 * rs.rand(U)
    return phi, mask, 100


def main():
    code = reference_loop()
    phi, mask, num_steps = inputs()

    class Args(object):
        pass
    args = Args()
    args.num_samples, args.num_steps = phi.shape[0], num_steps
    g = dict(numpy=numpy, gen_phi=phi, labels_mask_tr=mask, args=args, features_lengths=[], labels_lengths=[])
    with contextlib.redirect_stdout(io.StringIO()):
        exec(code, g)
    path = os.path.join(HERE, 'eou_golden.npz')
    numpy.savez_compressed(path, phi=phi, labels_mask=mask, num_steps=numpy.int32(num_steps),
                           features_lengths=numpy.asarray(g['features_lengths'], dtype='int64'),
                           labels_lengths=numpy.asarray(g['labels_lengths'], dtype='int64'))
    print(g['features_lengths'], g['labels_lengths'], '->', path)


if __name__ == '__main__':
    main()
