"""Executes the reference's own TBPTT segmentation (datasets.py:41-138, class SegmentSequence) and its raw-audio
transformer (datasets.py:20-29, 187-196, on the reference's quantize.py) -- the code as it stands in the reference file,
on a stand-in for Fuel's Transformer base class -- over seeded padded batches, and commits what came out as
tests/golden/datasets_golden.npz.  tests/test_datasets_cpu.py holds parrot_amd.datasets to it.

    python tests/golden/make_datasets_golden.py      # needs /root/reference (PARROT_REFERENCE overrides)
"""
import ast
import os
import sys

import numpy

REF = os.environ.get('PARROT_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {  # (time steps of the padded batch, seq_size, share_value, return_last, min_size)
    'tbptt_train': (173, 51, 1, False, 10),     # parrot_stream: seq_size + 1, share_value 1, return_last False
    'exact_multiple': (201, 51, 1, False, 10),
    'short_batch': (30, 51, 1, False, 10),
    'return_last': (130, 40, 0, True, 10),
    'overlap3': (95, 20, 3, True, 5),
}


def batch_for(case):
    n = CASES[case][0]
    rs = numpy.random.RandomState(len(case) * 7 + n)
    feats = rs.randn(n, 3, 5).astype('float32')
    mask = (rs.rand(n, 3) > 0.1).astype('float32')
    labels = rs.randint(0, 40, (3, 9)).astype('int32')
    return feats, mask, labels


def reference_namespace():
    src = open(os.path.join(REF, 'datasets.py')).read()
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    tree = ast.parse(str(tool.refactor_string(src + '\n', 'datasets.py')))
    keep = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == 'SegmentSequence') or
            (isinstance(n, ast.FunctionDef) and n.name in ('_chunk', '_transpose', '_length', 'get_raw_transformer'))]

    class PyDiv(ast.NodeTransformer):  # python-2 integer division in _chunk (data.shape[axis] / frame_size)
        def visit_BinOp(self, node):
            self.generic_visit(node)
            if isinstance(node.op, ast.Div):
                return ast.copy_location(ast.Call(func=ast.Name(id='_py2div', ctx=ast.Load()),
                                                  args=[node.left, node.right], keywords=[]), node)
            return node

    class Transformer(object):  # fuel.transformers.Transformer: what SegmentSequence uses of it
        def __init__(self, data_stream=None, produces_examples=None, **kwargs):
            self.data_stream = data_stream
            self.produces_examples = produces_examples
            self.child_epoch_iterator = iter(data_stream.batches)

    sys.path.insert(0, REF)
    import quantize as ref_quantize
    sys.path.remove(REF)
    g = dict(numpy=numpy, Transformer=Transformer, __batch_quantize=getattr(ref_quantize, '__batch_quantize'),
             _py2div=lambda a, b: a // b if isinstance(a, int) and isinstance(b, int) else a / b)
    mod = ast.fix_missing_locations(PyDiv().visit(ast.Module(body=keep, type_ignores=[])))
    exec(compile(mod, 'datasets.py', 'exec'), g)
    return g


class Stream(object):
    def __init__(self, batches, sources):
        self.batches, self.sources, self.produces_examples = batches, sources, False


def main():
    g = reference_namespace()
    blob = {}
    for case, (n, seq, share, ret_last, min_size) in CASES.items():
        batch = batch_for(case)
        seg = g['SegmentSequence'](Stream([batch, batch], ('features', 'features_mask', 'labels')), seq_size=seq,
                                   share_value=share, return_last=ret_last, add_flag=True, min_size=min_size,
                                   which_sources=('features', 'features_mask'))
        k = 0
        while True:
            try:
                out = seg.get_data()
            except StopIteration:
                break
            blob[f'{case}|{k}|features'] = out[0]
            blob[f'{case}|{k}|features_mask'] = out[1]
            blob[f'{case}|{k}|labels'] = out[2]
            blob[f'{case}|{k}|flag'] = numpy.int32(out[3])
            k += 1
        blob[f'{case}|n'] = numpy.int32(k)
    # raw-audio path: chunk into 80-sample frames, quantise per batch row, back to [frames, batch, 80]
    rs = numpy.random.RandomState(5)
    raw = rs.randn(4, 80 * 6).astype('float32') * 0.3
    chunked = g['_chunk'](raw)
    blob['raw|in'] = raw
    blob['raw|chunked'] = chunked
    for q_type in ('mu-law', 'linear'):
        blob[f'raw|{q_type}'] = g['get_raw_transformer'](q_type, 256)(chunked.copy())
    path = os.path.join(HERE, 'datasets_golden.npz')
    numpy.savez_compressed(path, **blob)
    print(len(blob), 'arrays ->', path, {c: int(blob[f'{c}|n']) for c in CASES})


if __name__ == '__main__':
    main()
