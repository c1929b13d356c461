"""Executes the reference's own batch pipeline -- datasets.parrot_stream (datasets.py:199-298) and every helper it calls, the
code as it stands in the reference file -- over a synthetic in-memory dataset, and commits every item the stream yields as
tests/golden/stream_golden.npz.  tests/test_datasets_cpu.py holds parrot_amd.datasets.parrot_stream to it.

What is the reference's and what is restated: the composition (sort window, re-batching, drop rule, padding, source
selection, time-major transpose, 80-sample chunking + per-row quantisation, TBPTT segmentation, start flag, noise source) is
executed from the reference file.  Fuel itself is not in the reference checkout; the dozen Fuel classes the pipeline
instantiates are restated below from Fuel's documented behaviour (mila-udem/fuel 0.2: Batch, Mapping/SortMapping, Unpack,
Filter, Padding, FilterSources, Rename, AgnosticSourcewiseTransformer, the example schemes).  One assumption: `features`
is the first source of the HDF5 file (the reference's sort key is `len(data[0])`, datasets.py:20-21).

    python tests/golden/make_stream_golden.py      # needs /root/reference (PARROT_REFERENCE overrides)
"""
import ast
import os
import sys

import numpy

REF = os.environ.get('PARROT_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CASES = {
    'text_raw_speaker': dict(use_speaker=True, raw_data=True, noise_level=0.5),
    'text_plain': dict(use_speaker=False, raw_data=False, noise_level=None),
}
COMMON = dict(which_sets=('valid',), batch_size=4, seq_size=20, sorting_mult=2, labels_type='text')
N_EXAMPLES = 19


# ------------------------------------------------------------------------------------------------ Fuel, restated
class ConstantScheme(object):
    def __init__(self, batch_size, num_examples=None):
        self.batch_size = batch_size


class SequentialExampleScheme(object):
    def __init__(self, examples):
        self.order = list(range(examples))


class ShuffledExampleScheme(SequentialExampleScheme):
    def __init__(self, examples):
        raise NotImplementedError("the golden uses the sequential (validation) order")


class _Stream(object):
    produces_examples = True
    axis_labels = None

    def get_epoch_iterator(self):
        self.reset()
        return self

    def __iter__(self):
        return self

    def __next__(self):
        return self.get_data()

    next = __next__


class DataStream(_Stream):
    def __init__(self, dataset, iteration_scheme):
        self.dataset, self.scheme = dataset, iteration_scheme
        self.sources = dataset.sources

    @classmethod
    def default_stream(cls, dataset, iteration_scheme=None):
        return cls(dataset, iteration_scheme)

    def reset(self):
        self._it = iter(self.scheme.order)

    def get_data(self, request=None):
        return self.dataset.example(next(self._it))


class Transformer(_Stream):
    def __init__(self, data_stream=None, produces_examples=None, **kwargs):
        self.data_stream = data_stream
        self.produces_examples = produces_examples
        self.axis_labels = kwargs.get('axis_labels')
        self.child_epoch_iterator = None

    @property
    def sources(self):
        return self.data_stream.sources

    def reset(self):
        self.child_epoch_iterator = self.data_stream.get_epoch_iterator()


class Batch(Transformer):
    def __init__(self, data_stream, iteration_scheme, strictness=0):
        super(Batch, self).__init__(data_stream, produces_examples=False)
        self.batch_size = iteration_scheme.batch_size

    def get_data(self, request=None):
        data = [[] for _ in self.sources]
        for _ in range(self.batch_size):
            try:
                for source_data, example in zip(data, next(self.child_epoch_iterator)):
                    source_data.append(example)
            except StopIteration:
                if not data[0]:
                    raise
                break
        return tuple(data)  # (Fuel: numpy.asarray per source; ragged sources stay sequences of arrays)


class Unpack(Transformer):
    def __init__(self, data_stream):
        super(Unpack, self).__init__(data_stream, produces_examples=True)
        self.data = None

    def get_data(self, request=None):
        if not self.data:
            data = next(self.child_epoch_iterator)
            self.data = list(zip(*data))[::-1]  # popped from the end: original order
        return self.data.pop()


class Mapping(Transformer):
    def __init__(self, data_stream, mapping, add_sources=None):
        super(Mapping, self).__init__(data_stream, produces_examples=data_stream.produces_examples)
        self.mapping, self.add_sources = mapping, add_sources

    @property
    def sources(self):
        return self.data_stream.sources + (self.add_sources if self.add_sources else ())

    def get_data(self, request=None):
        data = next(self.child_epoch_iterator)
        image = self.mapping(data)
        return image if not self.add_sources else tuple(data) + tuple(image)


class SortMapping(object):
    def __init__(self, key, reverse=False):
        self.key, self.reverse = key, reverse

    def __call__(self, batch):
        output = sorted(zip(*batch), key=self.key, reverse=self.reverse)  # stable
        return tuple(list(col) for col in zip(*output))


class Filter(Transformer):
    def __init__(self, data_stream, predicate):
        super(Filter, self).__init__(data_stream, produces_examples=data_stream.produces_examples)
        self.predicate = predicate

    def get_data(self, request=None):
        while True:
            data = next(self.child_epoch_iterator)
            if self.predicate(data):
                return data


class Rename(Transformer):
    def __init__(self, data_stream, names):
        super(Rename, self).__init__(data_stream, produces_examples=data_stream.produces_examples)
        self._sources = tuple(names.get(s, s) for s in data_stream.sources)

    @property
    def sources(self):
        return self._sources

    def get_data(self, request=None):
        return next(self.child_epoch_iterator)


class FilterSources(Transformer):
    def __init__(self, data_stream, sources):
        super(FilterSources, self).__init__(data_stream, produces_examples=data_stream.produces_examples)
        assert all(s in data_stream.sources for s in sources), (sources, data_stream.sources)
        self._sources = tuple(s for s in data_stream.sources if s in sources)  # the STREAM's order

    @property
    def sources(self):
        return self._sources

    def get_data(self, request=None):
        data = next(self.child_epoch_iterator)
        return tuple(d for d, s in zip(data, self.data_stream.sources) if s in self._sources)


class Padding(Transformer):
    def __init__(self, data_stream, mask_sources=None, mask_dtype=None):
        super(Padding, self).__init__(data_stream, produces_examples=False)
        self.mask_sources = data_stream.sources if mask_sources is None else mask_sources
        self.mask_dtype = mask_dtype or 'float32'  # theano.config.floatX of the reference run

    @property
    def sources(self):
        out = []
        for s in self.data_stream.sources:
            out.append(s)
            if s in self.mask_sources:
                out.append(s + '_mask')
        return tuple(out)

    def get_data(self, request=None):
        batch = next(self.child_epoch_iterator)
        out = []
        for s, source_batch in zip(self.data_stream.sources, batch):
            if s not in self.mask_sources:
                out.append(source_batch)
                continue
            shapes = [numpy.asarray(x).shape for x in source_batch]
            lengths = [sh[0] for sh in shapes]
            rest = shapes[0][1:]
            assert all(sh[1:] == rest for sh in shapes)
            dtype = numpy.asarray(source_batch[0]).dtype
            padded = numpy.zeros((len(source_batch), max(lengths)) + rest, dtype=dtype)
            mask = numpy.zeros((len(source_batch), max(lengths)), dtype=self.mask_dtype)
            for i, x in enumerate(source_batch):
                padded[i, :lengths[i]] = x
                mask[i, :lengths[i]] = 1
            out += [padded, mask]
        return tuple(out)


class AgnosticSourcewiseTransformer(Transformer):
    def __init__(self, data_stream, produces_examples, which_sources=None, **kwargs):
        super(AgnosticSourcewiseTransformer, self).__init__(data_stream, produces_examples=produces_examples, **kwargs)
        self.which_sources = data_stream.sources if which_sources is None else which_sources

    def get_data(self, request=None):
        data = list(next(self.child_epoch_iterator))
        for i, s in enumerate(self.data_stream.sources):
            if s in self.which_sources:
                data[i] = self.transform_any_source(data[i], s)
        return tuple(data)


class ShimDataset(object):
    """In-memory stand-in for VoiceData(H5PYDataset): the examples of parrot_amd.datasets.VoiceData (synthetic)."""
    sources = ('features', 'text', 'raw_audio', 'speaker_index')

    def __init__(self, examples):
        self.examples, self.num_examples = examples, len(examples)

    def example(self, i):
        e = self.examples[i]
        return (e['features'], e['labels'], e['raw_audio'], e['speaker_index'])


def synthetic_examples():
    from parrot_amd.datasets import VoiceData
    return VoiceData('vctk', ('valid',), num_examples=N_EXAMPLES, seed=77)


def reference_parrot_stream(dataset):
    src = open(os.path.join(REF, 'datasets.py')).read()
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    tree = ast.parse(str(tool.refactor_string(src + '\n', 'datasets.py')))
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name != 'VoiceData']

    class PyDiv(ast.NodeTransformer):  # python-2 integer division in _chunk
        def visit_BinOp(self, node):
            self.generic_visit(node)
            if isinstance(node.op, ast.Div):
                return ast.copy_location(ast.Call(func=ast.Name(id='_py2div', ctx=ast.Load()),
                                                  args=[node.left, node.right], keywords=[]), node)
            return node

    sys.path.insert(0, REF)
    import quantize as ref_quantize
    sys.path.remove(REF)
    g = dict(numpy=numpy, os=os, Transformer=Transformer, AgnosticSourcewiseTransformer=AgnosticSourcewiseTransformer,
             Batch=Batch, Filter=Filter, FilterSources=FilterSources, Mapping=Mapping, Padding=Padding, Rename=Rename,
             SortMapping=SortMapping, Unpack=Unpack, DataStream=DataStream, ConstantScheme=ConstantScheme,
             ShuffledExampleScheme=ShuffledExampleScheme, SequentialExampleScheme=SequentialExampleScheme,
             __batch_quantize=getattr(ref_quantize, '__batch_quantize'),
             VoiceData=lambda voice, which_sets: ShimDataset(dataset.examples),
             _py2div=lambda a, b: a // b if isinstance(a, (int, numpy.integer)) and isinstance(b, (int, numpy.integer)) else a / b)
    mod = ast.fix_missing_locations(PyDiv().visit(ast.Module(body=keep, type_ignores=[])))
    exec(compile(mod, 'datasets.py', 'exec'), g)
    return g['parrot_stream']


def main():
    ds = synthetic_examples()
    ref_stream = reference_parrot_stream(ds)
    blob = {}
    for case, kw in CASES.items():
        stream = ref_stream('vctk', **dict(COMMON, **kw))
        k = 0
        for item in stream.get_epoch_iterator():
            for name, val in zip(stream.sources, item):
                blob[f'{case}|{k}|{name}'] = numpy.asarray(val)
            k += 1
        blob[f'{case}|n'] = numpy.int32(k)
        blob[f'{case}|sources'] = numpy.array(sorted(stream.sources))
    path = os.path.join(HERE, 'stream_golden.npz')
    numpy.savez_compressed(path, **blob)
    print(len(blob), 'arrays ->', path, {c: int(blob[f'{c}|n']) for c in CASES})


if __name__ == '__main__':
    main()
