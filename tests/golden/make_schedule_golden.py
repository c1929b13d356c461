"""Executes the reference's own LearningRateSchedule (extensions.py:83-152) -- the class body as it stands in the reference
file, on stand-ins for the four Blocks names it touches -- over seeded validation-cost trajectories and commits what it did
at every check (learning rate, cuts so far, best-parameter reload, buffer reset, finish request) as
tests/golden/schedule_golden.json.  tests/test_train_logic_cpu.py holds parrot_amd.trainer.LearningRateSchedule to it.

    python tests/golden/make_schedule_golden.py      # needs /root/reference (PARROT_REFERENCE overrides)
"""
import ast
import io
import json
import os

import numpy

REF = os.environ.get('PARROT_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_class():
    src = open(os.path.join(REF, 'extensions.py')).read()
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    tree = ast.parse(str(tool.refactor_string(src + '\n', 'extensions.py')))
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'LearningRateSchedule'][0]

    class SimpleExtension(object):  # blocks.extensions.SimpleExtension: only the constructor is used
        def __init__(self, **kwargs):
            self.kwargs = kwargs

    class VariableFilter(object):   # blocks.filter.VariableFilter(roles=[ALGORITHM_BUFFER])(vars) -> the buffers
        def __init__(self, roles=None):
            pass

        def __call__(self, variables):
            return list(variables)

    events = []
    g = dict(numpy=numpy, SimpleExtension=SimpleExtension, VariableFilter=VariableFilter, ALGORITHM_BUFFER='buffer',
             LOADED_FROM='loaded_from', load_parameters=lambda source: events.append('reload') or {'p': 1},
             open=lambda path, mode='rb': io.BytesIO(b''))
    exec(compile(ast.Module(body=[cls], type_ignores=[]), 'extensions.py', 'exec'), g)
    return g['LearningRateSchedule'], events


class Shared(object):
    def __init__(self, v):
        self.v = numpy.asarray(v, dtype='float64')

    def get_value(self):
        return self.v

    def set_value(self, v):
        self.v = numpy.asarray(v, dtype='float64')


def trajectories():
    rs = numpy.random.RandomState(7)
    out = {}
    out['improve_then_flat'] = [float(x) for x in numpy.concatenate([numpy.linspace(5, 1, 8), 1.0 + 0.01 * rs.rand(70)])]
    out['noisy'] = [float(x) for x in 3.0 + rs.randn(90).cumsum() * 0.05]
    nan = [float(x) for x in numpy.linspace(4, 2, 12)]
    nan[5] = float('nan')
    out['nan_in_the_middle'] = nan + [2.0 + 0.001 * i for i in range(60)]
    out['always_better'] = [float(x) for x in numpy.linspace(9, 1, 40)]
    return out


def run(cls, events, values, patience, num_cuts, cut_size=.5):
    lr = Shared(1e-4)
    ext = cls(lr, 'valid_cost', '/nonexistent/best.tar', patience=patience, num_cuts=num_cuts, cut_size=cut_size)

    class Obj(object):
        pass
    ml = Obj()
    ml.log = Obj()
    ml.algorithm = Obj()
    bufs = [Shared(numpy.ones(3)), Shared(numpy.ones(2))]
    ml.algorithm.step_rule_updates = [(b, None) for b in bufs]
    ml.model = Obj()
    ml.model.set_parameter_values = lambda v: events.append('set')
    ext.main_loop = ml
    rows = []
    for v in values:
        ml.log.current_row = {'valid_cost': v}
        for b in bufs:
            b.set_value(numpy.ones_like(b.get_value()))
        del events[:]
        ext.do('after_batch')
        rows.append(dict(lr=float(lr.get_value()), cuts=int(ext.count_cuts), counter=int(ext.counter),
                         reloaded=('reload' in events and 'set' in events),
                         buffers_zeroed=bool(all(float(numpy.abs(b.get_value()).max()) == 0.0 for b in bufs)),
                         finish=bool(ml.log.current_row.get('training_finish_requested', False))))
        if rows[-1]['finish']:
            break
    return rows


def main():
    cls, events = reference_class()
    blob = {}
    for name, vals in trajectories().items():
        for (patience, num_cuts) in ((10, 5), (3, 2)):  # train.py:175-182 uses (10, 5)
            blob[f'{name}|p{patience}|c{num_cuts}'] = dict(values=[None if v != v else v for v in vals],
                                                           rows=run(cls, events, vals, patience, num_cuts))
    path = os.path.join(HERE, 'schedule_golden.json')
    json.dump(blob, open(path, 'w'))
    print({k: len(v['rows']) for k, v in blob.items()}, '->', path)


if __name__ == '__main__':
    main()
