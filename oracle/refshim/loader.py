"""Execute reference sources UNMODIFIED on top of the eager shims (TEST INFRASTRUCTURE).

The reference is Python 2.  Nothing is copied into this repository: a file is read from /root/reference at fixture
generation time, translated IN MEMORY and executed:
  * files that do not parse under Python 3 (print statements, `except E, e`, ...) go through lib2to3's fixers;
  * every `/` is rewritten to a call that floors when both operands are ints (Python-2 division: e.g.
    `DIM * BIG_FRAME_SIZE / FRAME_SIZE`, three_tier.py:356; `dim = mu.shape[-1] / k`, model.py:97);
  * `xrange`, `reduce`, `unicode`, `long` are provided as module globals.
`install()` puts the shim modules (theano, blocks, lasagne stub) into sys.modules; `load_sample_rnn()` /
`load_model()` build the reference's module graph (lib, lib.ops, models.conditional.three_tier, model)."""
from __future__ import annotations

import ast
import functools
import io
import os
import sys
import types

REF = os.environ.get('PARROT_REFERENCE', '/root/reference')


def _py2div(a, b):
    if isinstance(a, int) and isinstance(b, int) and not isinstance(a, bool):
        return a // b
    return a / b


class _Div(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(ast.Call(func=ast.Name(id='__py2div__', ctx=ast.Load()),
                                              args=[node.left, node.right], keywords=[]), node)
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            load = ast.fix_missing_locations(ast.parse(ast.unparse(node.target), mode='eval')).body
            call = ast.Call(func=ast.Name(id='__py2div__', ctx=ast.Load()), args=[load, node.value], keywords=[])
            return ast.copy_location(ast.Assign(targets=[node.target], value=call), node)
        return node


def _to_py3(src, path):
    try:
        ast.parse(src)
        return src
    except SyntaxError:
        pass
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        from lib2to3 import refactor
        fixers = [f for f in refactor.get_fixers_from_package('lib2to3.fixes') if not f.endswith('fix_import')]
        tool = refactor.RefactoringTool(fixers)
        return str(tool.refactor_string(src if src.endswith('\n') else src + '\n', path))


def exec_reference(relpath, modname, extra_globals=None, quiet=True):
    """Reads REF/relpath, translates in memory, executes it as module `modname` (registered in sys.modules)."""
    path = os.path.join(REF, relpath)
    with open(path) as f:
        src = f.read()
    tree = ast.fix_missing_locations(_Div().visit(ast.parse(_to_py3(src, path), filename=path)))
    mod = sys.modules.get(modname)
    if mod is None:
        mod = types.ModuleType(modname)
        sys.modules[modname] = mod
    mod.__file__ = path
    g = mod.__dict__
    g.update(__py2div__=_py2div, xrange=range, reduce=functools.reduce, unicode=str, long=int)
    if extra_globals:
        g.update(extra_globals)
    code = compile(tree, path, 'exec')
    if quiet:
        old = sys.stdout
        sys.stdout = io.StringIO()
        try:
            exec(code, g)
        finally:
            sys.stdout = old
    else:
        exec(code, g)
    return mod


def install(floatX='float64'):
    from . import blocks_shim, theano_shim
    theano_shim.config.floatX = floatX
    for name, m in {**theano_shim.build_modules(), **blocks_shim.build_modules()}.items():
        sys.modules[name] = m
    sys.modules.setdefault('lasagne', types.ModuleType('lasagne'))
    return sys.modules['theano']


def load_sample_rnn():
    """lib (sampleRNN/lib/__init__.py), lib.ops (sampleRNN/lib/ops.py) and models.conditional.three_tier, all the
    reference's own code.  Returns (lib, ops, three_tier)."""
    if 'theano' not in sys.modules or not hasattr(sys.modules['theano'], 'OrderedUpdates'):
        install()
    lib = types.ModuleType('lib')
    lib.__path__ = [os.path.join(REF, 'sampleRNN', 'lib')]
    sys.modules['lib'] = lib
    ops = exec_reference('sampleRNN/lib/ops.py', 'lib.ops')
    sys.modules['ops'] = ops          # `import ops` (implicit relative import, lib/__init__.py:1)
    os.environ.setdefault('MPLBACKEND', 'Agg')
    exec_reference('sampleRNN/lib/__init__.py', 'lib')
    lib.ops = ops
    for pkg in ('models', 'models.conditional'):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    argv = sys.argv
    sys.argv = ['three_tier.py']
    try:
        tt = exec_reference('sampleRNN/models/conditional/three_tier.py', 'models.conditional.three_tier')
    finally:
        sys.argv = argv
    sys.modules['models'].conditional = sys.modules['models.conditional']
    sys.modules['models.conditional'].three_tier = tt
    return lib, ops, tt


def load_model():
    """The reference's model.py (Parrot, Encoder, ...) on the Blocks / Theano shims.  Returns the module."""
    if 'models.conditional.three_tier' not in sys.modules:
        load_sample_rnn()
    return exec_reference('model.py', 'reference_model')


def quiet_call(fn, *a, **k):
    """Calls fn with stdout silenced (lib.floatX prints a warning per call when floatX is float64)."""
    old = sys.stdout
    sys.stdout = io.StringIO()
    try:
        return fn(*a, **k)
    finally:
        sys.stdout = old
