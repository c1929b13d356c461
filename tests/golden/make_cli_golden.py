"""Reads the argparse definitions of the reference's entry points (utils.py: train_parse / sample_parse) from its AST and
commits flag names, types and literal defaults as tests/golden/cli_defaults.json, so that tests/test_cli_cpu.py can hold
parrot_amd/utils.py to "same flags, same defaults" without the reference being present (it is not on the GPU box).

    python tests/golden/make_cli_golden.py      # needs /root/reference (PARROT_REFERENCE overrides)
"""
import ast
import json
import os

REF = os.environ.get('PARROT_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))


def flags_of(fn):
    out = {}
    for node in ast.walk(fn):
        if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'add_argument'):
            continue
        name = node.args[0].value
        ent = {'type': None, 'default': '<none given>'}
        for kw in node.keywords:
            if kw.arg == 'type':
                ent['type'] = ast.unparse(kw.value)
            elif kw.arg == 'default':
                try:
                    ent['default'] = ast.literal_eval(kw.value)
                except Exception:
                    ent['default'] = '<expr> ' + ast.unparse(kw.value)
        out[name] = ent
    return out


def main():
    src = open(os.path.join(REF, 'utils.py')).read()
    try:
        tree = ast.parse(src)
    except SyntaxError:  # python-2 only syntax: translate in memory like oracle/refshim/loader.py does
        from lib2to3 import refactor
        tool = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
        tree = ast.parse(str(tool.refactor_string(src + '\n', 'utils.py')))
    blob = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ('train_parse', 'sample_parse'):
            blob[node.name] = flags_of(node)
    path = os.path.join(HERE, 'cli_defaults.json')
    json.dump(blob, open(path, 'w'), indent=1, sort_keys=True)
    print({k: len(v) for k, v in blob.items()}, '->', path)


if __name__ == '__main__':
    main()
