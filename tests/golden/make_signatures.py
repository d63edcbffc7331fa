"""Extract the call signatures (argument names and literal defaults) of the reference's public functions from
modules/parcel_functions.py and modules/parcel_test.py into tests/golden/reference_signatures.json.  Data only: names and
default values, no source text.  Run in the build container (the reference is not available on the GPU box)."""
import ast
import json
import os

REF = '/root/reference/modules'
out = {}
for mod in ('parcel_functions', 'parcel_test'):
    tree = ast.parse(open(os.path.join(REF, mod + '.py')).read())
    sig = {}
    for node in tree.body:
        if not isinstance(node, ast.FunctionDef):
            continue
        a = node.args
        names = [x.arg for x in a.args]
        defaults = [None] * (len(names) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
        sig[node.name] = {'args': names, 'defaults': defaults, 'kwargs': a.kwarg.arg if a.kwarg else None,
                          'line': node.lineno}
    out[mod] = sig
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_signatures.json')
json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
print(path, {k: len(v) for k, v in out.items()})
