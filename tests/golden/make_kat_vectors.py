#!/usr/bin/env python3
"""
Extract the known-answer vectors held by the reference's own unit tests
(/root/reference/modules/unit_tests.py) into tests/golden/kat_vectors.json.

The reference cannot be imported here (xarray / metpy / pint / numba are not
installed: plain ModuleNotFoundError), so instead of running it this script
reads unit_tests.py as text, and with `ast` evaluates ONLY
  * the literal input arrays each test builds (vert_array([...]), np.array([...])
    + 273.15, xarray.DataArray(<number>)), and
  * the expected values / decimals of each assert_almost_equal /
    assert_array_almost_equal / assert(np.isnan(..)) statement,
plus the constant keyword arguments (lcl_interp=..., depth=...,
virtual_temperature_correction=...) of the calls under test.  The output is data
(inputs and expected outputs); no reference source text is stored.

Run once in the build container:  python tests/golden/make_kat_vectors.py
(/root/reference does not exist on the GPU box; the JSON is what travels.)
"""
import ast
import json
import os
import sys

import numpy as np

REF = '/root/reference/modules/unit_tests.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'kat_vectors.json')


class _Arr(np.ndarray):
    def expand_dims(self, *a, **k):
        return self


def _vert_array(x, units=None):
    return np.asarray(x, dtype=np.float64).view(_Arr)


class _XR:
    @staticmethod
    def DataArray(x, *a, **k):
        return np.asarray(x, dtype=np.float64)


SAFE = {'np': np, 'vert_array': _vert_array, 'xarray': _XR, '__builtins__': {}}


def _eval(node, env):
    code = compile(ast.Expression(body=node), '<kat>', 'eval')
    return eval(code, dict(SAFE), env)


def _jsonable(v):
    a = np.asarray(v, dtype=np.float64)
    if a.ndim == 0:
        x = float(a)
        return None if np.isnan(x) else x
    return [None if np.isnan(x) else float(x) for x in a.ravel()]


def _src(node):
    return ast.unparse(node)


def extract(fn, defaults_env, helpers):
    env = dict(defaults_env)
    inputs, expected, calls = {}, [], []
    for st in fn.body:
        if isinstance(st, ast.Assign):
            # tuple unpack from a helper that returns literal arrays
            if (isinstance(st.value, ast.Call) and isinstance(st.value.func, ast.Name)
                    and st.value.func.id in helpers and isinstance(st.targets[0], ast.Tuple)):
                vals = helpers[st.value.func.id]
                for t, v in zip(st.targets[0].elts, vals):
                    env[t.id] = v
                    inputs[t.id] = _jsonable(v)
                continue
            if len(st.targets) == 1 and isinstance(st.targets[0], ast.Name):
                name = st.targets[0].id
                try:
                    v = _eval(st.value, env)
                    if isinstance(v, (int, float, np.ndarray, list)):
                        env[name] = v
                        inputs[name] = _jsonable(v)
                        continue
                except Exception:
                    pass
            # a call under test: keep only its constant keyword arguments
            for call in [n for n in ast.walk(st.value) if isinstance(n, ast.Call)]:
                f = call.func
                if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id == 'parcel':
                    ck = {}
                    for kw in call.keywords:
                        if isinstance(kw.value, ast.Constant):
                            ck[kw.arg] = kw.value.value
                    calls.append({'func': f.attr, 'const_kwargs': ck})
        elif isinstance(st, ast.Expr) and isinstance(st.value, ast.Call):
            c = st.value
            fname = c.func.id if isinstance(c.func, ast.Name) else getattr(c.func, 'attr', '')
            if fname in ('assert_almost_equal', 'assert_array_almost_equal'):
                args = list(c.args)
                dec = 7 if fname == 'assert_almost_equal' else 6
                if len(args) > 2:
                    dec = _eval(args[2], env)
                for kw in c.keywords:
                    if kw.arg == 'decimal':
                        dec = _eval(kw.value, env)
                what = _src(args[0])
                if what.endswith('.values'):
                    what = what[:-7]
                try:
                    val = _jsonable(_eval(args[1], env))
                except Exception:
                    continue      # e.g. the broken pint-based test (unit_tests.py:1131-1140)
                expected.append({'what': what, 'value': val, 'decimal': int(dec)})
            elif fname == 'assert_array_equal':
                expected.append({'what': _src(c.args[0]), 'value': _jsonable(_eval(c.args[1], env)),
                                 'decimal': 15})
        elif isinstance(st, ast.Assert):
            t = st.test
            if (isinstance(t, ast.Call) and isinstance(t.func, ast.Attribute) and t.func.attr == 'isnan'):
                expected.append({'what': _src(t.args[0]), 'value': None, 'decimal': 0})
    return {'inputs': inputs, 'expected': expected, 'calls': calls}


def main():
    if not os.path.exists(REF):
        sys.exit('reference not mounted; kat_vectors.json is the committed artefact')
    tree = ast.parse(open(REF).read())
    fns = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    helpers = {}
    # helper returning (levels, temperatures, dewpoints)
    h = fns['multiple_intersections']
    env = {}
    for st in h.body:
        if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
            try:
                env[st.targets[0].id] = _eval(st.value, env)
            except Exception:
                pass
    helpers['multiple_intersections'] = (env['levels'], env['temperatures'], env['dewpoints'])
    out = {}
    for name, fn in fns.items():
        if not name.startswith('test_'):
            continue
        defaults = {}
        nd = len(fn.args.defaults)
        for a, d in zip(fn.args.args[len(fn.args.args) - nd:], fn.args.defaults):
            defaults[a.arg] = ast.literal_eval(d)
        rec = extract(fn, defaults, helpers)
        out[name] = rec
    # the run_moist_lapse_tests_looser variant (unit_tests.py:106-112)
    out['_meta'] = {'source': 'traupach/xarray_parcel modules/unit_tests.py @ 2025-07-18',
                    'table_mode_decimals': {'test_moist_lapse_uniform': 2},
                    'note': 'value null = NaN expected; decimal d means |actual-expected| < 1.5*10^-d'}
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote', OUT, len(out) - 1, 'tests')


if __name__ == '__main__':
    main()
