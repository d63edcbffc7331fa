"""
xarray access for the host-side mirror.  With xarray installed the real DataArray / Dataset are used.
xarray is not installable in the build or GPU images (no network), so a *minimal* stand-in with the
handful of members the mirror and its tests touch is provided; it is not a general xarray replacement.
"""
import numpy as np

try:  # pragma: no cover - not available in the build image
    import xarray as _xarray
    DataArray = _xarray.DataArray
    Dataset = _xarray.Dataset
    HAVE_XARRAY = True
except Exception:
    HAVE_XARRAY = False

    class DataArray:
        def __init__(self, data, dims=None, coords=None, attrs=None, name=None):
            self.values = np.asarray(data)
            if dims is None:
                dims = tuple(f'dim_{i}' for i in range(self.values.ndim))
            if isinstance(dims, str):
                dims = (dims,)
            self.dims = tuple(dims)
            assert len(self.dims) == self.values.ndim, 'dims do not match data'
            self.coords = {k: np.asarray(v) for k, v in (coords or {}).items()}
            self.attrs = dict(attrs or {})
            self.name = name

        shape = property(lambda self: self.values.shape)
        ndim = property(lambda self: self.values.ndim)
        dtype = property(lambda self: self.values.dtype)
        chunks = None

        def transpose(self, *dims):
            order = [self.dims.index(d) for d in dims]
            return DataArray(self.values.transpose(order), dims=dims, coords=self.coords, attrs=self.attrs,
                             name=self.name)

        def isel(self, indexers=None, **kw):
            out = self
            for d, i in {**(indexers or {}), **kw}.items():
                ax = out.dims.index(d)
                if isinstance(i, slice):
                    i = np.arange(out.shape[ax])[i]
                vals = np.take(out.values, i, axis=ax)
                coords = {k: (np.take(v, i, axis=0) if k == d else v) for k, v in out.coords.items()}
                if np.ndim(i) == 0:
                    dims = out.dims[:ax] + out.dims[ax + 1:]
                    coords = {k: v for k, v in coords.items() if k != d}
                else:
                    dims = out.dims
                out = DataArray(vals, dims=dims, coords=coords, attrs=out.attrs, name=out.name)
            return out

        def __getitem__(self, i):
            return self.isel({self.dims[0]: i})

        def _binary(self, other, op):
            o = other.values if isinstance(other, DataArray) else other
            if isinstance(other, DataArray):
                assert other.dims == self.dims, 'the stand-in does not broadcast by dimension name'
            return DataArray(op(self.values, o), dims=self.dims, coords=self.coords, attrs={}, name=self.name)

        def __add__(self, o): return self._binary(o, np.add)
        def __radd__(self, o): return self._binary(o, np.add)
        def __sub__(self, o): return self._binary(o, np.subtract)
        def __rsub__(self, o): return self._binary(o, lambda a, b: b - a)
        def __mul__(self, o): return self._binary(o, np.multiply)
        def __rmul__(self, o): return self._binary(o, np.multiply)
        def __truediv__(self, o): return self._binary(o, np.divide)

        def __array__(self, dtype=None, copy=None):
            return self.values if dtype is None else self.values.astype(dtype)

        def __float__(self):
            return float(self.values)

        def __repr__(self):
            return f'<DataArray {self.name} {dict(zip(self.dims, self.shape))}>'

    class Dataset:
        def __init__(self, data_vars=None, attrs=None):
            self._vars = dict(data_vars or {})
            self.attrs = attrs if attrs is not None else {}

        def __getitem__(self, k):
            return self._vars[k]

        def __setitem__(self, k, v):
            self._vars[k] = v

        def __getattr__(self, k):
            try:
                return self.__dict__['_vars'][k]
            except KeyError:
                raise AttributeError(k)

        def __contains__(self, k):
            return k in self._vars

        def keys(self):
            return self._vars.keys()

        def rename(self, mapping):
            return Dataset({mapping.get(k, k): v for k, v in self._vars.items()}, attrs=self.attrs)

        def __repr__(self):
            return f'<Dataset {list(self._vars)}>'


def merge(objs):
    """xarray.merge for the two cases the mirror needs: Datasets and named DataArrays."""
    if HAVE_XARRAY:  # pragma: no cover
        return _xarray.merge(objs)
    out = Dataset()
    for o in objs:
        if isinstance(o, Dataset):
            for k in o.keys():
                out[k] = o[k]
            out.attrs.update(o.attrs if isinstance(o.attrs, dict) else {})
        else:
            out[o.name] = o
    return out
