"""
Deterministic synthetic soundings.

Every value is a pure function of (seed, global column id, level), through a
counter-based hash, so a shard of the grid generated on one rank equals the same
slice of the whole grid generated anywhere else.  The reference's own data file
(test_data.nc, Aus400 subset) is not available (.MISSING_LARGE_BLOBS), so this
generator also provides the "Aus400-like" stand-in for config c1.

Layout: arrays are (nlev, ncol), level 0 = surface, pressure strictly decreasing,
units hPa / K / K -- the input contract of the reference (README.md:9,
parcel_functions.py:2319-2320).
"""
import numpy as np

P_TOP = 60.0
RD_OVER_G = 287.04749097718457 / 9.80665
N_UNIFORMS = 10
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_M64)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_M64)
    return z ^ (z >> np.uint64(31))


def column_uniforms(ncol, seed, col_offset=0):
    """(N_UNIFORMS, ncol) float64 in [0,1): hash of (seed, global column id, slot)."""
    with np.errstate(over='ignore'):
        ids = (np.arange(ncol, dtype=np.uint64) + np.uint64(col_offset))
        base = _splitmix64(ids ^ _splitmix64(np.full(ncol, seed, dtype=np.uint64)))
        out = np.empty((N_UNIFORMS, ncol), dtype=np.float64)
        for j in range(N_UNIFORMS):
            h = _splitmix64(base + np.uint64(j + 1) * np.uint64(0xD1B54A32D192ED03))
            out[j] = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return out


def column_uniforms_smooth(ncol, seed, col_offset=0, nx=1024, noise=0.005):
    """Spatially smooth variant of column_uniforms for sensitivity runs (NOT the headline data): every slot is a
    low-wavenumber wave over the (y, x) grid of row length nx plus a little hashed noise, so neighbouring columns --
    the lanes of a wavefront -- resemble each other the way model fields do."""
    ids = np.arange(ncol, dtype=np.float64) + float(col_offset)
    x, y = np.mod(ids, nx) / nx, np.floor(ids / nx) / nx
    h = column_uniforms(ncol, seed, col_offset)
    rng = np.random.default_rng(seed)
    out = np.empty((N_UNIFORMS, ncol), dtype=np.float64)
    for j in range(N_UNIFORMS):
        fx, fy, ph = rng.uniform(0.2, 0.8), rng.uniform(0.5, 3.0), rng.uniform(0, 2 * np.pi)
        out[j] = np.clip(0.5 + 0.47 * np.sin(2 * np.pi * (fx * x + fy * y) + ph) + noise * (h[j] - 0.5), 0.0, 1.0 - 1e-12)
    return out


def _fields(u, nlev, xp, k_index, saturate_some):
    """Shared formula; xp is numpy or torch.  u: (N_UNIFORMS, ncol); k_index: (nlev, 1) float."""
    p_sfc = 960.0 + 75.0 * u[0]
    sigma = 1.0 - (k_index / float(max(nlev - 1, 1))) ** 1.3
    p = P_TOP + (p_sfc - P_TOP) * sigma
    t_sfc = 283.0 + 26.0 * u[1]
    gamma = (5.5 + 3.0 * u[2]) * 1e-3
    t = t_sfc * (p / p_sfc) ** (RD_OVER_G * gamma)
    t = xp.maximum(t, t * 0.0 + 205.0)
    amp = (1.0 + 5.0 * u[4]) * (u[3] < 0.25)
    pc = 800.0 + 100.0 * u[5]
    t = t + amp * xp.exp(-((p - pc) / 25.0) ** 2)
    dd_sfc = 0.3 + 13.7 * u[6]
    if saturate_some:
        dd_sfc = dd_sfc * (u[8] >= 0.03)          # ~3 % saturated surface parcels (LCL == surface)
    frac = xp.log(p_sfc / p) / xp.log(p_sfc / P_TOP)
    td = t - (dd_sfc + (30.0 - dd_sfc) * frac)
    return p, t, td


def columns(nlev, ncol, seed=20250718, nan_fraction=0.0, dtype=np.float64, col_offset=0,
            saturate_some=None, nan_pressure_fraction=0.0):
    """NumPy soundings (nlev, ncol).  nan_fraction > 0 switches on the correctness-run extras:
    that fraction of columns gets NaN temperature/dewpoint levels (or a NaN surface, or is
    NaN throughout), and ~3 % of surface parcels are saturated.  nan_pressure_fraction > 0 (opt-in, outside the
    reference's input contract) additionally blanks ONE interior pressure level in that fraction of columns."""
    u = column_uniforms(ncol, seed, col_offset)
    if saturate_some is None:
        saturate_some = nan_fraction > 0
    k = np.arange(nlev, dtype=np.float64)[:, None]
    p, t, td = _fields(u, nlev, np, k, saturate_some)
    if nan_fraction > 0:
        affected = u[7] < nan_fraction
        kind = u[9]
        with np.errstate(over='ignore'):
            ids = (np.arange(ncol, dtype=np.uint64) + np.uint64(col_offset))[None, :]
            lev = np.arange(nlev, dtype=np.uint64)[:, None]
            h = _splitmix64(ids * np.uint64(1000003) + lev + np.uint64(seed))
        r = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        sprinkle = affected[None, :] & (kind[None, :] >= 0.25) & (r < 0.15)
        sfc = affected & (kind < 0.15)
        allnan = affected & (kind >= 0.15) & (kind < 0.25)
        t = np.where(sprinkle, np.nan, t)
        td = np.where(sprinkle, np.nan, td)
        t[0, sfc] = np.nan
        td[0, sfc] = np.nan
        t[:, allnan] = np.nan
        td[:, allnan] = np.nan
    if nan_pressure_fraction > 0:
        hit = column_uniforms(ncol, seed + 77, col_offset)
        cols = np.nonzero(hit[0] < nan_pressure_fraction)[0]
        levs = 1 + (hit[1][cols] * max(nlev - 2, 1)).astype(np.int64)          # never the surface level
        p = p.copy()
        p[np.minimum(levs, nlev - 1), cols] = np.nan
    p = np.ascontiguousarray(p.astype(dtype))
    t = np.ascontiguousarray(t.astype(dtype))
    td = np.ascontiguousarray(td.astype(dtype))
    return p, t, td


def columns_torch(nlev, ncol, device, seed=20250718, dtype=None, col_offset=0, chunk=1 << 20, smooth=False, nx=1024):
    """Same soundings built directly in device memory (perf runs: no NaNs).  Per-column
    uniforms come from the host hash; the (nlev, ncol) fields are evaluated on the device
    in float64 and cast, chunked so the float64 temporaries stay small."""
    import torch
    dtype = dtype or torch.float64
    p = torch.empty((nlev, ncol), dtype=dtype, device=device)
    t = torch.empty_like(p)
    td = torch.empty_like(p)
    k = torch.arange(nlev, dtype=torch.float64, device=device)[:, None]
    for c0 in range(0, ncol, chunk):
        c1 = min(ncol, c0 + chunk)
        uf = column_uniforms_smooth(c1 - c0, seed, col_offset + c0, nx) if smooth else column_uniforms(c1 - c0, seed, col_offset + c0)
        u = torch.from_numpy(uf).to(device)
        pp, tt, dd = _fields(u, nlev, torch, k, False)
        p[:, c0:c1] = pp.to(dtype)
        t[:, c0:c1] = tt.to(dtype)
        td[:, c0:c1] = dd.to(dtype)
    return p, t, td
