"""World-size-2 rehearsal of the sharded path on CPU (gloo): slab split + single gather reproduce the
single-process result.  The compute step is a stand-in here (there is no CPU product path); the GPU version
of the same code path is what bench.py --gpus N runs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NLEV, NY, NX = 24, 9, 16          # 9 rows: uneven slabs (5 + 4)


def _stand_in(p, t, td, want=None, **kw):
    from oracle import c_oracle
    sh = p.shape[1:]
    r = c_oracle.cape_cin_grid(np.asarray(p).reshape(p.shape[0], -1), np.asarray(t).reshape(p.shape[0], -1),
                               np.asarray(td).reshape(p.shape[0], -1), moist='rk4', nthreads=1)
    return {k: r[k].reshape(sh) for k in want}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from xarray_parcel_amd import distributed as D
    from xarray_parcel_amd import synth
    y0, y1 = D.slab_bounds(NY, world, rank)
    p, t, td = synth.columns(NLEV, (y1 - y0) * NX, seed=5, col_offset=y0 * NX)
    sh = (NLEV, y1 - y0, NX)
    out = D.sharded_cape_cin(p.reshape(sh), t.reshape(sh), td.reshape(sh), names=('cape', 'cin', 'lfc_index'),
                             compute=_stand_in)
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_slab_bounds_cover_the_grid():
    from xarray_parcel_amd import distributed as D
    for n in (1, 7, 8, 1024, 24 * 2048):
        for w in (1, 2, 3, 8):
            b = [D.slab_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


def test_two_rank_gather_equals_single_process():
    from xarray_parcel_amd import synth
    os.environ['PYTHONPATH'] = ROOT + os.pathsep + os.environ.get('PYTHONPATH', '')   # children import tests.*
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=90)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    p, t, td = synth.columns(NLEV, NY * NX, seed=5)
    ref = _stand_in(p, t, td, want=('cape', 'cin', 'lfc_index'))
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
