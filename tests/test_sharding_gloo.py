"""CPU suite, part 4: the N > 1 path.  Two processes over gloo: each renders the bands of its rank, one
reduce(sum) assembles the film on rank 0, which must equal the single-rank film BIT FOR BIT (disjoint bands,
counter sampler).  The stand-in renderer on the CPU is the oracle; on the GPU box the same plumbing
(jet_pbrt_amd.distributed) drives the HIP context."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, Hh, spp, out_path):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import harness as H
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hb = H.SCENES["cornell"](H.scenes.HostBackend("g"), W, Hh)
    sp = hb.flatten()

    def render_fn(params):
        film, _ = H.oracle_render(sp, params, 2)
        own = H.jp.distributed.bands_of(Hh, params.shard_index, max(1, params.shard_count))
        mask = np.zeros(Hh, bool)
        for y0, y1 in own:
            mask[y0:y1] = True
        assert (film[~mask] == 0).all() and (film[mask].sum() > 0)       # zero outside the rank's bands
        return torch.from_numpy(film)

    film = H.jp.distributed.render_sharded(render_fn, W, Hh, spp, dist=dist)
    if rank == 0:
        np.save(out_path, film.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("W,Hh,spp", [(40, 70, 2), (40, 80, 2)])      # 70 rows: uneven bands -> reduce(sum); 80 rows: 4 bands -> packed gather
def test_two_rank_band_shard_equals_single_rank(H, tmp_path, W, Hh, spp):
    out = str(tmp_path / "film.npy")
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, W, Hh, spp, out), nprocs=2, join=True)
    sharded = np.load(out)
    hb = H.SCENES["cornell"](H.scenes.HostBackend("g"), W, Hh)
    full, _ = H.oracle_render(hb.flatten(), H.jp.render_params(W, Hh, spp), 2)
    assert np.array_equal(sharded.view(np.uint32), full.view(np.uint32))


def test_band_assignment(H):
    b0 = H.jp.distributed.bands_of(70, 0, 3); b1 = H.jp.distributed.bands_of(70, 1, 3); b2 = H.jp.distributed.bands_of(70, 2, 3)
    assert b0 == [(0, 20), (60, 70)] and b1 == [(20, 40)] and b2 == [(40, 60)]
    rows = sorted(y for bs in (b0, b1, b2) for (a, b) in bs for y in range(a, b))
    assert rows == list(range(70))


def test_balanced_band_rows_deals_evenly(H):
    """N > 1 bench runs use the largest band height <= 20 that gives every rank the same number of rows"""
    D = H.jp.distributed
    assert D.balanced_band_rows(512, 1) == 16 and D.balanced_band_rows(512, 8) == 16 and D.balanced_band_rows(600, 4) == 15
    assert D.balanced_band_rows(7, 3) == 20                       # nothing divides: the reference's 20
    for height, world in ((512, 2), (512, 4), (512, 8), (600, 8), (1024, 8)):
        b = D.balanced_band_rows(height, world)
        rows = [sum(y1 - y0 for y0, y1 in D.bands_of(height, r, world, b)) for r in range(world)]
        assert len(set(rows)) == 1 and sum(rows) == height and b <= 20
