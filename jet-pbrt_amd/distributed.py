"""Multi-GPU plumbing: one process per GPU, 20-row bands of the film dealt round-robin to the ranks, one
collective that assembles the film on rank 0 (RCCL over xGMI on the GPU box; gloo in the CPU tests): `assemble_bands`
-- every rank sends only ITS rows, packed (1/N of the film per rank; round 3) -- when the bands deal evenly, else one
reduce(sum) of the per-rank films.

The path shards by pixels with no data-path exchange: a rank's film is zero outside its
bands, so the sum over ranks is the disjoint union and -- with the counter sampler -- bit-identical to the
single-rank film.  `render_fn(params) -> torch tensor (H, W, 3) float32` is supplied by the caller: the HIP
context on the GPU box, any stand-in renderer in CPU tests.
"""
from . import render_params


def shard_params(width, height, spp, rank, world, max_depth=5, seed=1234, band_rows=20):
    """JpRenderParams of rank `rank` of `world`: band b belongs to rank b % world (integrator.cc:53 bands)."""
    return render_params(width, height, spp, max_depth, seed, band_rows=band_rows, shard_index=rank, shard_count=world)


def balanced_band_rows(height, world, max_rows=20):
    """Largest band height <= max_rows (the reference's lines_per_task, integrator.cc:53) that deals the film evenly:
    it divides `height` and gives a band count that is a multiple of `world`.  With the counter sampler the film does
    not depend on the band height, only the load balance does (512 rows on 8 ranks: 26 bands of 20 rows would give
    two ranks 4 bands and six ranks 3; 32 bands of 16 rows give every rank 4).  Falls back to max_rows."""
    for b in range(max_rows, 0, -1):
        if height % b == 0 and (height // b) % world == 0:
            return b
    return max_rows


def bands_of(height, rank, world, band_rows=20):
    nb = (height + band_rows - 1) // band_rows
    return [(b * band_rows, min(height, (b + 1) * band_rows)) for b in range(nb) if b % world == rank]


def render_sharded(render_fn, width, height, spp, max_depth=5, seed=1234, band_rows=20, dist=None):
    """Every rank renders its bands; the films are summed onto rank 0.  Returns the full film on rank 0, None elsewhere."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return render_fn(shard_params(width, height, spp, 0, 1, max_depth, seed, band_rows))
    rank, world = dist.get_rank(), dist.get_world_size()
    film = render_fn(shard_params(width, height, spp, rank, world, max_depth, seed, band_rows))
    return assemble_bands(film, height, band_rows, dist)


def assemble_bands(film, height, band_rows, dist):
    """The film of this rank (H, W, 3; zero outside its bands) -> the full film on rank 0 (None elsewhere), moving only the rank's own
    rows: the bands b % world == rank are packed into one contiguous (rows/world, W, 3) tensor, one gather brings the packed shards
    to rank 0, which writes them back to their rows.  1/N of the film per rank on the wire instead of the whole film per rank of a
    reduce(sum) (FFilm::AddColor onto a zero film, film.h:64-68: disjoint bands, nothing to add).  Needs bands that deal evenly
    (balanced_band_rows); otherwise the reduce of render_sharded serves."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    nb = height // band_rows
    if height % band_rows != 0 or nb % world != 0:
        dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
        return film if rank == 0 else None
    W = film.shape[1]
    view = film.view(nb // world, world, band_rows, W, 3)            # band b = (b // world, b % world)
    mine = view[:, rank].contiguous()
    parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, gather_list=parts, dst=0)
    if rank != 0:
        return None
    out = torch.empty_like(film)
    ov = out.view(nb // world, world, band_rows, W, 3)
    for r in range(world):
        ov[:, r] = parts[r]
    return out
