"""Row-block sharding of one image across the ranks of a torch.distributed job (one process per GPU).

The path shards with NO data-path collective during the render: every pixel is independent and keyed by its
absolute coordinates (pathTracer.comp:357,393; mandelbrot.comp:30-38), so rank r renders the interleaved
ROW_BLOCK-row blocks r, r+n, r+2n, ... of the storage buffer with the GLOBAL (W, H) and gets the same bits as a
single-GPU render.  One exchange step follows: the fp32 tiles are gathered to rank 0 (RCCL over xGMI when the
backend is "nccl"), which re-assembles the storage buffer with mc_deinterleave_rows_device_async.
Samples are never split across ranks: the fp32 accumulation order is part of the parity contract.
"""
import torch
import torch.distributed as dist

from . import bindings as _b

ROW_BLOCK = int(_b.lib().mc_row_block())   # the library's one interleave-block height (csrc/mc_internal.h kRowBlock = 8)


def shard(p, rank, n, block=ROW_BLOCK):
    """Restricts params `p` (mc_mandelbrot_params / mc_pathtrace_params mirror) to rank's interleaved row blocks."""
    if n == 1:
        p.row_begin, p.row_end, p.row_block, p.row_stride = 0, p.height, 0, 0
    else:
        p.row_begin, p.row_end, p.row_block, p.row_stride = rank * block, p.height, block, block * n
    return p


def owns_rows(p):
    """False for a rank whose first row block lies beyond the image (more ranks than blocks): it launches nothing and
    contributes only padding to the gather."""
    return p.row_begin < p.row_end


def rank_rows(height, rank, n, block=ROW_BLOCK):
    """Storage rows owned by `rank`, in tile order."""
    return [r for r in range(height) if (r // block) % n == rank]


def padded_tile_rows(height, n, block=ROW_BLOCK):
    """Rows of the largest tile (rank 0's): every rank pads its tile to this so the gather is equal-sized."""
    return len(rank_rows(height, 0, n, block))


def gather_tiles(tile, rank, n, dst=0):
    """Gathers the equal-sized tiles to `dst`; returns an (n, *tile.shape) tensor there, None elsewhere."""
    if n == 1:
        return tile.unsqueeze(0)
    if tile.is_cuda and dist.get_backend() == "gloo":
        # rehearsal mode (several ranks sharing one GPU, no RCCL): stage through host memory
        host = tile.cpu()
        bufs = [torch.empty_like(host) for _ in range(n)] if rank == dst else None
        dist.gather(host, bufs, dst=dst)
        return torch.stack(bufs).to(tile.device) if rank == dst else None
    bufs = [torch.empty_like(tile) for _ in range(n)] if rank == dst else None
    dist.gather(tile, bufs, dst=dst)
    return torch.stack(bufs) if rank == dst else None


def assemble_device(ctx, gathered, width, height, n, out, stream=0, block=ROW_BLOCK):
    """Rank 0, on the GPU: gathered (n, rows_padded, W, C) -> out (H, W, C) in storage-row order."""
    if not gathered.is_cuda or not out.is_cuda:
        raise RuntimeError("assemble_device needs device tensors (there is no CPU path in the product)")
    bpp = gathered.element_size() * gathered.shape[-1] if gathered.dim() == 4 else gathered.element_size()
    ctx.deinterleave_rows_device(gathered.data_ptr(), width, height, n, block, gathered.shape[1], bpp, out.data_ptr(), stream)
    return out
