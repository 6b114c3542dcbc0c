"""Row-block sharding of one image across the ranks of a torch.distributed job (one process per GPU).

The path shards with NO data-path collective during the render: every pixel is independent and keyed by its
absolute coordinates (pathTracer.comp:357,393; mandelbrot.comp:30-38), so rank r renders the interleaved
ROW_BLOCK-row blocks r, r+n, r+2n, ... of the storage buffer with the GLOBAL (W, H) and gets the same bits as a
single-GPU render.  One exchange step follows: the tiles are gathered to rank 0 (RCCL over xGMI when the backend is "nccl")
— the path tracer's fp32 vec4 tiles, re-assembled with mc_deinterleave_rows_device_async, or the RGBA8 tiles
every rank has converted itself, 4 B/pixel, which rank 0 de-interleaves and point-reflects with mc_assemble_rgba8_device_async
into the image saveRenderedImage would write; for the Mandelbrot only the
iteration counts (uint16 for max_iter <= 65535: 2 B/pixel instead of 16), from which rank 0 rebuilds the vec4 buffer through the
colour table (mc_mandelbrot_assemble_device_async: the colour is a function of the count alone, mandelbrot.comp:50-59).
Samples are never split across ranks: the fp32 accumulation order is part of the parity contract.
"""
import os
import sys

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


class Exchange:
    """The exchange step of a multi-rank render, shaped for a loop of steps: receive buffers allocated ONCE (the gather
    writes straight into views of one (n, rows_padded, W, ...) tensor — no per-step allocation, no torch.stack), two
    tile / receive buffer sets used alternately, and the gather issued asynchronously so that step i's exchange and rank 0's
    re-assembly (on a side stream) overlap step i + 1's render.  `assemble(recv, stream_handle)` runs on rank 0's side stream.
    (16-bit iteration counts travel as a uint8 tensor with a trailing dimension of 2: RCCL has no 16-bit integer type.)

      ex = Exchange(rank, n, tile_shape, dtype, device)
      for i in steps:  tile = ex.tile(i);  render into tile;  ex.submit(i, assemble)      # returns at once
      ex.finish()                                                                          # before reading `out` / timing

    RCCL (backend "nccl"): dist.gather(async_op=True) runs on the process group's own stream after the render stream's work
    queued so far; Work.wait() makes the CURRENT stream wait for it, so it is called under the side stream.  gloo
    (rehearsal, ranks sharing a GPU): the tile is staged through the host and the call is synchronous."""

    def __init__(self, rank, n, tile_shape, dtype, device, dst=0, exchange_when_alone=False):
        self.rank, self.n, self.dst = rank, n, dst
        # A world of one has nothing to exchange (bench.py at N = 1).  exchange_when_alone=True sends it through the collective
        # all the same — a gather with itself — so that the asynchronous RCCL branch below can be driven on a ONE-GPU box against the
        # real backend (tests/test_gpu_multi.py): the same calls, streams and events as with N > 1.
        self.active = n > 1 or exchange_when_alone
        self.gloo = self.active and dist.get_backend() == "gloo"
        self.tiles = [torch.zeros(tile_shape, dtype=dtype, device=device) for _ in range(2 if self.active else 1)]
        self.recv = ([torch.empty((n,) + tuple(tile_shape), dtype=dtype, device=device) for _ in range(2)]
                     if self.active and rank == dst else None)
        self.side = torch.cuda.Stream(device=device) if (self.active and torch.device(device).type == "cuda") else None
        self.pending = [None, None]
        self.bytes_per_rank = self.tiles[0].numel() * self.tiles[0].element_size()
        # MC_BENCH_SYNC_EXCHANGE=1 (or an exception from the asynchronous path, reported on stderr): the same collective issued
        # synchronously on the render stream — no overlap, same bytes, same result.  Which one ran is reported by bench.py, and a
        # FALLBACK (`fell_back`: the asynchronous path was meant to run and raised) makes bench.py exit non-zero unless
        # --allow-sync-exchange was given: the collectives are completed first so that no rank is left waiting in one.
        self.sync_mode = os.environ.get("MC_BENCH_SYNC_EXCHANGE", "0") == "1"
        self.fell_back = False
        self.fallback_error = None

    def tile(self, i):
        """The tile buffer of step i.  The render stream is made to wait (on the device, not the host) for the exchange that
        last used this buffer set, two steps ago: its gather read the tile, its re-assembly read the receive buffer."""
        k = i % len(self.tiles)
        if self.pending[k] is not None:
            torch.cuda.current_stream().wait_event(self.pending[k])
            self.pending[k] = None
        return self.tiles[k]

    def _stream_handle(self):
        return torch.cuda.current_stream().cuda_stream if self.tiles[0].is_cuda else 0

    def submit(self, i, assemble=None, done_event=None):
        if not self.active:
            return
        k = i % 2
        tile = self.tiles[k]
        if self.gloo:   # (tiles on the GPU: staged through the host; tiles on the CPU — the CPU tests — gathered as they are)
            host = tile.cpu()
            bufs = list(torch.empty((self.n,) + tuple(host.shape), dtype=host.dtype).unbind(0)) if self.rank == self.dst else None
            dist.gather(host, bufs, dst=self.dst)
            if self.rank == self.dst:
                self.recv[k].copy_(torch.stack(bufs))
                if assemble is not None:
                    assemble(self.recv[k], self._stream_handle())
            if done_event is not None:
                done_event.record()
            return
        bufs = list(self.recv[k].unbind(0)) if self.rank == self.dst else None
        if not self.sync_mode:
            work = None
            try:
                work = dist.gather(tile, bufs, dst=self.dst, async_op=True)
                ev = torch.cuda.Event()
                with torch.cuda.stream(self.side):
                    work.wait()                 # the side stream waits for the collective; the render stream does not
                    if self.rank == self.dst and assemble is not None:
                        assemble(self.recv[k], self.side.cuda_stream)
                    ev.record()
                    if done_event is not None:
                        done_event.record()
                self.pending[k] = ev
                return
            except Exception as e:              # noqa: BLE001 — complete the collective (every rank is in it); bench.py then fails the job
                print(f"[sharding.Exchange] asynchronous exchange failed on rank {self.rank} ({e!r}); continuing synchronously",
                      file=sys.stderr, flush=True)
                self.sync_mode = True
                self.fell_back = True
                self.fallback_error = repr(e)
                if work is not None:            # the collective was issued: complete it on the render stream
                    work.wait()
                    if self.rank == self.dst and assemble is not None:
                        assemble(self.recv[k], torch.cuda.current_stream().cuda_stream)
                    if done_event is not None:
                        done_event.record()
                    return
        dist.gather(tile, bufs, dst=self.dst)
        if self.rank == self.dst and assemble is not None:
            assemble(self.recv[k], torch.cuda.current_stream().cuda_stream)
        if done_event is not None:
            done_event.record()

    def finish(self):
        for k in range(2):
            if self.pending[k] is not None:
                self.pending[k].synchronize()
                self.pending[k] = None
        if self.side is not None:
            self.side.synchronize()


def assemble_device(ctx, gathered, width, height, n, out, stream=0, block=ROW_BLOCK):
    """Rank 0, on the GPU: gathered (n, rows_padded, W, C) -> out (H, W, C) in storage-row order."""
    if not gathered.is_cuda or not out.is_cuda:
        raise RuntimeError("assemble_device needs device tensors (there is no CPU path in the product)")
    bpp = gathered.element_size() * gathered.shape[-1] if gathered.dim() == 4 else gathered.element_size()
    ctx.deinterleave_rows_device(gathered.data_ptr(), width, height, n, block, gathered.shape[1], bpp, out.data_ptr(), stream)
    return out


def assemble_mandelbrot_device(ctx, p, gathered, n, out_rgba, out_iters=None, stream=0, block=ROW_BLOCK):
    """Rank 0, on the GPU: gathered (n, rows_padded, W) int32 iteration counts — or (n, rows_padded, W, 2) uint8 holding 16-bit
    counts — -> the vec4 storage buffer out_rgba (H, W, 4) through the colour table, and optionally the int32 count plane."""
    if not gathered.is_cuda:
        raise RuntimeError("assemble_mandelbrot_device needs device tensors (there is no CPU path in the product)")
    iters_bytes = 2 if gathered.dim() == 4 else gathered.element_size()
    ctx.mandelbrot_assemble_device(p, gathered.data_ptr(), iters_bytes, n, block, gathered.shape[1],
                                   out_rgba.data_ptr() if out_rgba is not None else 0,
                                   out_iters.data_ptr() if out_iters is not None else 0, stream)
    return out_rgba
