"""world_size 2 / 4 / 8 tests of the N > 1 path on CPU (gloo): the sharding helpers bench.py uses, with the oracle
standing in for the renderer.  Checks that interleaved row-block tiles rendered independently, gathered to
rank 0 and re-assembled equal the single-process image bit for bit, for both hot paths — including heights that
leave a partial last block (601, 70) and worlds with more ranks than row blocks (ranks that own zero rows)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _deinterleave_numpy(gathered, H, n, block):
    """numpy mirror of deinterleave_rows_kernel (csrc/postprocess.hip): storage row r <- tile t, tile row k*B + j."""
    out = np.empty((H,) + gathered.shape[2:], gathered.dtype)
    for r in range(H):
        blk, j = divmod(r, block)
        t, k = blk % n, blk // n
        out[r] = gathered[t, k * block + j]
    return out


def _assemble_rgba8_numpy(gathered, W, H, n, block, rotate180):
    """numpy mirror of assemble_rgba8_kernel (csrc/postprocess.hip): output pixel (y, x) <- storage pixel (H-1-y, W-1-x), except an odd
    width's middle column (pathtracerApp.h:236-243 swaps x < W/2 only), each storage row taken from its owner's tile."""
    storage = _deinterleave_numpy(gathered, H, n, block)
    if not rotate180:
        return storage
    out = storage[::-1, ::-1].copy()
    if W % 2:
        out[:, W // 2] = storage[:, W // 2]
    return out


def _worker(rank, world, port, W, H, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    S = entry.load_package().sharding
    B = entry.load_package().bindings
    O = entry.load_oracle()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        padded = S.padded_tile_rows(H, world)
        rows = S.rank_rows(H, rank, world)
        # the C-ABI's own tile arithmetic agrees with the Python helper
        p = S.shard(B.pathtrace_params(W, H, 4), rank, world)
        assert B.tile_rows(p) == len(rows)
        results = {}
        for name in ("mandelbrot", "pathtrace"):
            if name == "mandelbrot":
                pad = np.zeros((padded, W), np.int32)
                for k, r in enumerate(rows):
                    pad[k] = O.mandelbrot_iters(W, H, 100, row_begin=r, row_end=r + 1, nthreads=1)[0]
            else:
                pad = np.zeros((padded, W, 4), np.float32)
                for k, r in enumerate(rows):
                    pad[k] = O.pathtrace(W, H, 4, math_mode=O.MATH_MC, row_begin=r, row_end=r + 1, nthreads=1)[0]
            assert S.owns_rows(p) == (len(rows) > 0)
            g = S.gather_tiles(torch.from_numpy(pad), rank, world)
            if rank == 0:
                results[name] = _deinterleave_numpy(g.numpy(), H, world, S.ROW_BLOCK)
            if name == "pathtrace":
                # round 6, the image route: every rank converts ITS tile (scale 1, no rotation) and sends 4 B/pixel; rank 0
                # de-interleaves and applies the point reflection on bytes.  The oracle's float -> u8 stands in for the device conversion.
                u8 = np.zeros((padded, W, 4), np.uint8)
                if rows:
                    u8[:len(rows)] = O.float_to_rgba8(pad[:len(rows)], 1.0).reshape(len(rows), W, 4)
                g8 = S.gather_tiles(torch.from_numpy(u8), rank, world)
                if rank == 0:
                    results["pathtrace_rgba8"] = _assemble_rgba8_numpy(g8.numpy(), W, H, world, S.ROW_BLOCK, True)
            # the step loop bench.py runs (S.Exchange: receive buffers allocated once, two buffer sets used alternately): three
            # steps with different tile contents; what rank 0 re-assembles in step i must be step i's image
            ex = S.Exchange(rank, world, pad.shape, torch.from_numpy(pad).dtype, "cpu")
            seen = []
            for i in range(3):
                t = ex.tile(i)
                t.copy_(torch.from_numpy(pad))
                if name == "mandelbrot":
                    t += i                           # a different image per step (same on every rank)
                else:
                    t *= float(i + 1)
                ex.submit(i, lambda recv, stream: seen.append(_deinterleave_numpy(recv.numpy().copy(), H, world, S.ROW_BLOCK)))
            ex.finish()
            if rank == 0:
                assert len(seen) == 3 and ex.bytes_per_rank == pad.nbytes
                for i, img in enumerate(seen):
                    exp = results[name] + i if name == "mandelbrot" else results[name] * np.float32(i + 1)
                    if name == "mandelbrot":         # padding rows of partial tiles also received + i: compare owned rows only
                        assert np.array_equal(img, exp), (name, i)
                    else:
                        assert np.array_equal(img.view(np.uint32), exp.view(np.uint32)), (name, i)
        dist.barrier()
        if rank == 0:
            q.put(results)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 70), (2, 64), (4, 601), (4, 70), (8, 70), (8, 20)])
def test_sharded_render_equals_single(O, B, world, H):
    """(8, 20): three row blocks for eight ranks — ranks 3..7 own no rows and contribute padding only."""
    W = 25 if (world, H) == (2, 70) else 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(res["mandelbrot"], O.mandelbrot_iters(W, H, 100).astype(np.int32))
    full = O.pathtrace(W, H, 4, math_mode=O.MATH_MC)
    assert np.array_equal(res["pathtrace"].view(np.uint32), full.view(np.uint32))
    # the RGBA8 exchange re-assembles to the image the reference's host post-process makes of the whole storage buffer (odd width 25
    # in one case: the middle column the reference's swap loop never touches)
    assert np.array_equal(res["pathtrace_rgba8"], O.rotate180(O.float_to_rgba8(full, 1.0).reshape(H, W, 4), W, H))


def test_sharding_helpers(B):
    import __graft_entry__ as entry
    S = entry.load_package().sharding
    for H in (600, 601, 70, 20, 16, 5):
        for n in (1, 2, 4, 8):
            owned = sorted(sum((S.rank_rows(H, r, n) for r in range(n)), []))
            assert owned == list(range(H))
            assert S.padded_tile_rows(H, n) == max(len(S.rank_rows(H, r, n)) for r in range(n))
    p = S.shard(B.mandelbrot_params(100, 600), 3, 8)
    assert (p.row_begin, p.row_end, p.row_block, p.row_stride) == (3 * S.ROW_BLOCK, 600, S.ROW_BLOCK, 8 * S.ROW_BLOCK)
    assert all(len(S.rank_rows(600 * n, r, n)) == 600 for n in (1, 2, 4, 8) for r in range(n))   # bench weak scaling: equal tiles
    p = S.shard(B.mandelbrot_params(100, 20), 5, 8)      # more ranks than row blocks: rank 5 owns nothing
    assert not S.owns_rows(p) and B.tile_rows(p) == 0 and S.rank_rows(20, 5, 8) == []
    assert S.owns_rows(S.shard(B.mandelbrot_params(100, 20), 2, 8)) and S.rank_rows(20, 2, 8) == [16, 17, 18, 19]
    assert S.ROW_BLOCK == B.lib().mc_row_block() == 8
    p = S.shard(B.mandelbrot_params(100, 600), 0, 1)
    assert (p.row_begin, p.row_end, p.row_block, p.row_stride) == (0, 600, 0, 0)
    with pytest.raises(RuntimeError):
        S.assemble_device(None, torch.zeros(1, 2, 2, 4), 2, 2, 1, torch.zeros(2, 2, 4))


def test_plain_bench_command_starts_one_rank_per_gpu_as_a_child_process():
    """The driver's plain command shape with --gpus N > 1: bench.py starts the launcher itself, as a child,
    before it touches the GPU, and relays the outcome.  Without a GPU here the ranks it started fail loudly ("no HIP device visible":
    there is no CPU fallback) — seen once per rank, so both ranks ran — and the launcher's failure is this command's exit code; under a
    profiler's preload nothing is started and the launcher line is the hint."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        pytest.skip("GPU box: tests/test_gpu_bench_contract.py runs the real thing")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and not p.stdout.strip()
    assert p.stderr.count("no HIP device visible") == 2 and "the launcher exited with" in p.stderr
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=dict(env, ROCPROFILER_TEST_MARK="1"),
                       capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "torch.distributed.run" in p.stderr and "no HIP device" not in p.stderr
