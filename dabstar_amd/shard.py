"""Multi-GPU sharding of independent ensembles: one process per GPU, streams are split across
ranks with no data-path collective; the only communication is a SUM all-reduce of the result counters and a MAX
all-reduce of the elapsed time (RCCL on GPUs, gloo in the CPU tests)."""
import numpy as np


def streams_for_rank(rank: int, world: int, streams_per_gpu: int):
    """Global stream ids owned by `rank` (contiguous block: stream s lives on GPU s // streams_per_gpu)."""
    return list(range(rank * streams_per_gpu, (rank + 1) * streams_per_gpu))


def stream_params(global_id: int, tf: int = 196608):
    """Deterministic per-stream channel parameters (timing offset in samples, CFO in Hz) from the global id."""
    rng = np.random.default_rng(100003 * (global_id + 1))
    toff = int(rng.integers(0, tf))
    cfo = float(rng.integers(-1900, 1901)) / 0.96      # phase-continuous over a 10-frame (0.96 s) ring
    return toff, cfo


def reduce_results(dist, torch, device, elapsed_s: float, counters):
    """MAX over ranks of the elapsed time, SUM over ranks of the integer counters. dist may be None (1 rank)."""
    if dist is None:
        return elapsed_s, [int(c) for c in counters]
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([int(v) for v in counters], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [int(v) for v in c.tolist()]
