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


def parse_pci_bus_id(text: str):
    """'0000:05:00.0' -> (domain, bus, device, function); anything else -> (-1, -1, -1, -1)."""
    try:
        dom, bus, rest = text.strip().split(":")
        dev, fn = rest.split(".")
        return int(dom, 16), int(bus, 16), int(dev, 16), int(fn, 16)
    except Exception:
        return -1, -1, -1, -1


def gather_rank_reports(dist, torch, device, rank: int, local_rank: int, pci, frames: int, elapsed_s: float, host_id: int = 0):
    """What every rank measured by itself, gathered on all ranks (two all_gathers of a few numbers: bookkeeping, not data
    path): [{rank, local_rank, pci_bus_id, host_id, frames, elapsed_s, frames_per_s}] ordered by rank.  With it the
    N-GPU line proves by itself that N ranks sat on N DIFFERENT devices and how evenly they ran.  dist None: one entry."""
    ints = [rank, local_rank, int(pci[0]), int(pci[1]), int(pci[2]), int(pci[3]), int(frames), int(host_id)]
    if dist is None:
        rows_i, rows_f = [ints], [[elapsed_s]]
    else:
        world = dist.get_world_size()
        ti = torch.tensor(ints, dtype=torch.int64, device=device)
        tf_ = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
        gi = [torch.zeros_like(ti) for _ in range(world)]
        gf = [torch.zeros_like(tf_) for _ in range(world)]
        dist.all_gather(gi, ti)
        dist.all_gather(gf, tf_)
        rows_i, rows_f = [g.tolist() for g in gi], [g.tolist() for g in gf]
    out = []
    for ri, rf in zip(rows_i, rows_f):
        bus = "%04x:%02x:%02x.%x" % (ri[2], ri[3], ri[4], ri[5]) if ri[2] >= 0 else "n/a"
        out.append({"rank": ri[0], "local_rank": ri[1], "pci_bus_id": bus, "host_id": ri[7], "frames": ri[6], "elapsed_s": round(rf[0], 6),
                    "frames_per_s": round(ri[6] / rf[0], 1) if rf[0] > 0 else 0.0})
    return sorted(out, key=lambda r: r["rank"])


def check_distinct_devices(reports):
    """Two ranks of one host on the same PCI device: the run is not an N-GPU run.  Returns the offending pairs."""
    seen, bad = {}, []
    for r in reports:
        key = (r["host_id"], r["pci_bus_id"])
        if r["pci_bus_id"] != "n/a" and key in seen:
            bad.append((seen[key], r["rank"], r["pci_bus_id"]))
        seen.setdefault(key, r["rank"])
    return bad
