"""Debug helper (GPU box): run the bench workload for a few steps and list streams whose FIBs fail."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, argparse
import bench
from dabstar_amd import lib as dx
from tools import dab_synth as ds
args = bench.parse()
from dabstar_amd import shard as _sh
if os.environ.get("OLD_PARAMS"):
    _rng = np.random.default_rng(77)
    _old = [(int(_rng.integers(0, bench.TF)), float(_rng.integers(-1900, 1901)) / 0.96) for _ in range(4096)]
    _sh.stream_params = lambda g, tf=196608: _old[g]
dev = torch.device("cuda", 0)
subch = ds.default_subchannels(18, 64)
eng = dx.Engine(n_streams=args.streams, ring_frames=10, max_subch=18, out_frames=2)
eng.set_subchannels(subch)
bench.fill_rings(eng, torch, dev, args, 0, subch)
eng.commit(9 * bench.TF)
for i in range(40):
    eng.commit(bench.TF); eng.process(1, sync=False)
eng.synchronize()
bad = []
for s in range(args.streams):
    st = eng.stats(s)
    if st["fic_ratio_percent"] < 100 or st["sf_fail"]:
        bad.append((s, st))
print("bad streams:", len(bad))
from dabstar_amd import shard
params = [shard.stream_params(g) for g in range(args.streams)]
for s, st in bad[:8]:
    print(s, params[s], {k: st[k] for k in ("frames", "state", "fic_ratio_percent", "freq_offs_bb_hz", "last_start_index", "fib_ok", "fib_total", "sf_ok", "sf_fail")})
good = [s for s in range(args.streams) if s not in [b[0] for b in bad]][:3]
for s in good:
    st = eng.stats(s)
    print("good", s, params[s], {k: st[k] for k in ("frames", "fic_ratio_percent", "freq_offs_bb_hz", "last_start_index")})

# dump the first bad stream's ring and run the oracle + a fresh single-stream engine on exactly that IQ
if bad:
    import ctypes as C
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_lib as ol
    s = bad[0][0]
    ptr, cap = eng.ring_ptr(s)
    host = np.zeros(cap, np.complex64)
    H = bench.hip()
    assert H.hipMemcpy(host.ctypes.data_as(C.c_void_p), ptr, cap * 8, 2) == 0
    x = np.ascontiguousarray(np.tile(host, 3))
    L = ol.oracle(); rx = L.ora_rx_create(ol.make_descs(subch), 18)
    n = L.ora_rx_run(rx, x, len(x), 100)
    capt = L.ora_rx_get_capture(rx).contents
    print("oracle on the same ring: frames", n, "fbb", np.ctypeslib.as_array(capt.fbb, (n,))[:12],
          "crc", np.ctypeslib.as_array(capt.fib_crc, (n * 12,)).reshape(n, 12).sum(1)[:12], "start", np.ctypeslib.as_array(capt.start_idx, (n,))[:12])
    e1 = dx.Engine(n_streams=1, ring_frames=31, max_subch=18)
    e1.set_subchannels(subch); e1.push_iq(0, x)
    fr = 0
    for i in range(14):
        e1.process(1); st = e1.stats(0)
        print(" gpu single:", st["frames"], st["state"], st["last_start_index"], "%.2f" % st["freq_offs_bb_hz"], st["fic_ratio_percent"], st["samples_consumed"])
