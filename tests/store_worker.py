"""One rank of a two-(or more-)process run of dmg_multi's store transport on ONE GPU: every process owns a band, maps its
neighbours' landing slots and flag words through hipIpc handles (the handles travel through files in a directory: the host's
own bootstrap, dmg_multi_store_blob / _connect), and the halo rows cross process boundaries by direct stores.  The path a
rank-per-GPU job takes over xGMI, exercised on the one-GPU test box.
usage: python store_worker.py <rank> <world> <dir> <W> <H> <iters> <ghost> <order: plain|overlap>"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demiurge_amd import capi, synth

rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
W, H, iters, ghost = (int(x) for x in sys.argv[4:8])
order = capi.MULTI_OVERLAP_LAST_LAUNCH if sys.argv[8] == "overlap" else capi.MULTI_EXCHANGE_THEN_SWEEP
grid = capi.make_grid(W, H)
m = capi.Multi(grid, rank=rank, world=world, unique_id=None, device=0, ghost=ghost, flags=capi.MULTI_STORE | order)
blob = m.store_blob()
tmp = os.path.join(d, "blob%d.tmp" % rank)
open(tmp, "wb").write(blob)
os.rename(tmp, os.path.join(d, "blob%d" % rank))                       # (atomic: a reader never sees half a blob)
blobs = []
for r in range(world):
    p = os.path.join(d, "blob%d" % r)
    t0 = time.time()
    while not os.path.exists(p):
        if time.time() - t0 > 120:
            raise SystemExit("rank %d: no blob from rank %d" % (rank, r))
        time.sleep(0.01)
    blobs.append(open(p, "rb").read())
m.store_connect(blobs)
m.set_terrain(synth.terrain(W, H)); m.set_divw(synth.divw(W, H))
for rep in range(2):                                                  # twice: the sequence numbers and slots carry on
    m.jacobi(iters, zero_first=True)
got = m.get(want=("pressure",))["pressure"]
st = m.stats()
np.save(os.path.join(d, "p%d.npy" % rank), got)
print("rank %d: %d exchanges, %d bytes, order %d" % (rank, st["exchanges"], st["exchange_bytes"], st["order"]), flush=True)
# nobody unmaps a neighbour that may still push: leave together
open(os.path.join(d, "done%d" % rank), "w").write("1")
t0 = time.time()
while not all(os.path.exists(os.path.join(d, "done%d" % r)) for r in range(world)):
    if time.time() - t0 > 120:
        break
    time.sleep(0.01)
m.close()
