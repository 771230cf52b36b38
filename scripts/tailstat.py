"""Reads the table an RTOW_TAILSTAT build writes (variants/tailstat.so, RTOW_TAILSTAT_OUT=path): per wave, the 100 MHz
times of its start, of the first trip in which a lane found the queue empty and of its end, its trips, and what it
still held at that moment (lanes with an item, samples left in their items, items left in its pool)."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
t0 = t[:, 0].min()
start, empty, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0  # us
trips, after, live, left, pool = t[:, 3], t[:, 4], t[:, 5], t[:, 6], t[:, 7]
print(f"waves {len(t)}; launch (first start to last end) {end.max():.1f} us; starts spread over {start.max():.1f} us")
print(f"queue first seen empty: min {empty.min():.1f}  mean {empty.mean():.1f}  max {empty.max():.1f} us")
print(f"wave ends: min {end.min():.1f}  mean {end.mean():.1f}  max {end.max():.1f} us; (end - empty): mean {np.mean(end-empty):.1f}  p50 {np.percentile(end-empty,50):.1f}  p90 {np.percentile(end-empty,90):.1f}  p99 {np.percentile(end-empty,99):.1f}  max {np.max(end-empty):.1f} us")
print(f"trips per wave {trips.mean():.1f} ({(end-start).mean()/trips.mean():.2f} us per trip over the launch); trips after empty: mean {after.mean():.1f}  p90 {np.percentile(after,90):.0f}  max {after.max():.0f}; us per trip after empty {np.sum(end-empty)/max(after.sum(),1):.2f}")
print(f"held at queue-empty: lanes with an item {live.mean():.1f}, samples left in them {left.mean():.1f}, items in the pool {pool.mean():.1f} (max {pool.max():.0f})")
late = end > np.percentile(end, 99)
print(f"the last 1 % of waves to end: (end - empty) {np.mean((end-empty)[late]):.1f} us, trips after empty {after[late].mean():.1f}, samples held {left[late].mean():.1f}, pool {pool[late].mean():.1f}")
h, edges = np.histogram(end.max() - end, bins=[0, 25, 50, 100, 200, 300, 400, 600, 800, 1200, 1e9])
print("waves by (launch end - own end) us:", dict(zip([f"<{int(e)}" for e in edges[1:-1]] + [">=1200"], h.tolist())))
