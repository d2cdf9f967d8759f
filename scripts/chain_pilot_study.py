"""CPU study: how well do the residuals of the first sweeps of the chain fixed point predict the sweep
count of a job?  (Question behind a "pilot" launch that would order the FIRST evaluation of a grid, which
today runs in launch order and pays ~13 % for its tail.)  Plain numpy, batched over the energies; the
iteration is the one of surfG1D.py:271-288 (see oracle/negf_oracle.py:chain1d_g)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import chain_lead, legendre_grid  # noqa: E402


def run(seed, M, eta=1e-4, conv=1e-5, r=0.1, max_iter=2000, marks=(10, 20, 30, 40, 60, 80)):
    alpha, Sa, beta, Sb = chain_lead(50, seed)
    E, _ = legendre_grid(M, -2.0, 2.0)
    z = (E + 1j * eta)[:, None, None]
    A = z * Sa - alpha
    B = z * Sb - beta
    Bd = np.conj(np.swapaxes(B, 1, 2))
    g = np.linalg.inv(A)
    count = np.zeros(M, dtype=int)
    active = np.ones(M, dtype=bool)
    hist = {}
    t0 = time.time()
    for k in range(1, max_iter + 1):
        idx = np.nonzero(active)[0]
        if idx.size == 0:
            break
        ga = g[idx]
        gn = np.linalg.inv(A[idx] - B[idx] @ ga @ Bd[idx])
        diff = np.max(np.abs(gn - ga) / np.maximum(np.abs(gn), 1e-12), axis=(1, 2))
        g[idx] = r * gn + (1 - r) * ga
        count[idx] = k
        if k in marks:
            d = np.full(M, np.nan); d[idx] = diff; hist[k] = d
        active[idx[diff <= conv]] = False
        if k % 200 == 0:
            print(f"  sweep {k}: {idx.size} active, {time.time() - t0:.0f} s", flush=True)
    return E, count, hist


def makespan(lengths, order, slots=768):
    import heapq
    h = [0.0] * slots
    heapq.heapify(h)
    for j in order:
        t = heapq.heappop(h)
        heapq.heappush(h, t + lengths[j])
    return max(h)


if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    res = [run(31, M), run(32, M)]
    # jobs interleaved (energy, contact) like the launch: job = 2 b + c
    counts = np.stack([res[0][1], res[1][1]], axis=1).reshape(-1).astype(float)
    np.save("/tmp/chain_pilot_counts.npy", counts)
    print("jobs", counts.size, "mean", counts.mean(), "frac full", np.mean(counts >= 2000))
    ideal = counts.sum() / 768
    print(f"ideal {ideal:.0f}  launch order {makespan(counts, np.arange(counts.size)):.0f}  "
          f"longest first {makespan(counts, np.argsort(-counts, kind='stable')):.0f}")
    for k in sorted(res[0][2]):
        d = np.stack([res[0][2][k], res[1][2][k]], axis=1).reshape(-1)
        # jobs finished before the mark: their count is known
        key = np.where(np.isnan(d), -1.0, d)
        order = np.argsort(-key, kind="stable")
        ms = makespan(counts, order)
        # a pilot of k sweeps for every job costs k * jobs / 768 on top
        print(f"pilot {k:3d}: makespan {ms:.0f} (+pilot {k * counts.size / 768:.0f}) "
              f"spearman {np.corrcoef(np.argsort(np.argsort(key)), np.argsort(np.argsort(counts)))[0, 1]:.3f}")
        np.save(f"/tmp/chain_pilot_diff{k}.npy", d)
