#!/usr/bin/env python3
"""Lane census of sim_column_kernel's accumulation, computed on the host from the URM alone (no GPU): how many of the 64 lanes
carry a live 8-entry chunk per ds_add instruction under a given dealing of a column's users to wavefronts / lane groups.

  python scripts/analysis/sim_lane_census.py [ml20m|netflix]

Model (sim.hip, accumulation phase): a column's users go to WAVES wavefronts; a wavefront takes 64 users per round; its 64 / G lane
groups take the round's users m = g, g + GPW, ...; a user with k = ceil(L / 8) chunks costs ceil(k / G) steps of its group; a round
lasts as many steps as its slowest group (every step issues the 8 ds_adds wave-wide, masked).  lanes per instruction = chunks / steps.
Variants: the order of a column's users (ascending row id as the CSC has it / by descending profile length) and how they are dealt to
the wavefronts (contiguous runs / round-robin), and G."""
import sys
import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402


def census(urm, G, order="row", deal="runs", waves=16, dynamic=False, sample=None):
    csc = urm.tocsc()
    lens = np.diff(urm.indptr).astype(np.int64)
    chunks_of = (lens + 7) // 8
    GPW = 64 // G
    tot_chunks = tot_steps = 0
    wave_steps_sum = wave_steps_max = 0            # per column: sum over waves / WAVES x slowest wave (barrier at the column's end)
    cols = range(urm.shape[1]) if sample is None else sample
    for c in cols:
        users = csc.indices[csc.indptr[c]:csc.indptr[c + 1]]
        n = len(users)
        if n == 0:
            continue
        k = chunks_of[users]
        if order == "len":
            k = np.sort(k)[::-1]
        elif order == "steps":                      # descending number of steps ceil(k / G): what a counting sort on the device gives
            k = k[np.argsort(-(-(-k // G)), kind="stable")]
        elif order == "lenclass":                   # 2 classes per octave, stable inside a class
            cls = np.floor(2 * np.log2(np.maximum(k, 1))).astype(np.int64)
            k = k[np.argsort(-cls, kind="stable")]
        if deal == "dyn":                           # wavefronts take the next 64 users of the column from a shared counter when they finish a round
            load = np.zeros(waves, np.int64)
            for r in range(0, n, 64):
                kr = k[r:r + 64]
                sr = -(-kr // G)
                pad = (-len(sr)) % GPW
                t = np.concatenate([sr, np.zeros(pad, np.int64)]).reshape(-1, GPW).sum(0).max()
                load[np.argmin(load)] += int(t)
                tot_chunks += int(kr.sum())
            tot_steps += int(load.sum())
            wave_steps_sum += int(load.sum())
            wave_steps_max += waves * int(load.max())
            continue
        if deal == "runs":
            per = -(-n // waves)
            lists = [k[w * per:(w + 1) * per] for w in range(waves)]
        else:                                       # round-robin in units of `deal` users ("s" suffix: serpentine, every other stripe reversed)
            serp = str(deal).endswith("s")
            unit = int(str(deal).rstrip("s"))
            idx = np.arange(n)
            owner = (idx // unit) % waves
            if serp:
                stripe = idx // (unit * waves)
                owner = np.where(stripe % 2 == 1, waves - 1 - owner, owner)
            lists = [k[owner == w] for w in range(waves)]
        col_wave = []
        for kw in lists:
            steps_w = 0
            for r in range(0, len(kw), 64):
                kr = kw[r:r + 64]
                s = -(-kr // G)
                if dynamic:                         # groups take the next user when they finish one: list scheduling
                    load = np.zeros(GPW, np.int64)
                    for x in s:
                        load[np.argmin(load)] += x
                    t = load.max()
                else:
                    pad = (-len(s)) % GPW
                    t = np.concatenate([s, np.zeros(pad, np.int64)]).reshape(-1, GPW).sum(0).max()
                steps_w += int(t)
                tot_chunks += int(kr.sum())
            tot_steps += steps_w
            col_wave.append(steps_w)
        wave_steps_sum += sum(col_wave)
        wave_steps_max += waves * max(col_wave) if col_wave else 0
    return tot_chunks / max(tot_steps, 1), wave_steps_sum / max(wave_steps_max, 1), tot_steps


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "ml20m"
    urm = bench.load_urm(name)
    rng = np.random.default_rng(0)
    cost = np.asarray(urm.T.dot(np.diff(urm.indptr).astype(np.float64))).ravel()      # pairs per column
    # cost-weighted sample of columns (the census of all 26 744 takes minutes in Python)
    sample = rng.choice(urm.shape[1], size=1500, replace=False, p=cost / cost.sum())
    print("%s: weighted mean profile %.0f entries" % (name, cost.sum() / urm.nnz))
    for G in (32, 16, 8, 4):
        for order, deal, dyn in (("row", "runs", False), ("len", str(64 // G), False), ("steps", str(64 // G), False), ("steps", str(64 // G) + "s", False),
                                 ("steps", "dyn", False), ("row", "dyn", False)):
            lanes, wave_bal, steps = census(urm, G, order, deal, dynamic=dyn, sample=sample)
            print("G=%2d order=%-8s deal=%-4s dynamic=%d : %.1f lanes per ds_add, wave balance %.3f, steps x waves %.3e (rel. cost %.3f)" % (
                G, order, deal, dyn, lanes, wave_bal, steps, steps / wave_bal))
