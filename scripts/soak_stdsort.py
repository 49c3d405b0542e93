"""Randomised soak of the device std::sort (stdsort.hip) against libstdc++'s own (the oracle's std::sort on (key, index) pairs, key-only comparator): random lengths
up to 300 000, key alphabets from 1 to n, structured patterns (sorted, reversed, organ pipe, sawtooth, blocks, ring-like piecewise monotone runs), one- and two-cloud calls. usage: soak_stdsort.py [trials] [seed]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
mla = importlib.import_module("m-loam_amd")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.build()
c = mla.Context(0)
t0 = time.time(); total = 0
for trial in range(trials):
    n = int(rng.choice([rng.integers(2, 3000), rng.integers(3000, 20000), rng.integers(20000, 130000), rng.integers(130000, 300000)], p=[0.2, 0.3, 0.4, 0.1]))
    nv = int(rng.choice([1, 2, 3, 17, max(n // 7, 1), n, 4 * n]))
    i = np.arange(n)
    kind = int(rng.integers(0, 8))
    def ring_like():
        # piecewise monotone runs of random length, direction and step, as the voxel keys along a scan ring are: median-of-three runs out of its depth budget on
        # such input and the leftovers are heap-sorted (the path the per-ring voxel filter's slowest rings take)
        out, base = [], int(rng.integers(0, 1000))
        while sum(len(o) for o in out) < n:
            m = int(rng.integers(3, 220)); step = int(rng.choice([-3, -1, -1, 0, 1, 1, 2])); rep = int(rng.integers(1, 4))
            run = base + step * (np.arange(m) // rep)
            out.append(run); base = int(run[-1]) + int(rng.integers(-40, 40))
        k = np.concatenate(out)[:n]
        return k - k.min()
    keys = [lambda: rng.integers(0, nv, n), lambda: i % nv, lambda: (n - i) % nv, lambda: np.where(i < n // 2, i, n - i) % nv, lambda: np.sort(rng.integers(0, nv, n)),
            lambda: np.sort(rng.integers(0, nv, n))[::-1], lambda: (i // max(n // max(nv, 1), 1)), ring_like][kind]().astype(np.int32)
    if rng.random() < 0.3 and n > 40:
        n0 = int(rng.integers(1, n - 1))
        want = np.concatenate([O.std_sort_permutation(keys[:n0]), n0 + O.std_sort_permutation(keys[n0:])])
        got = c.std_sort_permutation(keys, n0=n0, mode=1)
    else:
        want = O.std_sort_permutation(keys)
        got = c.std_sort_permutation(keys, mode=1)
    if not np.array_equal(got, want):
        raise SystemExit(f"STDSORT trial {trial}: n {n}, alphabet {nv}, pattern {kind}: first difference at {int(np.argmax(got != want))}")
    total += n
print(f"device std::sort: {trials} random sequences ({total} elements) equal to libstdc++'s permutation  [{time.time() - t0:.0f} s]")
