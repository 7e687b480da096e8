"""tools/cold_step_compare.py <dir> <tagA> <tagB> ... — entry-by-entry comparison of consecutive tools/cold_step_dump.py outputs"""
import sys, numpy as np
d = sys.argv[1]; tags = sys.argv[2:]
L = {k: np.load("%s/cold_%s.npz" % (d, k)) for k in tags}
for a, b in zip(tags[:-1], tags[1:]):
    out = []
    for k in L[a].files:
        x, y = L[a][k], L[b][k]
        if np.array_equal(x, y): out.append(k + ": identical")
        else:
            x = x.astype(float); y = y.astype(float); e = np.abs(x - y)
            out.append("%s: %d of %d entries differ, max abs %.2e, max rel (floor 1e-6) %.2e" % (k, int((x != y).sum()), x.size, e.max(), (e / np.maximum(np.abs(x), 1e-6)).max()))
    print(a, "vs", b, "|", " | ".join(out))
