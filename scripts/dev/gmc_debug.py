import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
from oracle import oracle
oracle.lib()
rng = np.random.default_rng(21)
x = np.where(rng.random((2, 70, 150, 1)) < 0.08, rng.uniform(1, 80, (2, 70, 150, 1)), 0).astype(np.float32)
ms = {"floats": np.where(x > 0, rng.choice([0.5, 1.0, 2.0, 3.5], x.shape), 0).astype(np.float32),
      "random01": (rng.random(x.shape) < 0.05).astype(np.float32),
      "negzero": np.where(rng.random(x.shape) < 0.3, np.float32(-0.0), (x > 0.1).astype(np.float32)).astype(np.float32),
      "ones": np.ones_like(x)}
for name, m in ms.items():
    xx = x.copy(); xx[0, 3, 5, 0] = -0.0
    got, want = pkg.generate_multi_channel(xx, m, 7, 4), oracle.generate_multi_channel(xx, m, 7, 4)
    for k, (g, w) in enumerate(zip(got, want)):
        bad = (g != w) | (np.signbit(g) != np.signbit(w))
        print(name, "step", k, "bad", int(bad.sum()))
        for idx in np.argwhere(bad)[:5]:
            b, i, j, _ = idx
            print("   ", idx[:3], "got", g[b, i, j, 0], "want", w[b, i, j, 0], "x", xx[b, i, j, 0], "m", m[b, i, j, 0])
