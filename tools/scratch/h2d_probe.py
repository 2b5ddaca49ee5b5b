import sys, time
import numpy as np
sys.path.insert(0, "tests")
from gpu_common import load_binding
B = load_binding(); ctx = B.Context(0)
for kb in (4, 32, 64, 128, 512, 1024, 2048, 4096, 8192, 65536):
    a = np.zeros(kb * 1024 // 8, np.uint64); d = ctx.alloc(a.nbytes)
    d.upload(a)
    t = time.perf_counter()
    for _ in range(5): d.upload(a)
    print(f"H2D pageable {kb} KB: {(time.perf_counter() - t) / 5 * 1e3:.3f} ms", flush=True)
    d.free()
