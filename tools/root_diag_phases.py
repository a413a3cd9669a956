import sys, numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.int64)
d = d[d[:,0] > 0]
us = (d - d[:, :1]) / 100.0
names = ["start"] + [f"b{b}:{p}" for b in range(4) for p in ("preF","postF","B2","B3","B4","B5")] + ["end"]
m = us.mean(axis=0)
prev = 0
for n, v in zip(names, m[:26]):
    print(f"{n:10s} {v:8.1f}  (+{v-prev:6.1f})"); prev = v
