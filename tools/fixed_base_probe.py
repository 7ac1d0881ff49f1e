#!/usr/bin/env python3
"""Developer tool: where the fixed-base kernel's time goes.  Times bjj_mul_fixed_base_dev on 2^20 items with
(a) random scalars (every gather a different 128-byte line of the table) and (b) one scalar repeated (all gathers hit
the same few lines: arithmetic only), per window width."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
dev = torch.device("cuda:0"); n = int(os.environ.get("N", 1 << 20))
st = torch.cuda.Stream()
sc = w.scalars_254(n)
same = np.tile(sc[:1], (n, 1))
d_rand = torch.from_numpy(sc.reshape(-1)).to(dev); d_same = torch.from_numpy(same.reshape(-1)).to(dev)
d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
for W in [int(a) for a in sys.argv[1:]] or [0]:
    ctx = bjj.Context(0, W); ctx.reserve(n)
    res = []
    for d_sc in (d_rand, d_same):
        with torch.cuda.stream(st):
            for _ in range(3): ctx.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr(), st.cuda_stream)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            ev[0].record(st)
            for i in range(10):
                ctx.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr(), st.cuda_stream); ev[i + 1].record(st)
        st.synchronize()
        res.append(min(ev[i].elapsed_time(ev[i + 1]) for i in range(10)))
    i = ctx.info()
    print("W=%2d windows=%2d table %8.1f MB   random scalars %.3f ms   one scalar repeated %.3f ms" % (i.window_bits, i.n_windows, i.table_bytes / 1e6, res[0], res[1]))
    ctx.close()
