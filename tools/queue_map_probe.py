#!/usr/bin/env python3
"""Developer probe (run under `rocprofv3 --kernel-trace`): which HSA queue does each HIP stream land on?  Launches a small
fixed-base batch on the context's stream, on two torch streams and through the host-pointer pipeline in a chosen order; the
trace's Queue_Id / Stream_Id columns give the mapping (tools/queue_map_report.py prints it).  argv[1] = order:
  "ab_first"   torch streams A, B used first, then the host-pointer call, then A, B again
  "host_first" the host-pointer call first, then A, B"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

order = sys.argv[1] if len(sys.argv) > 1 else "ab_first"
ctx = bjj.Context(0, 16)
dev = torch.device("cuda", 0)
n = 200000
sc = w.scalars_254(n)
d_sc = torch.from_numpy(np.ascontiguousarray(sc).reshape(-1)).to(dev)
d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
A, B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def dev_launch(st, n_items):      # the batch size tags the launch in the trace (grid size)
    ctx.mul_fixed_base_dev(d_sc.data_ptr(), n_items, d_out.data_ptr(), st)
    ctx.sync()


def host_call():
    ctx.mul_fixed_base(sc)          # 200 000 items: 3 chunks -> both lanes, s_in, s_out, s_out2
    ctx.eddsa_verify(np.zeros((64, 64), np.uint8), np.zeros((64, 64), np.uint8), np.zeros((64, 32), np.uint8), np.zeros((64, 32), np.uint8))


dev_launch(0, 64)                   # context stream
if order == "ab_first":
    dev_launch(A.cuda_stream, 128); dev_launch(B.cuda_stream, 192)
    host_call()
    dev_launch(A.cuda_stream, 128); dev_launch(B.cuda_stream, 192)
else:
    host_call()
    dev_launch(A.cuda_stream, 128); dev_launch(B.cuda_stream, 192)
torch.cuda.synchronize()
ctx.close()
