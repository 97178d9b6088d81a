#!/usr/bin/env python3
"""K2 alone on 256 ch x 2^21 @ 2 MS/s for the three decimator plans of the C4 workload (AM 10 kHz, FM 15 kHz, SSB 20 kHz
maximum bandwidth): ms per launch by stage sequence.  (GPU box)"""
import sys, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # the repo root
import torch, cutesdr_amd as ca
C,T=256,1<<21
dev=torch.device("cuda",0); st=torch.cuda.current_stream().cuda_stream
x=torch.randn((C,T,2),device=dev)*100
res={}
for bw in (10000.0, 15000.0, 20000.0):
    dc=ca.DownConvertBatch(C); dc.set_data_rate(2e6,bw)
    y=torch.empty((C,T//16,2),device=dev)
    f=lambda: dc.process_ptr(x.data_ptr(),T,T,y.data_ptr(),T//16,st)
    for _ in range(30): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    res[str(dc.stages())]=round(e0.elapsed_time(e1)/50,4); dc.close()
print(json.dumps(res))
